"""Per-step device time around the occupancy updates, with and without the draws made ahead (NGP_OCC_DRAW_AHEAD): the driver's
training loop (bench.Loop), 600 steps past the warm-up, one HIP event per step on the main stream and the host's wall clock."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def run(ahead, n=640):
    os.environ["NGP_OCC_DRAW_AHEAD"] = "1" if ahead else "0"
    argv, sys.argv = sys.argv, sys.argv[:1]
    try:
        args = bench.parse()
    finally:
        sys.argv = argv
    dev = torch.device("cuda:0")
    loop = bench.Loop("lego", args, dev, 0, 1, None)
    loop.steps(400)                                  # past the warm-up (256 steps)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    host = []
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(n):
        t = time.perf_counter()
        loop.steps(1)
        host.append(time.perf_counter() - t)
        ev[i + 1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    gs = loop.trainer.global_step - n
    dt = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    upd = [dt[i] for i in range(n) if (gs + i) % 16 == 0]
    nxt = [dt[i] for i in range(n) if (gs + i) % 16 == 1]
    rest = [dt[i] for i in range(n) if (gs + i) % 16 > 1]
    hu = [host[i] for i in range(n) if (gs + i) % 16 == 0]
    hr = [host[i] for i in range(n) if (gs + i) % 16 > 1]
    mean = lambda v: sum(v) / max(len(v), 1)         # noqa: E731
    return {"ahead": ahead, "wall_ms_per_step": wall / n * 1e3, "device_ms": {"update_step": mean(upd), "step_after": mean(nxt), "other": mean(rest)},
            "host_ms": {"update_step": mean(hu) * 1e3, "other": mean(hr) * 1e3}, "updates": len(upd)}


if __name__ == "__main__":
    out = [run(False), run(True), run(False), run(True)]
    print(json.dumps(out, indent=1))
