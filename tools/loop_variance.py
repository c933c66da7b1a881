"""Scratch (round 5): why does the SAME workload run at 0.39 or 0.73 ms per step from one Loop to the next inside one process?
Builds the same Loop several times in a row (small dataset: the step does not depend on it) and prints, per instance: ms per
step over 100 steps after 330, the stage times, the raw handle of its marching stream and sclk / mclk from rocm-smi.
Usage: loop_variance.py [workload] [n] ; env NGP_SHARED_SIDE=1 -> all instances share ONE marching stream."""
import os
import subprocess
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "lego_hard"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
args = types.SimpleNamespace(rays=0, res=800, images=10, setup_steps=320)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def clocks():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=10).stdout
        return " ".join(ln.split(":")[-1].strip() for ln in out.splitlines() if "sclk" in ln or "mclk" in ln)
    except Exception as e:      # noqa: BLE001
        return "n/a (%s)" % type(e).__name__


data = None
for i in range(n):
    loop = bench.Loop(wl, args, dev, 0, 1, None, data=data)
    data = loop.data
    loop.steps(330)
    dt, dte = loop.timed(100)
    r = bench.kernel_roofline(loop, dt / 100 * 1e3, n_steps=10)
    side = loop.trainer.side
    print("%d  %.4f ms/step (events %.4f)  side=%x  stage_sum=%.4f  %s  | %s" % (
        i, dt / 100 * 1e3, dte / 100 * 1e3, side.cuda_stream, r["main_stream_stage_sum_ms"],
        " ".join("%s=%.3f" % (s["stage"][:10], s["ms"]) for s in r["stages"][:6]), clocks()), flush=True)
    del loop
    torch.cuda.empty_cache()
