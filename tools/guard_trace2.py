"""Scratch: narrow down where zero rays come from in the unbounded loop (see tools/guard_trace.py)."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from ngp_pl_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--prelude", default="lego")
ap.add_argument("--prelude-steps", type=int, default=50)
ap.add_argument("--sync", type=int, default=0)
ap.add_argument("--workload", default="unbounded")
a = ap.parse_args()
args = argparse.Namespace(rays=0, res=800, images=100, setup_steps=320)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def zero_rays(t):
    return int((t.abs().sum(1) == 0).sum())


if a.prelude != "none":
    loop = bench.Loop(a.prelude, args, dev, 0, 1, None)
    loop.steps(a.prelude_steps)
    torch.cuda.synchronize()
    print("prelude %s done, guards %s" % (a.prelude, _lib.march_guard_counts()), flush=True)
    del loop
    torch.cuda.empty_cache()
loop = bench.Loop(a.workload, args, dev, 0, 1, None)
torch.cuda.synchronize()
print("built; ring zero rays (o, d):", [(zero_rays(r[0]), zero_rays(r[1])) for r in loop.ring], "guards", _lib.march_guard_counts(), flush=True)
for i in range(40):
    loop.steps(1)
    if a.sync or i >= 39:
        torch.cuda.synchronize()
        g = _lib.march_guard_counts()
        print("step %d: S=%d guards %s ring zero rays %s" % (i, loop.trainer.last["rm_samples"], g, [(zero_rays(r[0]), zero_rays(r[1])) for r in loop.ring]), flush=True)
        if g[0]:
            break
torch.cuda.synchronize()
print("end: guards", _lib.march_guard_counts(), "ring zero rays", [(zero_rays(r[0]), zero_rays(r[1])) for r in loop.ring], flush=True)
