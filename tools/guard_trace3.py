"""Scratch: the bench legs, then the unbounded loop with a zero-ray census of every drawn batch taken on the marching stream
right behind the sampler (no host sync inside the loop)."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from ngp_pl_amd import _lib
from ngp_pl_amd.bench_support import render_fps

ap = argparse.ArgumentParser()
ap.add_argument("--legs", default="roofline,render,render_ref,api")
ap.add_argument("--lego-steps", type=int, default=400)
a = ap.parse_args()
args = argparse.Namespace(rays=0, res=800, images=100, setup_steps=320)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
loop = bench.Loop("lego", args, dev, 0, 1, None)
loop.steps(a.lego_steps)
legs = a.legs.split(",")
if "roofline" in legs: bench.kernel_roofline(loop, 0.44)
if "render" in legs: render_fps(loop.model, loop.data, n_frames=2, chunk_scale=4, probe_cap=64)
if "render_ref" in legs: render_fps(loop.model, loop.data, n_frames=2)
if "api" in legs: bench.api_path_rate(loop, n_steps=10)
torch.cuda.synchronize()
print("legs %s done, guards %s" % (legs, _lib.march_guard_counts()), flush=True)
del loop
torch.cuda.empty_cache()
loop = bench.Loop("unbounded", args, dev, 0, 1, None)
tr = loop.trainer
print("poses min |row sum|", float(loop.data.poses.abs().sum((1, 2)).min()), "directions zero rows", int((loop.data.directions.abs().sum(1) == 0).sum()), flush=True)
N = 60
census = torch.zeros(N, 4, dtype=torch.int64, device=dev)
for i in range(N):
    nxt = loop.draw()
    with torch.cuda.stream(tr.side):
        census[i, 0] = (nxt[0].abs().sum(1) == 0).sum(); census[i, 1] = (nxt[1].abs().sum(1) == 0).sum()
        census[i, 2] = (loop.cur[0].abs().sum(1) == 0).sum(); census[i, 3] = (loop.cur[1].abs().sum(1) == 0).sum()
    tr.step(loop.cur[0], loop.cur[1], loop.cur[2], next_batch=(nxt[0], nxt[1]))
    loop.cur = nxt
torch.cuda.synchronize()
c = census.cpu()
bad = [(i, c[i].tolist()) for i in range(N) if c[i].sum() > 0]
print("guards", _lib.march_guard_counts(), "batches with zero rays (step, [next_o, next_d, cur_o, cur_d]):", bad, flush=True)
