"""Scratch: which leg of bench.py trips the marching guards?  Runs the legs one by one and prints the guard counts after each."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from ngp_pl_amd import _lib
from ngp_pl_amd.bench_support import render_fps

args = argparse.Namespace(rays=0, res=800, images=100, setup_steps=320)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def show(what):
    torch.cuda.synchronize()
    g = _lib.march_guard_counts()
    print("%-40s guards %s%s" % (what, g, (" first probe %s" % _lib.march_guard_first()) if g[0] else ""), flush=True)


loop = bench.Loop("lego", args, dev, 0, 1, None)
show("lego loop built")
loop.steps(400); show("400 lego steps")
bench.kernel_roofline(loop, 0.44); show("roofline leg")
render_fps(loop.model, loop.data, n_frames=2, chunk_scale=4, probe_cap=64); show("render leg (regrouped)")
render_fps(loop.model, loop.data, n_frames=2); show("render leg (reference chunking)")
bench.api_path_rate(loop, n_steps=10); show("api leg")
del loop
torch.cuda.empty_cache()
loop = bench.Loop("unbounded", args, dev, 0, 1, None)
show("unbounded loop built")
for k in range(8):
    loop.steps(50); show("unbounded steps %d" % (50 * (k + 1)))
render_fps(loop.model, loop.data, n_frames=1, exp_step_factor=1 / 256.); show("unbounded render (reference chunking)")
render_fps(loop.model, loop.data, n_frames=1, chunk_scale=4, probe_cap=64, exp_step_factor=1 / 256.); show("unbounded render (regrouped)")
