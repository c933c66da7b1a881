"""Train the bench's configs[1] workload for STEPS steps (default 30 000: the full recipe), then render FRAMES held-out poses with
the frame loop setting CONFIG (k2_cap64 = what bench.py reports on the trained field, device_exact = the reference's chunking).  Run under
`rocprofv3 --kernel-trace` by tools/render_trained_trace.sh, which summarises the frames' kernels into profiles/r*_render_trace.*"""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from ngp_pl_amd import synthetic as syn
from ngp_pl_amd.bench_support import render_eval

steps = int(os.environ.get("STEPS", 30000)); frames = int(os.environ.get("FRAMES", 20)); config = os.environ.get("CONFIG", "k2_cap64")
args = argparse.Namespace(rays=0, res=800, images=100)
dev = torch.device("cuda", 0)
loop = bench.Loop("lego", args, dev, 0, 1, None)
loop.trainer.steps_per_epoch = max(steps // 30, 1)
torch.cuda.synchronize(); t0 = time.perf_counter()
loop.steps(steps)
torch.cuda.synchronize(); train_s = time.perf_counter() - t0
kw = {"k4_cap64": dict(chunk_scale=4, probe_cap=64), "k2_cap64": dict(chunk_scale=2, probe_cap=64), "device_exact": dict()}[config]
poses = syn.hemisphere_poses(frames, seed=999).to(dev)
res = render_eval(loop.model, loop.data, poses, psnr=bool(int(os.environ.get("PSNR", "0"))), **kw)      # (PSNR needs ground-truth kernels between the frames: off under a trace)
res.update(config=config, train_steps=steps, train_s=train_s, frames=frames)
print(json.dumps(res), flush=True)
