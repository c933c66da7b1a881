"""Frame-loop settings on the TRAINED field (30 000 steps): chunk_scale x probe_cap sweep of ngp_render_test_frame, FPS by the
reference's protocol over 60 held-out poses each."""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from ngp_pl_amd import synthetic as syn
from ngp_pl_amd.bench_support import render_eval
args = argparse.Namespace(rays=0, res=800, images=100)
dev = torch.device("cuda", 0)
loop = bench.Loop("lego", args, dev, 0, 1, None)
loop.steps(int(os.environ.get("STEPS", 30000)))
torch.cuda.synchronize()
poses = syn.hemisphere_poses(60, seed=999).to(dev)
for cs, cap in ((1, 0), (2, 32), (2, 64), (4, 32), (4, 64), (4, 128), (8, 64), (8, 128), (16, 128), (1, 0), (4, 64)):
    r = render_eval(loop.model, loop.data, poses, psnr=False, chunk_scale=cs, probe_cap=cap)
    print(json.dumps({"chunk_scale": cs, "probe_cap": cap, "fps": round(r["fps"], 1), "ms": round(r["ms_per_frame"], 4), "spr": round(r["samples_per_ray"], 3), "iters": r.get("iterations_mean")}), flush=True)
