"""Scratch: trajectory of the 16 384-ray recipe (lr 2e-2) under the env switches given on the command line; prints rm_s / vr_s / psnr every 40 steps."""
import argparse, os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
args = argparse.Namespace(rays=int(os.environ.get("RAYS", 0)), res=800, images=100)
dev = torch.device("cuda", 0)
loop = bench.Loop(os.environ.get("WORKLOAD", "lego16k"), args, dev, 0, 1, None)
out = []
for i in range(11):
    loop.steps(40)
    m = loop.trainer.metrics()
    out.append((loop.trainer.global_step, round(m["rm_s"], 2), round(m["vr_s"], 2), round(m["psnr"], 2), round(m["loss"], 5)))
print(json.dumps(out))
