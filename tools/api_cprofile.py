"""cProfile of the reference-shaped step (plain: no next_rays) on an MI355X box: where the host time of Trainer.step_autograd goes,
function by function (cumulative).  python tools/api_cprofile.py > gpurun_out/api_cprofile.txt"""
import argparse, cProfile, io, os, pstats, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

args = argparse.Namespace(rays=0, res=800, images=100)
dev = torch.device("cuda", 0)
loop = bench.Loop("lego", args, dev, 0, 1, None)
loop.steps(600)
tr = loop.trainer
for _ in range(30):
    cur = loop.draw(on_side=False); tr.step_autograd(cur[0], cur[1], cur[2])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    cur = loop.draw(on_side=False); tr.step_autograd(cur[0], cur[1], cur[2])
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("cumulative")
ps.print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print(s.getvalue()[:6000])
