"""Where the host time of the step goes when the reference's OWN models/*.py + losses.py run on this package's bindings
(oracle/ref_on_binding.TrainingStep: train.py:159-185 statement for statement): cProfile over 96 steps (6 occupancy updates through
the reference's Python), cumulative and by own time; and the step time with and without the update steps.
    python tools/ref_files_cprofile.py > gpurun_out/ref_files_cprofile.txt"""
import argparse, cProfile, io, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from oracle import ref_on_binding as R
from ngp_pl_amd.optim import FusedAdam

args = argparse.Namespace(rays=0, res=800, images=100)
dev = torch.device("cuda", 0)
loop = bench.Loop("lego", args, dev, 0, 1, None)
loop.steps(600)
theirs = R.make_model(loop.model.scale, dev)
theirs.load_state_dict(loop.model.state_dict(), strict=False)
step = R.TrainingStep(theirs, FusedAdam, lr=loop.trainer.opt.param_groups[0]["lr"])
step.global_step = loop.trainer.global_step
for _ in range(40):
    cur = loop.draw(on_side=False); step(cur[0], cur[1], cur[2])
torch.cuda.synchronize()
# per-step wall time by position in the 16-step cycle
per = []
for _ in range(96):
    cur = loop.draw(on_side=False)
    torch.cuda.synchronize(); t = time.perf_counter()
    step(cur[0], cur[1], cur[2])
    torch.cuda.synchronize(); per.append(((step.global_step - 1) % 16, time.perf_counter() - t))
upd = [d for k, d in per if k == 0]; rest = [d for k, d in per if k != 0]
print("synchronised per step: update steps %.3f ms (n=%d), other steps %.3f ms (n=%d), mean %.3f ms" % (
    1e3 * sum(upd) / max(len(upd), 1), len(upd), 1e3 * sum(rest) / len(rest), len(rest), 1e3 * sum(d for _, d in per) / len(per)))
pr = cProfile.Profile()
pr.enable()
for _ in range(96):
    cur = loop.draw(on_side=False); step(cur[0], cur[1], cur[2])
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(60); print(s.getvalue()[:14000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35); print(s.getvalue()[:8000])
