"""Where the host time of the reference-shaped step goes (Trainer.step_autograd = render() + NeRFLoss + autograd + FusedAdam):
perf_counter stamps between the segments of a step, NO device synchronisation in between (the stamps are host enqueue time; the
one blocking point is render()'s wait for its batch's sample count).  Run on an MI355X box:  python tools/api_host_breakdown.py"""
import os, sys, time, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import argparse
import bench
from ngp_pl_amd.rendering import render

args = argparse.Namespace(rays=0, res=800, images=100)
dev = torch.device("cuda", 0)
loop = bench.Loop("lego", args, dev, 0, 1, None)
loop.steps(int(os.environ.get("SETUP", 600)))
tr = loop.trainer
torch.cuda.synchronize()


def run(prefetch, n=200):
    seg = {}
    def stamp(name, t0):
        t = time.perf_counter(); seg[name] = seg.get(name, 0.0) + (t - t0); return t
    cur = loop.draw(on_side=False)
    for i in range(n + 20):
        if i == 20:
            torch.cuda.synchronize(); seg.clear(); t_all = time.perf_counter()
        t = time.perf_counter()
        nxt = loop.draw(on_side=False); t = stamp("draw", t)
        tr._maybe_update_grid(); t = stamp("grid_update", t)
        kw = {"test_time": False}
        if prefetch and (tr.global_step + 1) % tr.update_interval != 0:
            kw["next_rays"] = (nxt[0], nxt[1])
        results = render(tr.model, cur[0], cur[1], **kw); t = stamp("render", t)
        loss_d = tr.loss_fn(results, {"rgb": cur[2]}); t = stamp("loss_terms", t)
        loss = sum(lo.mean() for lo in loss_d.values()); t = stamp("loss_mean_sum", t)
        loss.backward(); t = stamp("backward", t)
        tr.opt.step(); t = stamp("opt_step", t)
        tr.global_step += 1
        cur = nxt
    torch.cuda.synchronize()
    total = (time.perf_counter() - t_all) / n
    return {"ms_per_step": total * 1e3, "host_us": {k: round(v / n * 1e6, 1) for k, v in seg.items()}}


out = {"prefetch": run(True), "plain": run(False), "prefetch_again": run(True)}
print(json.dumps(out, indent=1))
