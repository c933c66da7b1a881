"""Round 5: what the frame marcher's block hops are worth.  Trains the headline workload (20 000 steps), then renders 40 held-out
800x800 poses with the hops and -- ngp_debug_render_block_hops(0) -- cell by cell, alternating, in the reference's chunking and
regrouped; prints frames/s per setting and checks one frame is the same bits both ways.   Usage: frame_hops_ab.py [workload] [steps]"""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ngp_pl_amd import _lib, synthetic as syn  # noqa: E402
from ngp_pl_amd.bench_support import render_eval  # noqa: E402
from ngp_pl_amd.rendering import render  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "lego"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
args = types.SimpleNamespace(rays=0, res=800, images=100)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
loop = bench.Loop(wl, args, dev, 0, 1, None)
loop.trainer.steps_per_epoch = max(steps // 30, 1)
loop.steps(steps)
poses = syn.hemisphere_poses(40, seed=999).to(dev)
with torch.no_grad():
    ro, rd = syn.get_rays(loop.data.directions, poses[0])
    on = render(loop.model, ro, rd, test_time=True)
    _lib.call("ngp_debug_render_block_hops", 0)
    off = render(loop.model, ro, rd, test_time=True)
    same = all(torch.equal(on[k], off[k]) for k in ("rgb", "depth", "opacity")) and int(on["total_samples"]) == int(off["total_samples"])
    print("%s after %d steps, pose 0: same bits %s, samples %d, iterations %d / %d" % (wl, steps, same, int(on["total_samples"]), on["n_iterations"],
                                                                                     off["n_iterations"]), flush=True)
for rnd in range(3):
    for enabled in (1, 0):
        _lib.call("ngp_debug_render_block_hops", enabled)
        for name, kw in (("reference", {}), ("regrouped", dict(chunk_scale=4, probe_cap=64))):
            r = render_eval(loop.model, loop.data, poses, psnr=False, **kw)
            print("round %d  hops %s  %-10s %.1f fps  (%.3f ms, median %.3f, iterations %.1f)" % (
                rnd, "on " if enabled else "off", name, r["fps"], r["ms_per_frame"], r["ms_per_frame_median"], r.get("iterations_mean", 0)), flush=True)
_lib.call("ngp_debug_render_block_hops", 1)
