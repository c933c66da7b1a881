"""Round 5: where the frame loop should switch from a thread per ray to a wave per ray (ngp_debug_render_wave_rays).  Trains the
headline workload, then renders 40 held-out 800x800 poses in the reference's chunking for several crossover ray counts, twice;
prints frames/s per setting and checks one frame is the same bits at every setting.   Usage: frame_wave_ab.py [workload] [steps]"""
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
limits = (0, 2048, 16384, 80000, 200000, 700000)
with torch.no_grad():
    ro, rd = syn.get_rays(loop.data.directions, poses[0])
    ref = None
    for lim in limits:
        _lib.call("ngp_debug_render_wave_rays", lim)
        out = render(loop.model, ro, rd, test_time=True)
        if ref is None:
            ref = out
        same = all(torch.equal(out[k], ref[k]) for k in ("rgb", "depth", "opacity")) and int(out["total_samples"]) == int(ref["total_samples"])
        print("%s after %d steps, pose 0, crossover %d: same bits as crossover 0: %s (samples %d, iterations %d)" % (
            wl, steps, lim, same, int(out["total_samples"]), out["n_iterations"]), flush=True)
for rnd in range(2):
    for lim in limits:
        _lib.call("ngp_debug_render_wave_rays", lim)
        r = render_eval(loop.model, loop.data, poses, psnr=False)
        print("round %d  crossover %7d rays  %.1f fps  (%.3f ms, median %.3f)" % (rnd, lim, r["fps"], r["ms_per_frame"], r["ms_per_frame_median"]), flush=True)
_lib.call("ngp_debug_render_wave_rays", -1)
