"""Round 5 diagnostics: where a launch of the frame loop's thread-per-ray marcher spends its time.  Needs a -DNGP_RENDER_TIMING build
(tools/build_variant.sh render_timing march.hip -DNGP_RENDER_TIMING; NGP_HIP_LIB selects it): every wave records its wall clock at
entry and exit and its longest / summed probe counts.  Trains the headline workload, renders one held-out 800x800 pose, prints per
iteration: waves, launch span, wave duration percentiles, when the waves started, probes per wave, the five longest waves, and what the longest lanes were doing (cell-sized hops, longer hops, samples)."""
import ctypes as C
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ngp_pl_amd import _lib, synthetic as syn  # noqa: E402
from ngp_pl_amd.rendering import render  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
args = types.SimpleNamespace(rays=0, res=800, images=100)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
loop = bench.Loop("lego", args, dev, 0, 1, None)
loop.trainer.steps_per_epoch = max(steps // 30, 1)
loop.steps(steps)
L = _lib.lib()
L.ngp_debug_render_timing_read.restype = C.c_int
L.ngp_debug_render_timing_read.argtypes = [C.c_void_p, C.c_int, C.c_int]
poses = syn.hemisphere_poses(3, seed=999).to(dev)
rows = np.zeros((1 << 18, 6), np.uint64)
with torch.no_grad():
    for i in range(3):
        ro, rd = syn.get_rays(loop.data.directions, poses[i])
        torch.cuda.synchronize()
        L.ngp_debug_render_timing_read(rows.ctypes.data, 0, 1)
        out = render(loop.model, ro, rd, test_time=True)
        torch.cuda.synchronize()
n = L.ngp_debug_render_timing_read(rows.ctypes.data, rows.shape[0], 1)
r = rows[:n]
t0, t1 = r[:, 0].astype(np.int64), r[:, 1].astype(np.int64)
mx, sm = (r[:, 2] & 0xffffffff).astype(np.int64), (r[:, 2] >> 32).astype(np.int64)
N, alive = (r[:, 3] & 0xffffffff).astype(np.int64), (r[:, 3] >> 32).astype(np.int64)
h_short, h_long, h_s = (r[:, 4] & 0xffff).astype(np.int64), ((r[:, 4] >> 16) & 0xffff).astype(np.int64), (r[:, 4] >> 32).astype(np.int64)
print("last frame: %d wave records, %d iterations in the frame" % (n, out["n_iterations"]))
us = 0.01                                              # wall_clock64: 100 MHz
for key in sorted(set(zip(alive.tolist(), N.tolist())), reverse=True):
    sel = (alive == key[0]) & (N == key[1])
    a, b = t0[sel], t1[sel]
    d = (b - a) * us
    o = np.argsort(-d)[:5]
    print("n_alive %7d N %2d: %5d waves, span %6.1f us, starts within %5.1f us, wave us p50 %5.1f p90 %5.1f p99 %5.1f max %5.1f | probes of the wave's longest lane "
          "p50 %3d p90 %3d p99 %3d max %4d, mean per lane %.1f | longest waves (us, longest lane's probes): %s" % (
              key[0], key[1], sel.sum(), (b.max() - a.min()) * us, (a.max() - a.min()) * us, *np.percentile(d, [50, 90, 99]), d.max(),
              *np.percentile(mx[sel], [50, 90, 99]).astype(int), mx[sel].max(), sm[sel].mean() / 64,
              " ".join("(%.0f,%d)" % (d[j], mx[sel][j]) for j in o)))
    slow = d >= np.percentile(d, 95)
    print("        the longest lanes of the slowest 5 %% of the waves: %.1f cell-sized hops, %.1f longer hops, %.1f samples on average; of all waves: %.1f / %.1f / %.1f" % (
        h_short[sel][slow].mean(), h_long[sel][slow].mean(), h_s[sel][slow].mean(), h_short[sel].mean(), h_long[sel].mean(), h_s[sel].mean()))
