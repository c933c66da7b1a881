"""Scratch: time ngp_hashgrid_bwd_sliced level by level (one-level GridMeta) and the forward, on ray-coherent samples."""
import ctypes as C
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pl_amd import _lib
from ngp_pl_amd._lib import GridMeta, call, ptr, stream

dev = "cuda"
meta = GridMeta()
call("ngp_grid_meta_init", C.byref(meta), 16, 2, 19, 16, float(math.exp(math.log(2048 * 0.5 / 16) / 15)))
S = 170000
R = S // 20
o = torch.rand(R, 1, 3, device=dev) - 0.5
d = torch.randn(R, 1, 3, device=dev); d = d / d.norm(dim=-1, keepdim=True)
t = torch.arange(20, device=dev).view(1, 20, 1) * 1.7e-3
x = ((o * 0.6 + d * t).clamp(-0.5, 0.5)).reshape(-1, 3).contiguous()
mn = torch.full((3,), -0.5, device=dev); mx = torch.full((3,), 0.5, device=dev)


def bench(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


total = meta.offset[16]
g16 = torch.zeros(total, 2, dtype=torch.half, device=dev)
dfe = (torch.randn(16, S, 2, device=dev) * 1e-2).half()
nb = _lib.lib().ngp_hashgrid_bwd_binned_workspace_bytes(C.byref(meta), S)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
print("binned, all levels: %.1f us (workspace %.0f MB)" % (bench(lambda: call("ngp_hashgrid_bwd_binned", ptr(x), ptr(mn), ptr(mx), ptr(dfe), C.byref(meta), S, None, None, ptr(ws), nb, ptr(g16), stream())), nb / 1e6))
print("all levels: %.1f us" % bench(lambda: call("ngp_hashgrid_bwd_sliced", ptr(x), ptr(mn), ptr(mx), ptr(dfe), C.byref(meta), S, None, None, ptr(g16), stream())))
for l in range(16):
    m1 = GridMeta()
    m1.n_levels = 1; m1.n_features = 2
    m1.offset[0] = 0
    for k in range(1, 17):
        m1.offset[k] = meta.offset[l + 1] - meta.offset[l]
    m1.resolution[0] = meta.resolution[l]; m1.scale[0] = meta.scale[l]
    d1 = dfe[l].contiguous()
    us = bench(lambda: call("ngp_hashgrid_bwd_sliced", ptr(x), ptr(mn), ptr(mx), ptr(d1), C.byref(m1), S, None, None, ptr(g16), stream()))
    nb1 = _lib.lib().ngp_hashgrid_bwd_binned_workspace_bytes(C.byref(m1), S)
    usb = bench(lambda: call("ngp_hashgrid_bwd_binned", ptr(x), ptr(mn), ptr(mx), ptr(d1), C.byref(m1), S, None, None, ptr(ws), max(nb1, 1), ptr(g16), stream())) if nb1 <= nb else float("nan")
    print("level %2d res %4d size %7d: %8.1f us   binned (bin + apply) %8.1f us" % (l, meta.resolution[l], m1.offset[1], us, usb))
