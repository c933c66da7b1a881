"""Scratch GPU micro-benchmarks used to steer kernel design (not part of the product or tests)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ngp_pl_amd import _lib
from ngp_pl_amd._lib import call, ptr, stream, GridMeta
import ctypes as C

torch.manual_seed(0)
dev = "cuda"
meta = GridMeta()
call("ngp_grid_meta_init", C.byref(meta), 16, 2, 19, 16, float(1.3195079107728942))
total = meta.offset[16]
print("entries", total)
table = (torch.rand(total, 2, device=dev) * 2e-4 - 1e-4).half()
mn = torch.full((3,), -0.5, device=dev); mx = torch.full((3,), 0.5, device=dev)


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e6


for S in (303000,):
    # ray-coherent samples: 8192-ish rays x consecutive steps
    R = S // 37
    S = R * 37
    o = (torch.rand(R, 1, 3, device=dev) - 0.5) * 0.6
    d = torch.randn(R, 1, 3, device=dev); d = d / d.norm(dim=-1, keepdim=True)
    t = torch.arange(37, device=dev).view(1, 37, 1) * 1.7e-3
    x = ((o + d * t).clamp(-0.5, 0.5)).reshape(-1, 3).contiguous()
    xr = (torch.rand(S, 3, device=dev) - 0.5)
    feats = torch.empty(16, S, 2, dtype=torch.half, device=dev)
    dfe = (torch.randn(16, S, 2, device=dev) * 1e-2).half()
    g16 = torch.zeros(total, 2, dtype=torch.half, device=dev)
    g32 = torch.zeros(total, 2, dtype=torch.float32, device=dev)
    for name, xx in (("coherent", x), ("random", xr)):
        f = bench(lambda: call("ngp_hashgrid_fwd", ptr(xx), ptr(mn), ptr(mx), ptr(table), C.byref(meta), S, ptr(feats), stream()))
        b16 = b32 = 1.0
        print("S=%8d %-8s fwd %8.1f us (%.1f Ggather/s)  bwd f16 %8.1f us (%.1f Gatom/s)  bwd f32 %8.1f us" % (
            S, name, f, S * 128 / f / 1e3, b16, S * 128 / b16 / 1e3, b32))
