"""Scratch: fixed costs of the sliced backward: zero gradients (no updates at all) vs real ones, one hashed level."""
import ctypes as C, math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pl_amd import _lib
from ngp_pl_amd._lib import GridMeta, call, ptr, stream
dev = "cuda"
meta = GridMeta()
call("ngp_grid_meta_init", C.byref(meta), 16, 2, 19, 16, float(math.exp(math.log(2048 * 0.5 / 16) / 15)))
mn = torch.full((3,), -0.5, device=dev); mx = torch.full((3,), 0.5, device=dev)
def bench(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6
l = 12
m1 = GridMeta(); m1.n_levels = 1; m1.n_features = 2; m1.offset[0] = 0
for k in range(1, 17): m1.offset[k] = meta.offset[l + 1] - meta.offset[l]
m1.resolution[0] = meta.resolution[l]; m1.scale[0] = meta.scale[l]
g16 = torch.zeros(m1.offset[1], 2, dtype=torch.half, device=dev)
for S in (1000, 20000, 85000, 170000, 340000):
    R = S // 20
    o = torch.rand(R, 1, 3, device=dev) - 0.5
    d = torch.randn(R, 1, 3, device=dev); d = d / d.norm(dim=-1, keepdim=True)
    t = torch.arange(20, device=dev).view(1, 20, 1) * 1.7e-3
    x = ((o * 0.6 + d * t).clamp(-0.5, 0.5)).reshape(-1, 3).contiguous()
    S = x.shape[0]
    nb = _lib.lib().ngp_hashgrid_bwd_binned_workspace_bytes(C.byref(m1), S)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    for name, dfe in (("zero grads", torch.zeros(1, S, 2, device=dev).half()), ("real grads", (torch.randn(1, S, 2, device=dev) * 1e-2).half())):
        a = bench(lambda: call("ngp_hashgrid_bwd_sliced", ptr(x), ptr(mn), ptr(mx), ptr(dfe), C.byref(m1), S, None, None, ptr(g16), stream()))
        b = bench(lambda: call("ngp_hashgrid_bwd_binned", ptr(x), ptr(mn), ptr(mx), ptr(dfe), C.byref(m1), S, None, None, ptr(ws), nb, ptr(g16), stream()))
        print("S=%6d %s: full scan %7.1f us   binned %7.1f us" % (S, name, a, b))
    cnt = ws[:19 * 4].view(torch.int32)
    print("   list lengths:", cnt.tolist())
