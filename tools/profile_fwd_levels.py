"""Scratch: time ngp_hashgrid_fwd level by level (one-level GridMeta) on ray-coherent samples."""
import ctypes as C, math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pl_amd import _lib
from ngp_pl_amd._lib import GridMeta, call, ptr, stream
dev = "cuda"
torch.manual_seed(0)
meta = GridMeta()
call("ngp_grid_meta_init", C.byref(meta), 16, 2, 19, 16, float(math.exp(math.log(2048 * 0.5 / 16) / 15)))
S = int(sys.argv[1]) if len(sys.argv) > 1 else 305000; R = S // 38
o = torch.rand(R, 1, 3, device=dev) - 0.5
d = torch.randn(R, 1, 3, device=dev); d = d / d.norm(dim=-1, keepdim=True)
t = torch.arange(38, device=dev).view(1, 38, 1) * 1.7e-3
x = ((o * 0.6 + d * t).clamp(-0.5, 0.5)).reshape(-1, 3).contiguous(); S = x.shape[0]
mn = torch.full((3,), -0.5, device=dev); mx = torch.full((3,), 0.5, device=dev)
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6
table = ((torch.rand(meta.offset[16], 2, device=dev) - 0.5)).half()
feats = torch.empty(16, S, 2, dtype=torch.half, device=dev)
def bits(t):
    return int(t.view(torch.int16).to(torch.int64).sum().item()) ^ int((t.view(torch.int16).to(torch.int64) * torch.arange(t.numel(), device=dev).view(t.shape) % 1000003).sum().item())
print("NGP_FWD_REUSE_MAX_RES=%s" % os.environ.get("NGP_FWD_REUSE_MAX_RES", "(unset)"))
print("all 16 levels, S=%d: %.1f us" % (S, bench(lambda: call("ngp_hashgrid_fwd", ptr(x), ptr(mn), ptr(mx), ptr(table), C.byref(meta), S, ptr(feats), stream()))))
if len(sys.argv) > 2:            # device-sized launch: the bound is argv[2] x the real count
    bound = int(S * float(sys.argv[2]))
    xb = torch.zeros(bound, 3, device=dev); xb[:S] = x
    fb = torch.empty(16, bound, 2, dtype=torch.half, device=dev)
    n_dev = torch.tensor([S], dtype=torch.int32, device=dev)
    print("all 16 levels, device count %d of bound %d: %.1f us" % (S, bound, bench(lambda: call("ngp_hashgrid_fwd_n", ptr(xb), ptr(mn), ptr(mx), ptr(table), C.byref(meta), bound, ptr(n_dev), ptr(fb), stream()))))
    sys.exit(0)
for l in range(16):
    m1 = GridMeta(); m1.n_levels = 1; m1.n_features = 2; m1.offset[0] = 0
    for k in range(1, 17): m1.offset[k] = meta.offset[l + 1] - meta.offset[l]
    m1.resolution[0] = meta.resolution[l]; m1.scale[0] = meta.scale[l]
    tl = table[meta.offset[l]:meta.offset[l + 1]].contiguous()
    us = bench(lambda: call("ngp_hashgrid_fwd", ptr(x), ptr(mn), ptr(mx), ptr(tl), C.byref(m1), S, ptr(feats), stream()))
    print("level %2d res %4d: %6.1f us   bits %d" % (l, meta.resolution[l], us, bits(feats[0])))
