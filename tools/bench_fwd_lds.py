"""Scratch: hash-grid forward with the coarse levels' tables resident in LDS (ngp_hashgrid_fwd_lds) against the L2 path
(ngp_hashgrid_fwd), on ray-coherent samples: time of levels 0-2 alone either way, of levels 3-15 alone, of all 16, and the
bit-for-bit check.  rocprofv3 --pmc TCC_REQ_sum over this script gives the L2 request counts quoted in profiles/."""
import ctypes as C, math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pl_amd import _lib
from ngp_pl_amd._lib import GridMeta, call, ptr, stream
dev = "cuda"
meta = GridMeta()
call("ngp_grid_meta_init", C.byref(meta), 16, 2, 19, 16, float(math.exp(math.log(2048 * 0.5 / 16) / 15)))
S = int(sys.argv[1]) if len(sys.argv) > 1 else 330000
P = 40; R = S // P
torch.manual_seed(0)
o = torch.rand(R, 1, 3, device=dev) - 0.5
d = torch.randn(R, 1, 3, device=dev); d = d / d.norm(dim=-1, keepdim=True)
t = torch.arange(P, device=dev).view(1, P, 1) * 1.7e-3
x = ((o * 0.6 + d * t).clamp(-0.5, 0.5)).reshape(-1, 3).contiguous(); S = x.shape[0]
mn = torch.full((3,), -0.5, device=dev); mx = torch.full((3,), 0.5, device=dev)
table = ((torch.rand(meta.offset[16], 2, device=dev) - 0.5)).half()


def sub_meta(l0, l1):
    m = GridMeta(); m.n_levels = l1 - l0; m.n_features = 2
    for k in range(17):
        m.offset[k] = meta.offset[min(l0 + k, l1)] - meta.offset[l0]
    for k in range(l1 - l0):
        m.resolution[k] = meta.resolution[l0 + k]; m.scale[k] = meta.scale[l0 + k]
    return m, table[meta.offset[l0]:meta.offset[l1]].contiguous()


def bench(fn, iters=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


feats = torch.zeros(16, S, 2, dtype=torch.half, device=dev)
ref = torch.zeros_like(feats)
call("ngp_hashgrid_fwd", ptr(x), ptr(mn), ptr(mx), ptr(table), C.byref(meta), S, ptr(ref), stream())
call("ngp_hashgrid_fwd_lds", ptr(x), ptr(mn), ptr(mx), ptr(table), C.byref(meta), 3, S, ptr(feats), stream())
torch.cuda.synchronize()
print("bit-identical levels 0-2:", bool(torch.equal(feats[:3], ref[:3])))
m03, t03 = sub_meta(0, 3); m316, t316 = sub_meta(3, 16)
print("S = %d ray-coherent samples" % S)
print("all 16 levels, L2 path              : %6.1f us" % bench(lambda: call("ngp_hashgrid_fwd", ptr(x), ptr(mn), ptr(mx), ptr(table), C.byref(meta), S, ptr(ref), stream())))
print("levels 0-2 alone, L2 path           : %6.1f us" % bench(lambda: call("ngp_hashgrid_fwd", ptr(x), ptr(mn), ptr(mx), ptr(t03), C.byref(m03), S, ptr(ref), stream())))
print("levels 0-2 alone, tables in LDS     : %6.1f us" % bench(lambda: call("ngp_hashgrid_fwd_lds", ptr(x), ptr(mn), ptr(mx), ptr(table), C.byref(meta), 3, S, ptr(feats), stream())))
print("levels 3-15 alone, L2 path          : %6.1f us" % bench(lambda: call("ngp_hashgrid_fwd", ptr(x), ptr(mn), ptr(mx), ptr(t316), C.byref(m316), S, ptr(ref), stream())))


def split():
    call("ngp_hashgrid_fwd_lds", ptr(x), ptr(mn), ptr(mx), ptr(table), C.byref(meta), 3, S, ptr(feats), stream())
    call("ngp_hashgrid_fwd", ptr(x), ptr(mn), ptr(mx), ptr(t316), C.byref(m316), S, ptr(ref), stream())


print("LDS kernel (0-2) + L2 kernel (3-15) : %6.1f us back to back" % bench(split))
