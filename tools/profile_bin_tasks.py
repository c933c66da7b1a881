"""Scratch: per-task phase times of the binned backward (needs a -DNGP_BIN_TIMING build via NGP_HIP_LIB)."""
import ctypes as C, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pl_amd import _lib
from ngp_pl_amd._lib import GridMeta, call, ptr, stream
dev = "cuda"
meta = GridMeta()
call("ngp_grid_meta_init", C.byref(meta), 16, 2, 19, 16, float(math.exp(math.log(2048 * 0.5 / 16) / 15)))
S = 170000; R = S // 20
o = torch.rand(R, 1, 3, device=dev) - 0.5
d = torch.randn(R, 1, 3, device=dev); d = d / d.norm(dim=-1, keepdim=True)
t = torch.arange(20, device=dev).view(1, 20, 1) * 1.7e-3
x = ((o * 0.6 + d * t).clamp(-0.5, 0.5)).reshape(-1, 3).contiguous()
mn = torch.full((3,), -0.5, device=dev); mx = torch.full((3,), 0.5, device=dev)
g16 = torch.zeros(meta.offset[16], 2, dtype=torch.half, device=dev)
dfe = (torch.randn(16, S, 2, device=dev) * 1e-2).half()
nb = _lib.lib().ngp_hashgrid_bwd_binned_workspace_bytes(C.byref(meta), S)
ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
for _ in range(3):
    call("ngp_hashgrid_bwd_binned", ptr(x), ptr(mn), ptr(mx), ptr(dfe), C.byref(meta), S, None, None, ptr(ws), nb, ptr(g16), stream())
torch.cuda.synchronize()
tm = ws[256:256 + 65536].view(torch.int64).view(-1, 4).cpu().double()
tm = tm[tm[:, 0] > 0]
t0 = tm[:, 0].min()
print("tasks", len(tm), "span %.1f us (100 MHz clock)" % ((tm[:, 3].max() - t0) / 100))
nd = len(tm) - 760
for name, sel in (("dense tasks", tm[:nd]), ("hashed tasks", tm[nd:])):
    pro = (sel[:, 1] - sel[:, 0]).mean() / 100; scan = (sel[:, 2] - sel[:, 1]).mean() / 100; wr = (sel[:, 3] - sel[:, 2]).mean() / 100
    print("%s: n=%d prologue %.1f us, scan %.1f us (max %.1f), write-out %.1f us" % (name, len(sel), pro, scan, (sel[:, 2] - sel[:, 1]).max() / 100, wr))
scan = (tm[:, 2] - tm[:, 1]) / 100
bounds = [("L0", 0, 16), ("L1", 16, 32), ("L2", 32, 48), ("L3", 48, 80), ("L4", 80, 152), ("L5", 152, 312)]
for name, a, b in bounds:
    sel = scan[a:b]
    print("%s: %d tasks, scan mean %.1f max %.1f us, start of first %.1f us" % (name, b - a, sel.mean(), sel.max(), (tm[a, 0] - t0) / 100))
# how much of the launch is tail: the sum of the tasks' durations over 256 workgroups against the span
dur = (tm[:, 3] - tm[:, 0]) / 100
end = (tm[:, 3] - t0) / 100
print("sum of task durations %.1f us x WG -> %.1f us on 256 workgroups; span %.1f us; tasks ending in the last 10 us: %d, last 20 us: %d; hashed task %.1f us, dense task %.1f us (max %.1f)" % (
    float(dur.sum()), float(dur.sum()) / 256, float(end.max()), int((end > end.max() - 10).sum()), int((end > end.max() - 20).sum()),
    float(dur[nd:].mean()), float(dur[:nd].mean()), float(dur[:nd].max())))
