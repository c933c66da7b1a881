"""Scratch: the binned table backward alone (bin + apply + merge) on ray-coherent samples, HIP-event time over N launches.
NGP_HIP_LIB selects an A/B build (tools/build_variant.sh); a -DNGP_BIN_TIMING build also prints per-task phase times.
  python tools/bench_bwd.py [n_samples]"""
import ctypes as C, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pl_amd import _lib
from ngp_pl_amd._lib import GridMeta, call, ptr, stream
dev = "cuda"
meta = GridMeta()
call("ngp_grid_meta_init", C.byref(meta), 16, 2, 19, 16, float(math.exp(math.log(2048 * 0.5 / 16) / 15)))
S = int(sys.argv[1]) if len(sys.argv) > 1 else 155000
torch.manual_seed(0)
R = S // 20
o = torch.rand(R, 1, 3, device=dev) - 0.5
d = torch.randn(R, 1, 3, device=dev); d = d / d.norm(dim=-1, keepdim=True)
t = torch.arange(20, device=dev).view(1, 20, 1) * 1.7e-3
x = ((o * 0.6 + d * t).clamp(-0.5, 0.5)).reshape(-1, 3).contiguous()
S = x.shape[0]
mn = torch.full((3,), -0.5, device=dev); mx = torch.full((3,), 0.5, device=dev)
g16 = torch.zeros(meta.offset[16], 2, dtype=torch.half, device=dev)
dfe = (torch.randn(16, S, 2, device=dev) * 1e-2).half()
S_plan = int(os.environ.get("S_PLAN", 2 * S))          # the trainer plans for the marched count and processes the active ones
n_active = torch.tensor([S], dtype=torch.int32, device=dev)
nb = _lib.lib().ngp_hashgrid_bwd_binned_workspace_bytes(C.byref(meta), S_plan)
ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
dfe_p = torch.zeros(16, S_plan, 2, dtype=torch.half, device=dev); dfe_p[:, :S] = dfe
xp = torch.zeros(S_plan, 3, device=dev); xp[:S] = x


def run():
    call("ngp_hashgrid_bwd_binned", ptr(xp), ptr(mn), ptr(mx), ptr(dfe_p), C.byref(meta), S_plan, None, ptr(n_active), ptr(ws), nb, ptr(g16), stream())


for _ in range(5):
    run()
torch.cuda.synchronize()
ref = g16.clone()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 50
e0.record()
for _ in range(N):
    run()
e1.record(); torch.cuda.synchronize()
print("lib %s: %d active samples (plan %d): %.1f us per launch (bin + apply + merge); checksum %.6f; run-to-run identical: %s" % (
    os.path.basename(os.environ.get("NGP_HIP_LIB", "libngp_hip.so")), S, S_plan, e0.elapsed_time(e1) / N * 1e3, float(g16.float().abs().sum()),
    bool(torch.equal(ref, g16))))
tm = ws[256:256 + 65536].view(torch.int64).view(-1, 4).cpu().double()
tm = tm[tm[:, 0] > 0]
if len(tm) > 100:
    t0 = tm[:, 0].min()
    print("  tasks %d, apply span %.1f us" % (len(tm), (tm[:, 3].max() - t0) / 100))
    pro = (tm[:, 1] - tm[:, 0]).mean() / 100; scan = (tm[:, 2] - tm[:, 1]).mean() / 100; wr = (tm[:, 3] - tm[:, 2]).mean() / 100
    print("  per task: prologue %.2f us, segments %.2f us (max %.1f), write-out %.2f us" % (pro, scan, (tm[:, 2] - tm[:, 1]).max() / 100, wr))
    # marks: [0] task top, [1] after the decode (old kernel) / wave 0's rows done (pipelined kernel), [2] behind barrier A, [3] behind barrier B
    nd = len(tm) - 760
    if nd > 0 and len(tm) == 1072:
        for lv in range(10):                                    # hashed levels 6..15: 76 slice tasks each, in level order
            sel = tm[nd + 76 * lv: nd + 76 * (lv + 1)]
            d = (sel[:, 1:] - sel[:, :-1]) / 100
            print("    level %2d: scan %.2f us (max %.1f), write-out %.2f us" % (6 + lv, (d[:, 0] + d[:, 1]).mean(), (d[:, 0] + d[:, 1]).max(), d[:, 2].mean()))
    for name, sel in (("dense", tm[:nd]), ("hashed", tm[nd:])):
        d = (sel[:, 1:] - sel[:, :-1]) / 100
        print("  %s tasks (%d): top -> mark1 %.2f us, mark1 -> barrier A %.2f us, A -> barrier B %.2f us; task %.2f us (max %.1f)" % (
            name, len(sel), d[:, 0].mean(), d[:, 1].mean(), d[:, 2].mean(), d.sum(1).mean(), d.sum(1).max()))
