"""Scratch: where do rays stop?  After N training steps: per-ray marched (N_r) and composited counts -> how many samples a two-round
forward (first K of every ray, then the rest of the rays still transparent) would have to evaluate."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pl_amd.bench_support import GpuDataset
from ngp_pl_amd.networks import NGP
from ngp_pl_amd.rendering import render
from ngp_pl_amd.trainer import Trainer
torch.manual_seed(1337)
dev = torch.device("cuda")
model = NGP(0.5).to(dev); model.register_training_buffers()
tr = Trainer(model)
data = GpuDataset(800, 100, dev)
for target in (540, 2000, 10000):
    while tr.global_step < target:
        b = data.sample_native(8192, tr.global_step); tr.step(*b)
    b = data.sample_native(8192, 10 ** 6 + target)
    with torch.no_grad():
        model.fused_render = False
        res = render(model, b[0], b[1])
        model.fused_render = True
    N = res["rays_a"][:, 2]
    # composited count per ray: ws > 0 prefix length (+1 for the sample that triggers the stop)
    ws = res["ws"]; rays_a = res["rays_a"]
    owner = torch.repeat_interleave(torch.arange(N.numel(), device=dev), N)
    live = torch.zeros(N.numel(), device=dev).index_add_(0, owner, (ws > 0).float()).long()
    S = int(N.sum())
    line = "step %5d: S %d, live (ws > 0) %d (%.2f)" % (target, S, int(live.sum()), float(live.sum()) / S)
    for K in (4, 8, 16, 24):
        first = torch.minimum(N, torch.tensor(K, device=dev))
        cont = (live > K) | ((live == N) & (N > K))      # not stopped within the first K
        need = int(first.sum() + ((N - K).clamp(min=0) * cont).sum())
        line += " | K=%d: %.2f" % (K, need / S)
    print(line)
