"""Scratch: stage times of the training step late in training (few live samples per ray): where does a late step spend its time?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pl_amd.bench_support import GpuDataset
from ngp_pl_amd.networks import NGP
from ngp_pl_amd.trainer import Trainer
steps = int(os.environ.get("STEPS", 8000))
torch.manual_seed(1337)
dev = torch.device("cuda")
model = NGP(0.5).to(dev); model.register_training_buffers()
tr = Trainer(model, lr=1e-2, num_epochs=30)
data = GpuDataset(800, 100, dev, seed=0)
cur = data.sample_native(8192, 0)
for i in range(steps):
    nxt = data.sample_native(8192, i + 1)
    tr.step(cur[0], cur[1], cur[2], next_batch=(nxt[0], nxt[1])); cur = nxt
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200):
    nxt = data.sample_native(8192, steps + i + 1)
    tr.step(cur[0], cur[1], cur[2], next_batch=(nxt[0], nxt[1])); cur = nxt
torch.cuda.synchronize()
print("step %d: %.4f ms/step; metrics %s" % (steps, (time.perf_counter() - t0) / 200 * 1e3, tr.metrics()))
acc = {}
tr.events = []
for i in range(32):
    nxt = data.sample_native(8192, steps + 300 + i)
    tr.step(cur[0], cur[1], cur[2], next_batch=(nxt[0], nxt[1])); cur = nxt
    for name, ms in tr.stage_times_ms():
        acc[name] = acc.get(name, 0.0) + ms / 32
B = tr._buf
n2 = int(B.view("two_round_counts", torch.int32, 4)[0].item())
tot = tr.last["total"].float()
ra = B.view("rays_a%d" % 0, torch.int64, 8192, 3)[:, 2].float()
print("two-round counters: rest-list length %d of S %d; rays with samples %d, rays that never stop (total == N > 0) %d, mean stop index of the others %.1f, N mean %.1f" % (
    n2, tr.last["rm_samples"], int((ra > 0).sum()), int(((tot == ra) & (ra > 0)).sum()), float(tot[(tot < ra)].mean()) if bool((tot < ra).any()) else -1, float(ra.mean())))
print("stages (ms):", {k: round(v, 4) for k, v in acc.items()}, "sum main", round(sum(v for k, v in acc.items() if "side" not in k), 4), "n_active", int(tr.last["n_active"].item()))
