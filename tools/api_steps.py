"""Scratch: N steps through the reference-shaped surface (Trainer.step_autograd) for kernel traces."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pl_amd.bench_support import GpuDataset
from ngp_pl_amd.networks import NGP
from ngp_pl_amd.trainer import Trainer
torch.manual_seed(0)
dev = torch.device("cuda")
model = NGP(0.5).to(dev); model.register_training_buffers()
tr = Trainer(model)
data = GpuDataset(800, 20, dev)
cur = data.sample_native(8192, 0)
for i in range(400):
    nxt = data.sample_native(8192, i + 1); tr.step(cur[0], cur[1], cur[2], next_batch=(nxt[0], nxt[1])); cur = nxt
tr._drop_pending()
torch.cuda.synchronize()
for i in range(20):
    b = data.sample_native(8192, 5000 + i); tr.step_autograd(b[0], b[1], b[2])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(100):
    b = data.sample_native(8192, 6000 + i); tr.step_autograd(b[0], b[1], b[2])
torch.cuda.synchronize()
print("api ms/step %.3f" % ((time.perf_counter() - t0) / 100 * 1e3))
