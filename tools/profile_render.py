"""Scratch: train briefly, then time render(test_time=True) frames (for rocprofv3 kernel traces of the test path)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pl_amd import synthetic as syn
from ngp_pl_amd.bench_support import GpuDataset, render_fps
from ngp_pl_amd.networks import NGP
from ngp_pl_amd.trainer import Trainer
torch.manual_seed(0)
dev = torch.device("cuda")
model = NGP(0.5).to(dev); model.register_training_buffers()
tr = Trainer(model)
data = GpuDataset(800, 20, dev)
for i in range(400):
    b = data.sample_native(8192, i)
    tr.step(*b)
torch.cuda.synchronize()
print(render_fps(model, data, n_frames=int(os.environ.get("FRAMES", 5))))
