"""Scratch: train briefly, then time render(test_time=True) frames under several loop settings
(for rocprofv3 kernel traces of the test path set CONFIGS=one of the names)."""
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
for i in range(int(os.environ.get("STEPS", 400))):
    b = data.sample_native(8192, i)
    tr.step(*b)
torch.cuda.synchronize()
configs = {
    "host_loop": dict(host_loop=True),
    "device_exact": dict(),
    "cap32": dict(probe_cap=32),
    "cap64": dict(probe_cap=64),
    "k2": dict(chunk_scale=2),
    "k2_cap32": dict(chunk_scale=2, probe_cap=32),
    "k4_cap32": dict(chunk_scale=4, probe_cap=32),
    "k4_cap64": dict(chunk_scale=4, probe_cap=64),
    "k8_cap64": dict(chunk_scale=8, probe_cap=64),
}
want = os.environ.get("CONFIGS")
for name, kw in configs.items():
    if want and name not in want.split(","):
        continue
    print(name, render_fps(model, data, n_frames=int(os.environ.get("FRAMES", 5)), **kw), flush=True)
