"""Scratch: cProfile of the training loop's host side (optionally with a 1-rank process group: PG=1)."""
import cProfile, os, pstats, sys, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pl_amd.bench_support import GpuDataset
from ngp_pl_amd.ddp import GradientExchange
from ngp_pl_amd.networks import NGP
from ngp_pl_amd.trainer import Trainer
dev = torch.device("cuda", 0)
dist = None
if os.environ.get("PG"):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
torch.manual_seed(0)
model = NGP(0.5).to(dev); model.register_training_buffers()
tr = Trainer(model)
if dist is not None:
    GradientExchange(model, dist, 1).install(tr)
data = GpuDataset(800, 20, dev)
cur = data.sample_native(8192, 0)
for i in range(330):
    nxt = data.sample_native(8192, i + 1)
    tr.step(cur[0], cur[1], cur[2], next_batch=(nxt[0], nxt[1])); cur = nxt
torch.cuda.synchronize()
import time
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for i in range(200):
    nxt = data.sample_native(8192, 1000 + i)
    tr.step(cur[0], cur[1], cur[2], next_batch=(nxt[0], nxt[1])); cur = nxt
pr.disable()
torch.cuda.synchronize()
print("ms/step %.3f" % ((time.perf_counter() - t0) / 200 * 1e3))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30); print(s.getvalue()[:6000])
