import os, sys, time, torch
sys.path.insert(0, "/root/repo")
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29544")
import torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
from ngp_pl_amd.bench_support import GpuDataset
from ngp_pl_amd.ddp import GradientExchange
from ngp_pl_amd.networks import NGP
from ngp_pl_amd.trainer import Trainer
torch.manual_seed(0)
model = NGP(0.5).to(dev); model.register_training_buffers()
tr = Trainer(model)
ex = GradientExchange(model, dist, 1).install(tr)
acc = {}
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t; return r
    return w
tr.mlp_grad_hook = timed("reduce_mlp", ex.reduce_mlp); tr.grad_hook = timed("reduce_grid", ex.reduce_grid); tr.group_hook = timed("reduce_piece", ex.reduce_piece)
orig_ar = dist.all_reduce
dist.all_reduce = timed("all_reduce", orig_ar); ex.dist = dist
data = GpuDataset(800, 20, dev)
cur = data.sample_native(8192, 0)
for i in range(330):
    nxt = data.sample_native(8192, i + 1); tr.step(cur[0], cur[1], cur[2], next_batch=(nxt[0], nxt[1])); cur = nxt
torch.cuda.synchronize(); acc.clear()
t0 = time.perf_counter()
for i in range(200):
    nxt = data.sample_native(8192, 1000 + i); tr.step(cur[0], cur[1], cur[2], next_batch=(nxt[0], nxt[1])); cur = nxt
torch.cuda.synchronize()
print("ms/step %.3f" % ((time.perf_counter() - t0) / 200 * 1e3))
print({k: "%.1f us/step" % (v / 200 * 1e6) for k, v in acc.items()})
dist.destroy_process_group()
