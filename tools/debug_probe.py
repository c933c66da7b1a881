"""Scratch: the probe kernel against the composite's own stop statistics on a trained field."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pl_amd import _lib
from ngp_pl_amd._lib import call, ptr, stream
from ngp_pl_amd.bench_support import GpuDataset
from ngp_pl_amd.networks import NGP
from ngp_pl_amd.trainer import Trainer
os.environ["NGP_TWO_ROUND"] = "off"
torch.manual_seed(1337)
dev = torch.device("cuda")
model = NGP(0.5).to(dev); model.register_training_buffers()
tr = Trainer(model)
data = GpuDataset(800, 100, dev, seed=0)
cur = data.sample_native(8192, 0)
for i in range(4000):
    nxt = data.sample_native(8192, i + 1)
    tr.step(cur[0], cur[1], cur[2], next_batch=(nxt[0], nxt[1])); cur = nxt
torch.cuda.synchronize()
B = tr._buf
k = call("ngp_stepper_last_set", tr._stepper)
S = tr.last["rm_samples"]
rays_a = B.view("rays_a%d" % k, torch.int64, 8192, 3).clone()
sig = B.view("sigmas", torch.float32, S).clone(); dl = B.view("deltas", torch.float32, S).clone()
total = tr.last["total"].clone()
N = rays_a[:, 2]
print("S", S, "sum N", int(N.sum()), "n_active", int(tr.last["n_active"].item()), "rays stopping", int((total < N).sum()), "never", int(((total == N) & (N > 0)).sum()))
for K in (4, 8, 16):
    lst = torch.full((S,), -7, dtype=torch.int32, device=dev); cnt = torch.zeros(4, dtype=torch.int32, device=dev)
    call("ngp_composite_probe", ptr(sig), ptr(dl), ptr(rays_a), K, 1e-4, 8192, ptr(lst), ptr(cnt), stream())
    torch.cuda.synchronize()
    # expectation from the composite's totals: a ray continues iff N > K and its stop lies beyond the first K samples (total >= K)
    cont = (N > K) & (total >= K)
    print("K=%d: probe lists %d samples; expected %d (rays continuing %d)" % (K, int(cnt[0]), int((N - K)[cont].sum()), int(cont.sum())))
