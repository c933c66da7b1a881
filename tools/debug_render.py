import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_train_gpu import make_model, batch
from ngp_pl_amd.rendering import render
from ngp_pl_amd.trainer import Trainer
m = make_model(seed=5)
tr = Trainer(m)
bs = [batch(4096, seed=300 + i) for i in range(4)]
for it in range(150):
    tr.step(*bs[it % 4])
ro, rd, _ = batch(30000, seed=78)
host = render(m, ro, rd, test_time=True, host_loop=True)
for kw in (dict(), dict(chunk_scale=4), dict(probe_cap=16), dict(chunk_scale=3, probe_cap=48), dict(chunk_scale=2)):
    fast = render(m, ro, rd, test_time=True, **kw)
    d = (fast["rgb"] - host["rgb"]).abs().max(dim=1).values
    bad = (d > 1e-5).nonzero().flatten()
    print(kw, "iters", fast["n_iterations"], "total", int(fast["total_samples"]), int(host["total_samples"]), "nbad", len(bad), "max", float(d.max()))
    for r in bad[:5].tolist():
        print("   ray", r, "host rgb", host["rgb"][r].tolist(), "fast", fast["rgb"][r].tolist(), "op", float(host["opacity"][r]), float(fast["opacity"][r]),
              "depth", float(host["depth"][r]), float(fast["depth"][r]))
