"""Full-length training run on the procedural Lego-like scene: 30 epochs x 1000 steps x 8192 rays (the
reference's quick-start recipe, README.md:71 / opt.py:32,40), then test PSNR on held-out views."""
import json, math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pl_amd import synthetic as syn
from ngp_pl_amd.bench_support import GpuDataset, surface_ground_truth
from ngp_pl_amd.networks import NGP
from ngp_pl_amd.rendering import render
from ngp_pl_amd.trainer import Trainer

steps = int(os.environ.get("STEPS", 30000))
torch.manual_seed(1337)
dev = torch.device("cuda")
model = NGP(0.5).to(dev); model.register_training_buffers()
tr = Trainer(model, lr=1e-2, num_epochs=30, steps_per_epoch=max(steps // 30, 1))
data = GpuDataset(800, 100, dev, seed=0)
test_poses = syn.hemisphere_poses(8, seed=999).to(dev)
log = []
torch.cuda.synchronize(); t0 = time.perf_counter()
cur = data.sample_native(8192, 0)
for i in range(steps):
    nxt = data.sample_native(8192, i + 1)
    tr.step(cur[0], cur[1], cur[2], next_batch=(nxt[0], nxt[1]))
    cur = nxt
    if (i + 1) % 5000 == 0 or i + 1 in (500, 1000, 2000):
        m = tr.metrics(); m["step"] = i + 1; m["elapsed_s"] = time.perf_counter() - t0
        log.append(m); print(m, flush=True)
torch.cuda.synchronize(); train_s = time.perf_counter() - t0
psnrs, frames_ms = [], []
for p in test_poses:
    ro, rd = syn.get_rays(data.directions, p)
    torch.cuda.synchronize(); t = time.perf_counter()
    out = render(model, ro, rd, test_time=True)
    torch.cuda.synchronize(); frames_ms.append((time.perf_counter() - t) * 1e3)
    gt = surface_ground_truth(ro, rd)
    mse = ((out["rgb"] - gt) ** 2).mean().item()
    psnrs.append(-10 * math.log10(mse))
occ = float(((model.density_bitfield[:, None] >> torch.arange(8, device=dev)) & 1).float().mean())
res = {"steps": steps, "train_seconds": train_s, "rays_per_s": steps * 8192 / train_s, "test_psnr_mean": sum(psnrs) / len(psnrs),
       "test_psnr": psnrs, "render_ms": frames_ms[1:], "occupancy": occ, "log": log,
       "finite_params": bool(torch.isfinite(model.xyz_encoder.params).all() and torch.isfinite(model.rgb_net.params).all())}
print(json.dumps(res))
