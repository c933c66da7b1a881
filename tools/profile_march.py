"""Scratch: time the train march (count pass) with and without the coarse LDS mask on a realistic sparse bitfield."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pl_amd import synthetic as syn
from ngp_pl_amd._lib import call, ptr, stream
dev = "cuda"
grid = syn.analytic_density_grid(1, 0.5, 128)
bf = torch.from_numpy(syn.pack_bitfield_np(grid, 10.0)).to(dev)
print("occupancy %.3f" % ((grid > 10).mean()))
K = syn.intrinsics(800); dirs = syn.get_ray_directions(800, 800, K, device=dev); poses = syn.hemisphere_poses(100).to(dev)
for n in (8192, 640000):
    img = torch.randint(100, (n,), device=dev); pix = torch.randint(640000, (n,), device=dev)
    ro, rd = syn.get_rays(dirs[pix], poses[img])
    hits = torch.empty(n, 2, device=dev)
    c = torch.zeros(1, 3, device=dev); h = torch.full((1, 3), 0.5, device=dev)
    call("ngp_ray_aabb_near", ptr(ro), ptr(rd), ptr(c), ptr(h), 0.01, n, ptr(hits), stream())
    noise = torch.rand(n, device=dev); rays_a = torch.empty(n, 3, dtype=torch.int64, device=dev); counter = torch.empty(2, dtype=torch.int32, device=dev)
    scratch = torch.empty(n * 1024, device=dev); coarse = torch.empty(4096, dtype=torch.uint8, device=dev)
    for name, ws in (("no mask", None), ("coarse mask", coarse)):
        def f():
            call("ngp_raymarching_train_count", ptr(ro), ptr(rd), ptr(hits), ptr(bf), 1, 0.5, 0.0, ptr(noise), 128, 1024, n, ptr(rays_a), ptr(counter), ptr(scratch), ptr(ws), stream())
        for _ in range(3): f()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): f()
        torch.cuda.synchronize()
        print("n=%6d %-12s %8.1f us   S=%d" % (n, name, (time.perf_counter() - t) / 20 * 1e6, int(counter[0])))
