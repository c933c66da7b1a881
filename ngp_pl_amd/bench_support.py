"""Pieces bench.py and the smoke test share: the HBM-resident synthetic dataset (the GPU analogue
of datasets/base.py + nerf.py for a procedural scene), the frame renderer used for the FPS figure,
the per-stage event timer and the native-gradient all-reduce."""
import time

import torch

from . import synthetic as syn
from . import tcnn
from .rendering import render


@torch.no_grad()
def surface_ground_truth(rays_o, rays_d):
    """Opaque rendering of the analytic scene (nearest ray/primitive hit, white background):
    cheap enough to produce 100 x 800x800 ground-truth images in seconds."""
    n = rays_o.shape[0]
    dev = rays_o.device
    inv = 1.0 / rays_d
    best = torch.full((n,), float("inf"), device=dev)
    for c, h in syn._BOXES:
        c = rays_o.new_tensor(c); h = rays_o.new_tensor(h)
        t0 = (c - h - rays_o) * inv; t1 = (c + h - rays_o) * inv
        tn = torch.minimum(t0, t1).max(-1).values; tf = torch.maximum(t0, t1).min(-1).values
        hit = (tf > tn) & (tf > 0)
        best = torch.where(hit & (tn.clamp(min=0) < best), tn.clamp(min=0), best)
    a = (rays_d * rays_d).sum(-1)
    for c, r in syn._SPHERES:
        co = rays_o - rays_o.new_tensor(c)
        hb = (rays_d * co).sum(-1)
        disc = hb * hb - a * ((co * co).sum(-1) - r * r)
        t = (-hb - disc.clamp(min=0).sqrt()) / a
        hit = (disc > 0) & (t > 0)
        best = torch.where(hit & (t < best), t, best)
    hit = torch.isfinite(best)
    x = rays_o + torch.where(hit, best, torch.zeros_like(best))[:, None] * rays_d
    dn = rays_d / rays_d.norm(dim=-1, keepdim=True)
    col = syn.colour(x, dn)
    return torch.where(hit[:, None], col, torch.ones_like(col))


class GpuDataset:
    """poses (N,3,4), directions (H*W,3) and ground-truth colours (N, H*W, 3) resident in HBM;
    `sample` draws img/pix indices like BaseDataset.__getitem__ ('all_images', base.py:22-35) and
    forms the rays like NeRFSystem.forward (train.py:78-91), all on the GPU."""

    def __init__(self, res, n_images, device, seed=0):
        self.W = self.H = res
        self.K = syn.intrinsics(res)
        self.directions = syn.get_ray_directions(res, res, self.K, device=device)
        self.poses = syn.hemisphere_poses(n_images, seed=seed).to(device)
        self.rgb = torch.empty(n_images, res * res, 3, dtype=torch.float32, device=device)
        for i in range(n_images):
            ro, rd = syn.get_rays(self.directions, self.poses[i])
            self.rgb[i] = surface_ground_truth(ro, rd)
        self.device = device

    def sample_native(self, n, step, seed=0, want_indices=False):
        """Same draw as `sample` in ONE kernel (ngp_sample_rays): indices from a counter-based RNG
        keyed by (seed, step), colour gather, pose rotation.  Returns rays_o, rays_d, rgb[, img, pix]."""
        from ._lib import call, ptr, stream
        dev = self.device
        ro = torch.empty(n, 3, device=dev); rd = torch.empty(n, 3, device=dev); rgb = torch.empty(n, 3, device=dev)
        img = pix = None
        if want_indices:
            img = torch.empty(n, dtype=torch.int32, device=dev); pix = torch.empty(n, dtype=torch.int32, device=dev)
        call("ngp_sample_rays", ptr(self.poses), ptr(self.directions), ptr(self.rgb), self.poses.shape[0], self.W * self.H, n,
             (int(seed) << 32) | (int(step) & 0xFFFFFFFF), ptr(ro), ptr(rd), ptr(rgb), None, ptr(img), ptr(pix), stream())
        return (ro, rd, rgb, img, pix) if want_indices else (ro, rd, rgb)

    def sample(self, n, gen):
        img = torch.randint(self.poses.shape[0], (n,), device=self.device, generator=gen)
        pix = torch.randint(self.W * self.H, (n,), device=self.device, generator=gen)
        ro, rd = syn.get_rays(self.directions[pix], self.poses[img])
        return ro, rd, self.rgb[img, pix].contiguous()


@torch.no_grad()
def render_fps(model, data, n_frames=5, **render_kwargs):
    """Frames/s of render(test_time=True) on full res x res images incl. ray generation, as the
    reference measures it (test.ipynb cell 2, show_gui.py:73-93)."""
    times = []
    n_total = 0
    for i in range(n_frames + 1):
        torch.cuda.synchronize()
        t = time.perf_counter()
        ro, rd = syn.get_rays(data.directions, data.poses[i % data.poses.shape[0]])
        out = render(model, ro, rd, test_time=True, **render_kwargs)
        torch.cuda.synchronize()
        if i > 0:                      # first frame warms the allocator
            times.append(time.perf_counter() - t)
            n_total += float(out["total_samples"])
    mean = sum(times) / len(times)
    res = {"fps": 1.0 / mean, "ms_per_frame": mean * 1e3, "samples_per_ray": n_total / len(times) / (data.W * data.H)}
    if "n_iterations" in out:
        res["iterations"] = out["n_iterations"]
    return res


def all_reduce_native_mlp(model, dist):
    """First half of DDP's gradient all-reduce on the native buffers: the MLP partial sums (10 240
    floats) are reduced asynchronously as soon as the MLP backward has produced them, i.e. while the
    hash-grid backward (the longest kernel of the step) is still running."""
    nat = model._native
    if nat is None:
        return
    enc, net = model.xyz_encoder, model.rgb_net
    n_d, n_r, n_part = enc.n_mlp, net.params.numel(), nat["n_partials"]
    dp, rp = nat["density_partials"], nat["rgb_partials"]
    if dp.is_cuda:                               # two launches into one buffer (the torch sum/sum/cat costs ~40 us of GPU time)
        from ._lib import call, ptr, stream
        small = torch.empty(n_d + n_r, dtype=torch.float32, device=dp.device)
        call("ngp_reduce_partials", ptr(dp), n_part, n_d, ptr(small), stream())
        call("ngp_reduce_partials", ptr(rp), n_part, n_r, ptr(small[n_d:]), stream())
    else:                                        # host tensors: the gloo test of this logic
        small = torch.cat([dp.view(n_part, n_d).sum(0), rp.view(n_part, n_r).sum(0)])
    nat["_mlp_small"] = small
    nat["_mlp_work"] = dist.all_reduce(small, async_op=True)


def all_reduce_native(model, dist, world):
    """DDP's gradient all-reduce (mean) on the native buffers: one collective for the packed-f16
    grid gradient (22.9 MB instead of DDP's 45.7 MB f32) and one for the MLP partial sums (started
    earlier by `all_reduce_native_mlp` when the trainer offers the hook, otherwise here)."""
    nat = model._native
    if nat is None:
        return
    enc = model.xyz_encoder
    if "_mlp_work" not in nat:
        all_reduce_native_mlp(model, dist)
    dist.all_reduce(nat["grid16"])
    nat.pop("_mlp_work").wait()
    small = nat.pop("_mlp_small")
    nat["density_partials"] = small[:enc.n_mlp].contiguous()
    nat["rgb_partials"] = small[enc.n_mlp:].contiguous()
    nat["n_partials"] = 1
    nat["scale"] = nat["scale"] * world      # mean over ranks folded into the unscale
