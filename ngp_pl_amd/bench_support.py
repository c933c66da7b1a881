"""Pieces bench.py and the smoke test share: the HBM-resident synthetic dataset (the GPU analogue
of datasets/base.py + nerf.py / colmap.py for a procedural scene) and the frame renderer used for the
FPS figure.  The multi-GPU gradient exchange lives in ngp_pl_amd/ddp.py."""
import math
import time

import torch

from . import synthetic as syn
from . import tcnn
from .rendering import render


# "unbounded" stand-in for the mip-NeRF360 recipe (benchmarking/benchmark_mipnerf360.sh:21-24: --scale 16 => 6
# cascades, exp_step_factor 1/256, black background): the Lego-like object twice as large at the centre, a ground
# slab and far-away primitives out to |x| ~ 13, inward-facing cameras at radius 4.
def unbounded_scene():
    boxes = [(tuple(2 * v for v in c), tuple(2 * v for v in h)) for c, h in syn._BOXES]
    spheres = [(tuple(2 * v for v in c), 2 * r) for c, r in syn._SPHERES]
    boxes.append(((0.0, 0.0, -0.65), (13.0, 13.0, 0.1)))                 # ground
    for k in range(12):
        a = 2 * math.pi * k / 12
        rad = 5.0 + 3.5 * (k % 3)
        c = (rad * math.cos(a), rad * math.sin(a), 0.4 + 0.3 * (k % 4))
        if k % 2:
            spheres.append((c, 0.8 + 0.2 * (k % 3)))
        else:
            boxes.append((c, (0.7, 0.5 + 0.1 * (k % 3), 1.0)))
    return boxes, spheres


@torch.no_grad()
def surface_ground_truth(rays_o, rays_d, boxes=None, spheres=None, white_bg=True):
    """Opaque rendering of the analytic scene (nearest ray/primitive hit, white or black background):
    cheap enough to produce 100 x 800x800 ground-truth images in seconds."""
    boxes = syn._BOXES if boxes is None else boxes
    spheres = syn._SPHERES if spheres is None else spheres
    n = rays_o.shape[0]
    dev = rays_o.device
    inv = 1.0 / rays_d
    best = torch.full((n,), float("inf"), device=dev)
    for c, h in boxes:
        c = rays_o.new_tensor(c); h = rays_o.new_tensor(h)
        t0 = (c - h - rays_o) * inv; t1 = (c + h - rays_o) * inv
        tn = torch.minimum(t0, t1).max(-1).values; tf = torch.maximum(t0, t1).min(-1).values
        hit = (tf > tn) & (tf > 0)
        best = torch.where(hit & (tn.clamp(min=0) < best), tn.clamp(min=0), best)
    a = (rays_d * rays_d).sum(-1)
    for c, r in spheres:
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
    return torch.where(hit[:, None], col, torch.ones_like(col) if white_bg else torch.zeros_like(col))


class GpuDataset:
    """poses (N,3,4), directions (H*W,3) and ground-truth colours (N, H*W, 3) resident in HBM;
    `sample` draws img/pix indices like BaseDataset.__getitem__ ('all_images', base.py:22-35) and
    forms the rays like NeRFSystem.forward (train.py:78-91), all on the GPU."""

    def __init__(self, res, n_images, device, seed=0, scene="lego"):
        self.W = self.H = res
        self.K = syn.intrinsics(res)
        self.directions = syn.get_ray_directions(res, res, self.K, device=device)
        self.scene = scene
        if scene == "lego":
            self.poses = syn.hemisphere_poses(n_images, seed=seed).to(device)
            boxes = spheres = None
        elif scene == "unbounded":
            self.poses = syn.hemisphere_poses(n_images, radius=4.0, seed=seed, min_elev_deg=5.0, max_elev_deg=40.0).to(device)
            boxes, spheres = unbounded_scene()
        else:
            raise ValueError("unknown scene %r" % scene)
        self.rgb = torch.empty(n_images, res * res, 3, dtype=torch.float32, device=device)
        for i in range(n_images):
            ro, rd = syn.get_rays(self.directions, self.poses[i])
            self.rgb[i] = surface_ground_truth(ro, rd, boxes, spheres, white_bg=(scene == "lego"))
        self.device = device

    def sample_native(self, n, step, seed=0, want_indices=False, out=None, stream_handle=None):
        """Same draw as `sample` in ONE kernel (ngp_sample_rays): indices from a counter-based RNG
        keyed by (seed, step), colour gather, pose rotation.  Returns rays_o, rays_d, rgb[, img, pix].
        `out` = (rays_o, rays_d, rgb) buffers to fill instead of allocating; `stream_handle` = raw stream to launch on."""
        from ._lib import call, ptr, stream
        dev = self.device
        if out is None:
            ro = torch.empty(n, 3, device=dev); rd = torch.empty(n, 3, device=dev); rgb = torch.empty(n, 3, device=dev)
        else:
            ro, rd, rgb = out
        img = pix = None
        if want_indices:
            img = torch.empty(n, dtype=torch.int32, device=dev); pix = torch.empty(n, dtype=torch.int32, device=dev)
        call("ngp_sample_rays", ptr(self.poses), ptr(self.directions), ptr(self.rgb), self.poses.shape[0], self.W * self.H, n,
             (int(seed) << 32) | (int(step) & 0xFFFFFFFF), ptr(ro), ptr(rd), ptr(rgb), None, ptr(img), ptr(pix),
             stream_handle if stream_handle is not None else stream())
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


@torch.no_grad()
def render_eval(model, data, poses, psnr=True, **render_kwargs):
    """The reference's evaluation loop (train.py:193-237 `validation_step`, test.ipynb cell 2) over `poses` (N,3,4): per pose,
    wall-clock of ray generation + `render(test_time=True)` bracketed by synchronize (what `fps = 1 / mean(t)` is quoted on), and --
    outside the timed bracket -- the PSNR of the frame against the analytic scene's ground truth (`psnr=True`).  One untimed frame
    first (allocator, workspaces)."""
    import math
    n_pix = data.W * data.H
    ro, rd = syn.get_rays(data.directions, poses[0])
    render(model, ro, rd, test_time=True, **render_kwargs)
    times, psnrs, n_samples, iters = [], [], 0.0, []
    for i in range(poses.shape[0]):
        torch.cuda.synchronize()
        t = time.perf_counter()
        ro, rd = syn.get_rays(data.directions, poses[i])
        out = render(model, ro, rd, test_time=True, **render_kwargs)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t)
        n_samples += float(out["total_samples"])
        if "n_iterations" in out:
            iters.append(int(out["n_iterations"]))
        if psnr:
            gt = surface_ground_truth(ro, rd)
            mse = float(((out["rgb"] - gt) ** 2).mean())
            psnrs.append(-10.0 * math.log10(max(mse, 1e-12)))
    mean = sum(times) / len(times)
    srt = sorted(times)
    res = {"fps": 1.0 / mean, "ms_per_frame": mean * 1e3, "ms_per_frame_median": srt[len(srt) // 2] * 1e3, "ms_per_frame_max": srt[-1] * 1e3,
           "n_frames": len(times), "poses": "held-out (hemisphere_poses seed 999; training cameras: seed 0)",
           "samples_per_ray": n_samples / len(times) / n_pix, "protocol": "1 / mean(wall of get_rays + render(test_time=True) per pose), test.ipynb cell 2"}
    if iters:
        res["iterations_mean"] = sum(iters) / len(iters)
    if psnr:
        res["psnr"] = sum(psnrs) / len(psnrs)
        res["psnr_min_max"] = [min(psnrs), max(psnrs)]
    return res
