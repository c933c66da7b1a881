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


# "lego_hard": the Lego-like object with the things a real Lego bulldozer has and the plain scene lacks -- studs on the plates, two
# treads made of a lattice of 4 mm bars (half an occupancy cell wide: a ray through a tread region meets occupied cells, mostly empty
# space and up to ten thin crossings), a HOLLOW cabin (5 mm walls with window gaps: rays enter, cross air, leave), 3-4 mm arms -- and
# a finite density (sigma 30-60 per unit length instead of an opaque surface): a ray keeps compositing for up to ln(1e4) / sigma =
# 0.15-0.3 units of solid, i.e. tens of samples, instead of stopping at the first surface.  Ground truth = exact volume rendering
# of the union of primitives at constant density (`volumetric_ground_truth`).
def lego_hard_scene(size=1.0):
    """`size` scales the whole object about the origin (1.2: the blade reaches |x| = 0.49 of the [-0.5, 0.5] box, the object
    fills the frame the way the real Lego does in its 800x800 renders)."""
    boxes, spheres = _lego_hard_primitives()
    if size != 1.0:
        boxes = [(tuple(size * v for v in c), tuple(size * v for v in h)) for c, h in boxes]
        spheres = [(tuple(size * v for v in c), size * r) for c, r in spheres]
    return boxes, spheres


def _lego_hard_primitives():
    boxes, spheres = [], []
    boxes.append(((0.0, 0.0, -0.24), (0.34, 0.24, 0.03)))                      # base plate
    boxes.append(((0.17, 0.0, -0.13), (0.10, 0.12, 0.06)))                     # hood (solid)
    boxes.append(((-0.12, 0.0, -0.15), (0.16, 0.14, 0.06)))                    # body (solid)
    # studs: 6 x 4 on the body top, 3 x 3 on the hood
    for i in range(6):
        for j in range(4):
            boxes.append(((-0.26 + 0.056 * i, -0.105 + 0.07 * j, -0.083), (0.012, 0.012, 0.007)))
    for i in range(3):
        for j in range(3):
            boxes.append(((0.10 + 0.07 * i, -0.08 + 0.08 * j, -0.063), (0.012, 0.012, 0.007)))
    # hollow cabin: four walls with window gaps (each wall = sill + two posts + lintel) and a roof, 5 mm thick
    cx, cz, hx, hy, hz, w = -0.15, 0.0, 0.09, 0.10, 0.09, 0.0025
    for sy in (-1, 1):
        y = sy * hy
        boxes.append(((cx, y, cz - hz + 0.02), (hx, w, 0.02)))                 # sill
        boxes.append(((cx, y, cz + hz - 0.012), (hx, w, 0.012)))               # lintel
        for sx in (-1, 1):
            boxes.append(((cx + sx * (hx - 0.01), y, cz), (0.01, w, hz)))      # posts
    for sx in (-1, 1):
        x = cx + sx * hx
        boxes.append(((x, 0.0, cz - hz + 0.02), (w, hy, 0.02)))
        boxes.append(((x, 0.0, cz + hz - 0.012), (w, hy, 0.012)))
        for sy in (-1, 1):
            boxes.append(((x, sy * (hy - 0.01), cz), (w, 0.01, hz)))
    boxes.append(((cx, 0.0, cz + hz), (hx + 0.01, hy + 0.01, 0.004)))          # roof
    # treads: per side 28 vertical bars of 4 mm x 4 mm on a 22 mm pitch between two 4 mm rails
    for sy in (-1, 1):
        y = sy * 0.205
        for k in range(28):
            boxes.append(((-0.30 + 0.022 * k, y, -0.165), (0.002, 0.012, 0.045)))
        for z in (-0.21, -0.12):
            boxes.append(((0.0, y, z), (0.31, 0.012, 0.002)))
    # arms + blade
    for sy in (-1, 1):
        boxes.append(((0.16, sy * 0.15, -0.02), (0.20, 0.002, 0.0015)))
        boxes.append(((0.16, sy * 0.15, -0.06), (0.20, 0.0015, 0.002)))
    boxes.append(((0.40, 0.0, -0.12), (0.006, 0.20, 0.07)))
    spheres += [((0.30, 0.0, 0.10), 0.06)] + [((sx, sy, -0.17), 0.05) for sx in (-0.24, 0.24) for sy in (-0.205, 0.205)]
    return boxes, spheres


@torch.no_grad()
def volumetric_ground_truth(rays_o, rays_d, boxes, spheres, sigma, white_bg=True, chunk=1 << 17):
    """EXACT volume rendering of {density = sigma inside the union of the primitives, 0 outside; colour = syn.colour(x, view)}:
    per ray the entry / exit parameters of every primitive are sorted, the inside-count is swept along the ray, and each
    constant-density segment [a, b] contributes T(a) (1 - exp(-sigma (b - a))) times its weight-averaged colour, formed by a two-node
    rule at mean +- standard deviation of the segment's own transmittance weight (exact for a colour that is quadratic along the
    segment; the colour stripes have a period of 0.2-0.3, the weight of a segment sits within ~1/sigma = 0.02-0.07 of its entry).
    Transmittance and opacity are exact.  Rays that miss every primitive are skipped."""
    n, dev = rays_o.shape[0], rays_o.device
    out = torch.ones(n, 3, device=dev) if white_bg else torch.zeros(n, 3, device=dev)
    bc = torch.tensor([b[0] for b in boxes], device=dev); bh = torch.tensor([b[1] for b in boxes], device=dev)
    sc = torch.tensor([s_[0] for s_ in spheres], device=dev); sr = torch.tensor([s_[1] for s_ in spheres], device=dev)
    inf = float("inf")
    for lo in range(0, n, chunk):
        o, d = rays_o[lo:lo + chunk], rays_d[lo:lo + chunk]
        inv = 1.0 / d
        t0 = (bc[None] - bh[None] - o[:, None]) * inv[:, None]; t1 = (bc[None] + bh[None] - o[:, None]) * inv[:, None]
        tn = torch.minimum(t0, t1).max(-1).values; tf = torch.maximum(t0, t1).min(-1).values
        hit = (tf > tn) & (tf > 0)
        t_in = [torch.where(hit, tn.clamp(min=0), torch.full_like(tn, inf))]; t_out = [torch.where(hit, tf, torch.full_like(tf, inf))]
        if len(spheres):
            a = (d * d).sum(-1, keepdim=True)
            co = o[:, None] - sc[None]
            hb = (d[:, None] * co).sum(-1)
            disc = hb * hb - a * ((co * co).sum(-1) - sr[None] ** 2)
            rt = disc.clamp(min=0).sqrt()
            s0, s1 = (-hb - rt) / a, (-hb + rt) / a
            hs = (disc > 0) & (s1 > 0)
            t_in.append(torch.where(hs, s0.clamp(min=0), torch.full_like(s0, inf))); t_out.append(torch.where(hs, s1, torch.full_like(s1, inf)))
        t_in, t_out = torch.cat(t_in, 1), torch.cat(t_out, 1)
        rows = torch.isfinite(t_in).any(1).nonzero()[:, 0]                       # rays that meet the object at all
        if rows.numel() == 0:
            continue
        t_in, t_out, o, d = t_in[rows], t_out[rows], o[rows], d[rows]
        K = t_in.shape[1]
        ev, order = torch.cat([t_in, t_out], 1).sort(1)
        step = torch.where(order < K, 1, -1).to(torch.int32)
        inside = step.cumsum(1)[:, :-1] > 0                                        # between event j and j + 1
        a_, b_ = ev[:, :-1], ev[:, 1:]
        L = torch.where(inside & torch.isfinite(b_), b_ - a_, torch.zeros_like(a_)).clamp(min=0)
        keep = (L > 0).any(0)                                                       # (event slots no ray of the chunk uses: all the inf tail)
        a_, L = a_[:, keep], L[:, keep]
        tau = sigma * L
        T = torch.exp(-(tau.cumsum(1) - tau))
        e = torch.exp(-tau)
        w = T * (1 - e)
        # colour of a segment = the weight-averaged colour over it, by a two-node rule that is exact for a colour QUADRATIC along the
        # segment: nodes at mean +- standard deviation of the truncated exponential weight sigma e^(-sigma t) on [0, L]
        # (mean 1/sigma - L e / (1 - e), variance 1/sigma^2 - L^2 e / (1 - e)^2 with e = exp(-sigma L); L/2 and L^2/12 as sigma L -> 0)
        one_m_e = (1 - e).clamp(min=1e-12)
        mean = torch.where(tau > 1e-3, 1.0 / sigma - L * e / one_m_e, 0.5 * L)
        var = torch.where(tau > 1e-3, 1.0 / sigma ** 2 - L * L * e / (one_m_e * one_m_e), L * L / 12.0).clamp(min=0)
        sd = var.sqrt()
        dn = d / d.norm(dim=-1, keepdim=True)
        acc = torch.zeros(o.shape[0], 3, device=dev)
        for j0 in range(0, a_.shape[1], 32):                                       # (colour over 32 segment slots at a time: bounded temporaries)
            wj = 0.5 * w[:, j0:j0 + 32, None]
            for sign in (-1.0, 1.0):
                tm = torch.where(L[:, j0:j0 + 32] > 0, a_[:, j0:j0 + 32] + (mean[:, j0:j0 + 32] + sign * sd[:, j0:j0 + 32]).clamp(min=0), torch.zeros_like(a_[:, j0:j0 + 32]))
                x = o[:, None] + tm[..., None] * d[:, None]
                acc += (wj * syn.colour(x, dn[:, None].expand_as(x))).sum(1)
        opac = w.sum(1)
        out[lo + rows] = (acc + ((1 - opac)[:, None] if white_bg else 0.0)).clamp(0, 1)
    return out


class GpuDataset:
    """poses (N,3,4), directions (H*W,3) and ground-truth colours (N, H*W, 3) resident in HBM;
    `sample` draws img/pix indices like BaseDataset.__getitem__ ('all_images', base.py:22-35) and
    forms the rays like NeRFSystem.forward (train.py:78-91), all on the GPU."""

    def __init__(self, res, n_images, device, seed=0, scene="lego", sigma=None, size=1.0):
        """scene: "lego" (opaque surfaces), "unbounded" (the scale-16 stand-in), "lego_hard" (thin structures, hollow cabin, finite
        density `sigma`: volumetric ground truth)."""
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
        elif scene == "lego_hard":
            self.poses = syn.hemisphere_poses(n_images, seed=seed).to(device)
            boxes, spheres = lego_hard_scene(size)
            sigma = 45.0 if sigma is None else float(sigma)
        else:
            raise ValueError("unknown scene %r" % scene)
        self.boxes, self.spheres, self.sigma, self.white_bg = boxes, spheres, sigma, scene != "unbounded"
        self.rgb = torch.empty(n_images, res * res, 3, dtype=torch.float32, device=device)
        for i in range(n_images):
            ro, rd = syn.get_rays(self.directions, self.poses[i])
            self.rgb[i] = self.ground_truth(ro, rd)
        self.device = device

    def ground_truth(self, rays_o, rays_d):
        """Pixel colours of the scene for these rays (training images and the evaluation's reference frames alike)."""
        if self.scene == "lego_hard":
            return volumetric_ground_truth(rays_o, rays_d, self.boxes, self.spheres, self.sigma, white_bg=True)
        return surface_ground_truth(rays_o, rays_d, self.boxes, self.spheres, white_bg=self.white_bg)

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
            gt = data.ground_truth(ro, rd) if hasattr(data, "ground_truth") else surface_ground_truth(ro, rd)
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


def sharded_eval(render_pose, n_poses, rank, world, dist=None, device="cpu"):
    """The reference's validation protocol across ranks (train.py:193-237 under Lightning's DDP strategy, train.py:268-272): the
    poses are dealt round-robin (DistributedSampler: pose i -> rank i % world), every rank renders ITS poses -- `render_pose(i)` ->
    (wall ms of get_rays + render(test_time=True), PSNR of the frame) --, the per-pose metrics are `all_gather`ed (train.py:227-237)
    and every rank ends up with the same record: gathered mean PSNR over all poses, frames/s PER GPU (1 / mean frame time of that
    rank's poses, the README's FPS) and their sum.  Device agnostic in its collective logic (tests/test_ddp_gloo.py drives it with
    gloo and a stand-in renderer)."""
    mine = list(range(rank, n_poses, world))
    per = -(-n_poses // world)
    rec = torch.full((per, 3), float("nan"), dtype=torch.float64)
    for j, i in enumerate(mine):
        ms, psnr = render_pose(i)
        rec[j, 0], rec[j, 1], rec[j, 2] = float(i), float(ms), float(psnr)
    if dist is not None and world > 1:
        mine_t = rec.to(device)
        parts = [torch.empty_like(mine_t) for _ in range(world)]
        dist.all_gather(parts, mine_t)
        allrec = torch.stack([p.cpu() for p in parts])                  # (world, per, 3)
    else:
        allrec = rec[None]
    ok = ~torch.isnan(allrec[..., 0])
    seen = sorted(int(v) for v in allrec[..., 0][ok].tolist())
    if seen != list(range(n_poses)):
        raise RuntimeError("sharded evaluation: poses %s were rendered, expected each of 0..%d exactly once" % (seen[:8], n_poses - 1))
    fps = []
    for r in range(allrec.shape[0]):
        m = ok[r]
        fps.append(1e3 / float(allrec[r, :, 1][m].mean()) if bool(m.any()) else None)
    have = [f for f in fps if f is not None]
    return {"n_poses": n_poses, "ranks": int(allrec.shape[0]), "poses_per_rank": [int(ok[r].sum()) for r in range(allrec.shape[0])],
            "psnr": float(allrec[..., 2][ok].mean()), "psnr_min_max": [float(allrec[..., 2][ok].min()), float(allrec[..., 2][ok].max())],
            "render_fps_per_gpu": fps, "render_fps_per_gpu_mean": sum(have) / len(have), "render_fps_aggregate": sum(have),
            "protocol": "poses dealt round-robin over ranks, per-pose metrics all_gather'ed (train.py:193-237); FPS per GPU = 1 / mean(wall of "
                        "get_rays + render(test_time=True)) over that rank's poses"}
