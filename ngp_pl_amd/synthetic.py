"""Procedural stand-in for the Synthetic-NeRF "Lego" inputs (there is no dataset on the box).

Mirrors what the reference's data layer hands to the hot path, nothing more:
  * pinhole intrinsics as datasets/nerf.py:25-31 (fx = fy = 0.5*W/tan(0.5*0.6911), cx = cy = W/2),
  * pixel-centre ray directions, un-normalised, camera frame [right, down, front]
    (datasets/ray_utils.py:7-43), world rays by rotating with c2w (ray_utils.py:46-70),
  * cameras on the upper hemisphere at radius 1.5 looking at the origin (the blender poses are
    rescaled to that radius, datasets/nerf.py:69-72),
  * a scene inside the [-0.5, 0.5]^3 box: an analytic density/colour field ("bricks": a union of
    boxes, studs and a sphere, occupying a few percent of the 128^3 grid) that can be
    (a) rasterised into the Morton-ordered occupancy bitfield the marcher reads, and
    (b) volume-rendered into ground-truth pixel colours for training.
Everything is deterministic given the seed.
"""
import math

import numpy as np
import torch

CAMERA_ANGLE_X = 0.6911112070083618  # Lego transforms_train.json value used by the blender loader


def intrinsics(W, H=None):
    """(3,3) K as datasets/nerf.py:25-31 builds it."""
    H = W if H is None else H
    fx = fy = 0.5 * W / math.tan(0.5 * CAMERA_ANGLE_X)
    return torch.tensor([[fx, 0, W / 2], [0, fy, H / 2], [0, 0, 1]], dtype=torch.float32)


def get_ray_directions(H, W, K, device="cpu"):
    """(H*W, 3) directions through pixel centres; datasets/ray_utils.py:28-43 (random=False)."""
    v, u = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=device),
                          torch.arange(W, dtype=torch.float32, device=device), indexing="ij")
    fx, fy, cx, cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
    d = torch.stack([(u - cx + 0.5) / fx, (v - cy + 0.5) / fy, torch.ones_like(u)], -1)
    return d.reshape(-1, 3)


def get_rays(directions, c2w):
    """datasets/ray_utils.py:46-70: rays_d = R @ dir (NOT normalised), rays_o = camera centre.  One camera, CUDA tensors, no
    gradient wanted: one native launch (ngp_get_rays); otherwise torch ops (differentiable: pose optimisation, train.py:86-89)."""
    if c2w.ndim == 2 and directions.is_cuda and c2w.is_cuda and directions.dtype == torch.float32 and c2w.dtype == torch.float32 \
            and not (torch.is_grad_enabled() and (directions.requires_grad or c2w.requires_grad)):
        from ._lib import call, device_guard, ptr, stream
        d = directions if directions.is_contiguous() else directions.contiguous()
        p = c2w if c2w.is_contiguous() else c2w.contiguous()
        rays_o = torch.empty_like(d); rays_d = torch.empty_like(d)
        with device_guard(d.device):
            call("ngp_get_rays", ptr(d), ptr(p), d.shape[0], ptr(rays_o), ptr(rays_d), stream())
        return rays_o, rays_d
    if c2w.ndim == 2:
        rays_d = directions @ c2w[:, :3].T
    else:
        rays_d = (directions[:, None, :] * c2w[..., :3]).sum(-1)     # elementwise: a batched GEMM launch costs 70 us here
    rays_o = c2w[..., 3].expand_as(rays_d)
    return rays_o.contiguous(), rays_d.contiguous()


def hemisphere_poses(n, radius=1.5, seed=0, min_elev_deg=10.0, max_elev_deg=80.0):
    """(n,3,4) camera-to-world, camera frame [right, down, front], looking at the origin."""
    g = np.random.RandomState(seed)
    az = g.uniform(0, 2 * np.pi, n)
    el = np.deg2rad(g.uniform(min_elev_deg, max_elev_deg, n))
    pos = radius * np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], -1)
    poses = np.zeros((n, 3, 4), np.float32)
    for i in range(n):
        front = -pos[i] / np.linalg.norm(pos[i])
        right = np.cross(front, np.array([0.0, 0.0, 1.0]))
        right /= np.linalg.norm(right)
        down = np.cross(front, right)
        poses[i, :, 0], poses[i, :, 1], poses[i, :, 2], poses[i, :, 3] = right, down, front, pos[i]
    return torch.from_numpy(poses)


# ---------------------------------------------------------------------------------------------
# analytic scene
# ---------------------------------------------------------------------------------------------
_BOXES = [  # centre, half extent
    ((0.0, 0.0, -0.22), (0.34, 0.22, 0.05)),     # base plate
    ((-0.12, 0.0, -0.07), (0.16, 0.14, 0.10)),   # body
    ((0.17, 0.0, -0.10), (0.10, 0.12, 0.07)),    # hood
    ((-0.15, 0.0, 0.10), (0.09, 0.10, 0.07)),    # cabin
    ((0.05, 0.17, 0.02), (0.22, 0.02, 0.02)),    # arm
    ((0.05, -0.17, 0.02), (0.22, 0.02, 0.02)),   # arm
]
_SPHERES = [((0.30, 0.0, 0.12), 0.09), ((-0.25, 0.20, -0.12), 0.07), ((-0.25, -0.20, -0.12), 0.07),
            ((0.22, 0.20, -0.12), 0.07), ((0.22, -0.20, -0.12), 0.07)]
SIGMA_INSIDE = 120.0
EDGE = 0.004  # soft edge width of the density


def signed_distance(x):
    """x (...,3) torch -> signed distance to the union of primitives (negative inside)."""
    d = None
    for c, h in _BOXES:
        q = (x - x.new_tensor(c)).abs() - x.new_tensor(h)
        sd = q.clamp(min=0).norm(dim=-1) + q.max(dim=-1).values.clamp(max=0)
        d = sd if d is None else torch.minimum(d, sd)
    for c, r in _SPHERES:
        sd = (x - x.new_tensor(c)).norm(dim=-1) - r
        d = torch.minimum(d, sd)
    return d


def density(x):
    return SIGMA_INSIDE * torch.sigmoid(-signed_distance(x) / EDGE)


def colour(x, d_unit=None):
    """Albedo stripes + a mild view-dependent term, in [0,1]."""
    base = 0.5 + 0.5 * torch.sin(x * x.new_tensor([23.0, 17.0, 29.0]) + x.new_tensor([0.3, 1.1, 2.0]))
    base = 0.15 + 0.7 * base
    if d_unit is not None:
        base = base + 0.1 * (d_unit * d_unit.new_tensor([0.0, 0.0, 1.0])).sum(-1, keepdim=True)
    return base.clamp(0, 1)


@torch.no_grad()
def render_ground_truth(rays_o, rays_d, n_steps=384, scale=0.5, white_bg=True):
    """Quadrature volume rendering of the analytic field inside the [-scale,scale]^3 box."""
    inv = 1.0 / rays_d
    t0 = (-scale - rays_o) * inv
    t1 = (scale - rays_o) * inv
    tn = torch.minimum(t0, t1).max(-1).values.clamp(min=0)
    tf = torch.maximum(t0, t1).min(-1).values
    hit = tf > tn
    tf = torch.where(hit, tf, tn)
    u = (torch.arange(n_steps, device=rays_o.device, dtype=torch.float32) + 0.5) / n_steps
    t = tn[:, None] + (tf - tn)[:, None] * u[None]
    dt = ((tf - tn) / n_steps)[:, None]
    x = rays_o[:, None] + t[..., None] * rays_d[:, None]
    dn = rays_d / rays_d.norm(dim=-1, keepdim=True)
    sig = density(x)
    a = 1 - torch.exp(-sig * dt)
    T = torch.cumprod(torch.cat([torch.ones_like(a[:, :1]), 1 - a[:, :-1]], 1), 1)
    w = a * T
    rgb = (w[..., None] * colour(x, dn[:, None].expand_as(x))).sum(1)
    opacity = w.sum(1)
    if white_bg:
        rgb = rgb + (1 - opacity)[:, None]
    return rgb.clamp(0, 1), opacity


# ---------------------------------------------------------------------------------------------
# occupancy grid in the reference's layout (Morton order per cascade, bit i of byte n = cell 8n+i)
# ---------------------------------------------------------------------------------------------
def morton3D_np(coords):
    """numpy version of raymarching.cu:35-50 (3 x 10 bit interleave)."""
    def expand(v):
        v = v.astype(np.uint32)
        v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
        v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
        v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
        v = (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
        return v
    c = np.asarray(coords)
    return (expand(c[..., 0]) | (expand(c[..., 1]) << 1) | (expand(c[..., 2]) << 2)).astype(np.int64)


def analytic_density_grid(cascades=1, scale=0.5, grid_size=128):
    """(cascades, G^3) float32 density at cell centres, Morton order (networks.py:240-264 layout)."""
    G = grid_size
    ii = np.stack(np.meshgrid(np.arange(G), np.arange(G), np.arange(G), indexing="ij"), -1).reshape(-1, 3)
    idx = morton3D_np(ii)
    grid = np.zeros((cascades, G ** 3), np.float32)
    for c in range(cascades):
        s = min(2.0 ** (c - 1), scale)
        centres = ((ii + 0.5) / G * 2 - 1) * s
        with torch.no_grad():
            sig = density(torch.from_numpy(centres.astype(np.float32))).numpy()
        grid[c, idx] = sig
    return grid


def pack_bitfield_np(grid, threshold):
    """numpy restatement of packbits (raymarching.cu:122-141) for building test inputs."""
    bits = (grid.reshape(-1, 8) > threshold).astype(np.uint8)
    return (bits << np.arange(8, dtype=np.uint8)).sum(1).astype(np.uint8)


def random_blob_bitfield(cascades=1, grid_size=128, fill=0.08, seed=0):
    """Random but spatially coherent occupancy (smoothed noise thresholded to `fill`)."""
    g = np.random.RandomState(seed)
    G = grid_size
    out = np.zeros((cascades, G ** 3), np.float32)
    ii = np.stack(np.meshgrid(np.arange(G), np.arange(G), np.arange(G), indexing="ij"), -1).reshape(-1, 3)
    idx = morton3D_np(ii)
    for c in range(cascades):
        coarse = g.rand(G // 8 + 1, G // 8 + 1, G // 8 + 1).astype(np.float32)
        fine = np.kron(coarse, np.ones((8, 8, 8), np.float32))[:G, :G, :G]
        fine = 0.5 * fine + 0.25 * np.roll(fine, 3, 0) + 0.25 * np.roll(fine, 5, 1)
        thr = np.quantile(fine, 1 - fill)
        out[c, idx] = (fine.reshape(-1) > thr).astype(np.float32)
    return pack_bitfield_np(out, 0.5)
