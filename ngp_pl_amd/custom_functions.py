"""Autograd operator surface of the reference, same class names and positional arguments
(/root/reference/models/custom_functions.py:8-173), on top of `ngp_pl_amd.vren`.

AMP contract kept: the vren-backed ops see float32 (the reference uses
custom_fwd(cast_inputs=torch.float32), custom_functions.py:27,50,78,138,164).
"""
import torch
from torch.amp import custom_bwd, custom_fwd

from . import vren

_fwd = custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd = custom_bwd(device_type="cuda")


class RayAABBIntersector(torch.autograd.Function):
    """rays x axis-aligned boxes -> (hit_cnt (R), hits_t (R,max_hits,2), hits_voxel_idx (R,max_hits)),
    near to far, -1 where empty (custom_functions.py:8-29)."""

    @staticmethod
    @_fwd
    def forward(ctx, rays_o, rays_d, center, half_size, max_hits):
        out = vren.ray_aabb_intersect(rays_o, rays_d, center, half_size, max_hits)
        ctx.mark_non_differentiable(*out)
        return tuple(out)


class RaySphereIntersector(torch.autograd.Function):
    """rays x spheres, same outputs (custom_functions.py:32-52)."""

    @staticmethod
    @_fwd
    def forward(ctx, rays_o, rays_d, center, radii, max_hits):
        out = vren.ray_sphere_intersect(rays_o, rays_d, center, radii, max_hits)
        ctx.mark_non_differentiable(*out)
        return tuple(out)


def segment_sum(values, rays_a):
    """Per-ray sums of packed per-sample rows; replaces torch_scatter.segment_csr in
    RayMarcher.backward (custom_functions.py:107-110).  Results are placed at ray_idx, so the
    gradient lands on the right ray whatever the row order of rays_a (the reference returns them
    in row order, which is only right when rows happen to be ray-ordered)."""
    n_rays = rays_a.shape[0]
    seg = torch.repeat_interleave(rays_a[:, 0], rays_a[:, 2])
    out = torch.zeros((n_rays,) + values.shape[1:], dtype=values.dtype, device=values.device)
    return out.index_add_(0, seg, values)


class RayMarcher(torch.autograd.Function):
    """March rays through the occupancy bitfield (custom_functions.py:55-112).
    Returns rays_a (R,3) [ray_idx, start_idx, N_samples], xyzs (S,3), dirs (S,3), deltas (S), ts (S),
    total_samples (0-dim)."""

    @staticmethod
    @_fwd
    def forward(ctx, rays_o, rays_d, hits_t, density_bitfield, cascades, scale, exp_step_factor, grid_size, max_samples):
        noise = torch.rand_like(rays_o[:, 0])        # jitter of the first sample (custom_functions.py:83)
        rays_a, xyzs, dirs, deltas, ts, counter = vren.raymarching_train(
            rays_o, rays_d, hits_t, density_bitfield, cascades, scale, exp_step_factor, noise, grid_size, max_samples)
        ctx.save_for_backward(rays_a, ts)
        ctx.mark_non_differentiable(rays_a, deltas, ts)
        return rays_a, xyzs, dirs, deltas, ts, counter[0]

    @staticmethod
    @_bwd
    def backward(ctx, dL_drays_a, dL_dxyzs, dL_ddirs, dL_ddeltas, dL_dts, dL_dtotal_samples):
        rays_a, ts = ctx.saved_tensors
        dL_drays_o = segment_sum(dL_dxyzs, rays_a)
        dL_drays_d = segment_sum(dL_dxyzs * ts[:, None] + dL_ddirs, rays_a)
        return dL_drays_o, dL_drays_d, None, None, None, None, None, None, None


class VolumeRenderer(torch.autograd.Function):
    """Front-to-back compositing over packed samples, training only (custom_functions.py:115-159).
    Returns total_samples (0-dim), opacity (R), depth (R), rgb (R,3), ws (S)."""

    @staticmethod
    @_fwd
    def forward(ctx, sigmas, rgbs, deltas, ts, rays_a, T_threshold):
        total_samples, opacity, depth, rgb, ws = vren.composite_train_fw(
            sigmas.contiguous(), rgbs.contiguous(), deltas, ts, rays_a, T_threshold)
        ctx.save_for_backward(sigmas, rgbs, deltas, ts, rays_a, opacity, depth, rgb, ws)
        ctx.T_threshold = T_threshold
        return total_samples.sum(), opacity, depth, rgb, ws

    @staticmethod
    @_bwd
    def backward(ctx, dL_dtotal_samples, dL_dopacity, dL_ddepth, dL_drgb, dL_dws):
        sigmas, rgbs, deltas, ts, rays_a, opacity, depth, rgb, ws = ctx.saved_tensors
        dL_dsigmas, dL_drgbs = vren.composite_train_bw(
            dL_dopacity.contiguous(), dL_ddepth.contiguous(), dL_drgb.contiguous(), dL_dws.contiguous(),
            sigmas.contiguous(), rgbs.contiguous(), ws, deltas, ts, rays_a, opacity, depth, rgb, ctx.T_threshold)
        return dL_dsigmas, dL_drgbs, None, None, None, None


class TruncExp(torch.autograd.Function):
    """exp with the backward clamped to [-15, 15] (custom_functions.py:162-173)."""

    @staticmethod
    @_fwd
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    @_bwd
    def backward(ctx, dL_dout):
        (x,) = ctx.saved_tensors
        return dL_dout * torch.exp(x.clamp(-15, 15))
