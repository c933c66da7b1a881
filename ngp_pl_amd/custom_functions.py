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


def _intersector(native_fn, summary):
    """Both intersection operators are the same shell around one native call: no gradients, three
    outputs (hit count (R), hit intervals (R,max_hits,2) near to far with -1 padding, primitive
    index (R,max_hits))."""

    class _Op(torch.autograd.Function):
        __doc__ = summary

        @staticmethod
        @_fwd
        def forward(ctx, origins, directions, centres, extents, max_hits):
            found = tuple(native_fn(origins, directions, centres, extents, max_hits))
            ctx.mark_non_differentiable(*found)
            return found

    return _Op


RayAABBIntersector = _intersector(vren.ray_aabb_intersect, "rays x axis-aligned boxes (custom_functions.py:8-29)")
RayAABBIntersector.__name__ = RayAABBIntersector.__qualname__ = "RayAABBIntersector"
RaySphereIntersector = _intersector(vren.ray_sphere_intersect, "rays x spheres, extents = radii (custom_functions.py:32-52)")
RaySphereIntersector.__name__ = RaySphereIntersector.__qualname__ = "RaySphereIntersector"


def segment_sum(values, rays_a):
    """Per-ray sums of packed per-sample rows; replaces torch_scatter.segment_csr in
    RayMarcher.backward (custom_functions.py:107-110).  Results are placed at ray_idx, and the owner of a
    sample is derived from the rows' start_idx, so the gradient lands on the right ray whatever the row order
    of rays_a (this package packs in ray order; the reference's atomics hand out ray_count and start_idx
    independently, and its segment_csr over rays_a[:,1] is only right when the rows happen to be start-ordered)."""
    n_rows, n_samples = rays_a.shape[0], values.shape[0]
    acc = values.new_zeros((n_rows,) + tuple(values.shape[1:]))
    if n_samples == 0 or n_rows == 0:
        return acc
    rows = torch.nonzero(rays_a[:, 2] > 0)[:, 0]                      # segments that hold samples
    starts, order = torch.sort(rays_a[rows, 1])
    seg = torch.searchsorted(starts, torch.arange(n_samples, device=values.device), right=True) - 1   # latest segment start <= sample
    acc.index_add_(0, rays_a[rows[order], 0][seg], values)
    return acc


class RayMarcher(torch.autograd.Function):
    """March rays through the occupancy bitfield (custom_functions.py:55-112).
    Returns rays_a (R,3) [ray_idx, start_idx, N_samples], xyzs (S,3), dirs (S,3), deltas (S), ts (S),
    total_samples (0-dim).  Gradients reach the ray origins/directions only (pose optimisation):
    x = o + t d  =>  dL/do = sum_seg dL/dx, dL/dd = sum_seg (t dL/dx + dL/ddir)."""

    @staticmethod
    @_fwd
    def forward(ctx, origins, directions, hits_t, bitfield, cascades, scale, exp_step_factor, grid_size, max_samples):
        jitter = torch.rand_like(origins[:, 0])              # first-sample jitter (custom_functions.py:83)
        packed = vren.raymarching_train(origins, directions, hits_t, bitfield, cascades, scale, exp_step_factor,
                                        jitter, grid_size, max_samples)
        rays_a, xyzs, dirs, deltas, ts, counter = packed
        ctx.mark_non_differentiable(rays_a, deltas, ts)
        ctx.save_for_backward(rays_a, ts)
        return rays_a, xyzs, dirs, deltas, ts, counter[0]

    @staticmethod
    @_bwd
    def backward(ctx, _g_rays_a, g_xyz, g_dir, _g_deltas, _g_ts, _g_total):
        rays_a, ts = ctx.saved_tensors
        g_origin = segment_sum(g_xyz, rays_a)
        g_direction = segment_sum(g_dir + ts.unsqueeze(1) * g_xyz, rays_a)
        return (g_origin, g_direction) + (None,) * 7


class VolumeRenderer(torch.autograd.Function):
    """Front-to-back compositing over packed samples, training only (custom_functions.py:115-159).
    Returns total_samples (0-dim), opacity (R), depth (R), rgb (R,3), ws (S)."""

    @staticmethod
    @_fwd
    def forward(ctx, sigmas, rgbs, deltas, ts, rays_a, T_threshold):
        sigmas, rgbs = sigmas.contiguous(), rgbs.contiguous()
        per_ray_samples, opacity, depth, rgb, ws = vren.composite_train_fw(sigmas, rgbs, deltas, ts, rays_a, T_threshold)
        ctx.T_threshold = T_threshold
        ctx.save_for_backward(sigmas, rgbs, ws, deltas, ts, rays_a, opacity, depth, rgb)
        return per_ray_samples.sum(), opacity, depth, rgb, ws

    @staticmethod
    @_bwd
    def backward(ctx, _g_total, g_opacity, g_depth, g_rgb, g_ws):
        seeds = [g.contiguous() for g in (g_opacity, g_depth, g_rgb, g_ws)]
        g_sigmas, g_rgbs = vren.composite_train_bw(*seeds, *ctx.saved_tensors, ctx.T_threshold)
        return g_sigmas, g_rgbs, None, None, None, None


class TruncExp(torch.autograd.Function):
    """exp whose backward evaluates exp on the input clamped to [-15, 15] (custom_functions.py:162-173)."""

    @staticmethod
    @_fwd
    def forward(ctx, pre_activation):
        ctx.save_for_backward(pre_activation)
        return pre_activation.exp()

    @staticmethod
    @_bwd
    def backward(ctx, g_out):
        (pre_activation,) = ctx.saved_tensors
        return g_out * pre_activation.clamp(min=-15, max=15).exp()
