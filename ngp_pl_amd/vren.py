"""`vren` -- the reference's native op module, same 12 functions, same signatures and returns
(/root/reference/models/csrc/binding.cpp:234-250), backed by libngp_hip.so.

`import ngp_pl_amd.vren as vren` is a drop-in for `import vren` in models/custom_functions.py,
models/rendering.py, models/networks.py and losses.py.  Outputs are freshly allocated torch
tensors on the inputs' device; in-place arguments are the same ones the reference mutates.
Differences that are visible and intended:
  * raymarching_train packs samples in ray order (deterministic) and returns sample tensors
    of exactly counter[0] rows instead of N_rays*max_samples zero-filled rows (slicing them
    with [:counter[0]] as RayMarcher.forward does is a no-op);
  * inputs must be float32 (the autograd wrappers cast, custom_functions.py:27).
"""
import torch

from . import _lib
from ._lib import call, ptr, require_cuda, stream


def _f32(*ts):
    for t in ts:
        if t.dtype != torch.float32:
            raise RuntimeError("expected a float32 tensor, got %s" % t.dtype)


def ray_aabb_intersect(rays_o, rays_d, centers, half_sizes, max_hits):
    require_cuda(rays_o, rays_d, centers, half_sizes); _f32(rays_o, rays_d, centers, half_sizes)
    n, v = rays_o.shape[0], centers.shape[0]
    dev = rays_o.device
    hit_cnt = torch.empty(n, dtype=torch.int32, device=dev)
    hits_t = torch.empty(n, max_hits, 2, dtype=torch.float32, device=dev)
    hits_idx = torch.empty(n, max_hits, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        call("ngp_ray_aabb_intersect", ptr(rays_o), ptr(rays_d), ptr(centers), ptr(half_sizes), n, v, max_hits,
             ptr(hit_cnt), ptr(hits_t), ptr(hits_idx), stream())
    return [hit_cnt, hits_t, hits_idx]


def ray_sphere_intersect(rays_o, rays_d, centers, radii, max_hits):
    require_cuda(rays_o, rays_d, centers, radii); _f32(rays_o, rays_d, centers, radii)
    n, v = rays_o.shape[0], centers.shape[0]
    dev = rays_o.device
    hit_cnt = torch.empty(n, dtype=torch.int32, device=dev)
    hits_t = torch.empty(n, max_hits, 2, dtype=torch.float32, device=dev)
    hits_idx = torch.empty(n, max_hits, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        call("ngp_ray_sphere_intersect", ptr(rays_o), ptr(rays_d), ptr(centers), ptr(radii), n, v, max_hits,
             ptr(hit_cnt), ptr(hits_t), ptr(hits_idx), stream())
    return [hit_cnt, hits_t, hits_idx]


def packbits(density_grid, density_threshold, density_bitfield):
    require_cuda(density_grid, density_bitfield)
    if density_grid.dtype not in (torch.float32, torch.float16):
        raise RuntimeError("density_grid must be float32 or float16")
    with torch.cuda.device(density_grid.device):
        call("ngp_packbits", ptr(density_grid), int(density_grid.dtype == torch.float16), density_bitfield.shape[0],
             float(density_threshold), ptr(density_bitfield), stream())


def morton3D(coords):
    require_cuda(coords)
    if coords.dtype != torch.int32:
        raise RuntimeError("coords must be int32")
    out = torch.empty(coords.shape[0], dtype=torch.int32, device=coords.device)
    with torch.cuda.device(coords.device):
        call("ngp_morton3D", ptr(coords), coords.shape[0], ptr(out), stream())
    return out


def morton3D_invert(indices):
    require_cuda(indices)
    if indices.dtype != torch.int32:
        raise RuntimeError("indices must be int32")
    out = torch.empty(indices.shape[0], 3, dtype=torch.int32, device=indices.device)
    with torch.cuda.device(indices.device):
        call("ngp_morton3D_invert", ptr(indices), indices.shape[0], ptr(out), stream())
    return out


def raymarching_train(rays_o, rays_d, hits_t, density_bitfield, cascades, scale, exp_step_factor, noise,
                      grid_size, max_samples):
    require_cuda(rays_o, rays_d, hits_t, density_bitfield, noise); _f32(rays_o, rays_d, hits_t, noise)
    n = rays_o.shape[0]
    dev = rays_o.device
    rays_a = torch.empty(n, 3, dtype=torch.int64, device=dev)
    counter = torch.empty(2, dtype=torch.int32, device=dev)
    scratch = torch.empty(max(n, 1) * max_samples, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        call("ngp_raymarching_train_count", ptr(rays_o), ptr(rays_d), ptr(hits_t), ptr(density_bitfield), int(cascades),
             float(scale), float(exp_step_factor), ptr(noise), int(grid_size), int(max_samples), n,
             ptr(rays_a), ptr(counter), ptr(scratch), stream())
        S = int(counter[0].item())   # the one host sync of the step (the reference syncs here too)
        xyzs = torch.empty(S, 3, dtype=torch.float32, device=dev)
        dirs = torch.empty(S, 3, dtype=torch.float32, device=dev)
        deltas = torch.empty(S, dtype=torch.float32, device=dev)
        ts = torch.empty(S, dtype=torch.float32, device=dev)
        call("ngp_raymarching_train_write", ptr(rays_o), ptr(rays_d), ptr(rays_a), ptr(scratch), float(scale),
             float(exp_step_factor), int(grid_size), int(max_samples), n, ptr(xyzs), ptr(dirs), ptr(deltas), ptr(ts), stream())
    return [rays_a, xyzs, dirs, deltas, ts, counter]


def raymarching_test(rays_o, rays_d, hits_t, alive_indices, density_bitfield, cascades, scale, exp_step_factor,
                     grid_size, max_samples, N_samples):
    require_cuda(rays_o, rays_d, hits_t, alive_indices, density_bitfield); _f32(rays_o, rays_d, hits_t)
    na = alive_indices.shape[0]
    dev = rays_o.device
    xyzs = torch.empty(na, N_samples, 3, dtype=torch.float32, device=dev)
    dirs = torch.empty(na, N_samples, 3, dtype=torch.float32, device=dev)
    deltas = torch.empty(na, N_samples, dtype=torch.float32, device=dev)
    ts = torch.empty(na, N_samples, dtype=torch.float32, device=dev)
    n_eff = torch.empty(na, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        call("ngp_raymarching_test", ptr(rays_o), ptr(rays_d), ptr(hits_t), ptr(alive_indices), ptr(density_bitfield),
             int(cascades), float(scale), float(exp_step_factor), int(grid_size), int(max_samples), int(N_samples), na,
             ptr(xyzs), ptr(dirs), ptr(deltas), ptr(ts), ptr(n_eff), stream())
    return [xyzs, dirs, deltas, ts, n_eff]


def composite_train_fw(sigmas, rgbs, deltas, ts, rays_a, T_threshold):
    require_cuda(sigmas, rgbs, deltas, ts, rays_a); _f32(sigmas, rgbs, deltas, ts)
    R, S = rays_a.shape[0], sigmas.shape[0]
    dev = sigmas.device
    total = torch.empty(R, dtype=torch.int64, device=dev)
    opacity = torch.empty(R, dtype=torch.float32, device=dev)
    depth = torch.empty(R, dtype=torch.float32, device=dev)
    rgb = torch.empty(R, 3, dtype=torch.float32, device=dev)
    ws = torch.empty(S, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        call("ngp_composite_train_fw", ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(ts), ptr(rays_a), float(T_threshold), R, S,
             ptr(total), ptr(opacity), ptr(depth), ptr(rgb), ptr(ws), None, stream())
    return [total, opacity, depth, rgb, ws]


def composite_train_bw(dL_dopacity, dL_ddepth, dL_drgb, dL_dws, sigmas, rgbs, ws, deltas, ts, rays_a,
                       opacity, depth, rgb, T_threshold):
    args = (dL_dopacity, dL_ddepth, dL_drgb, dL_dws, sigmas, rgbs, ws, deltas, ts, rays_a, opacity, depth, rgb)
    require_cuda(*args)
    R, S = rays_a.shape[0], sigmas.shape[0]
    dev = sigmas.device
    dsig = torch.empty(S, dtype=torch.float32, device=dev)
    drgbs = torch.empty(S, 3, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        call("ngp_composite_train_bw", *[ptr(a) for a in args], float(T_threshold), R, S, ptr(dsig), ptr(drgbs), None, None, None, None, stream())
    return [dsig, drgbs]


def composite_test_fw(sigmas, rgbs, deltas, ts, hits_t, alive_indices, T_threshold, N_eff_samples,
                      opacity, depth, rgb):
    require_cuda(sigmas, rgbs, deltas, ts, alive_indices, N_eff_samples, opacity, depth, rgb)
    na, ns = sigmas.shape
    with torch.cuda.device(sigmas.device):
        call("ngp_composite_test_fw", ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(ts), ptr(alive_indices), float(T_threshold),
             ptr(N_eff_samples), na, ns, ptr(opacity), ptr(depth), ptr(rgb), stream())


def distortion_loss_fw(ws, deltas, ts, rays_a):
    require_cuda(ws, deltas, ts, rays_a)
    R, S = rays_a.shape[0], ws.shape[0]
    dev = ws.device
    loss = torch.empty(R, dtype=torch.float32, device=dev)
    a = torch.empty(S, dtype=torch.float32, device=dev)
    b = torch.empty(S, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        call("ngp_distortion_loss_fw", ptr(ws), ptr(deltas), ptr(ts), ptr(rays_a), R, S, ptr(loss), ptr(a), ptr(b), stream())
    return [loss, a, b]


def distortion_loss_bw(dL_dloss, ws_inclusive_scan, wts_inclusive_scan, ws, deltas, ts, rays_a):
    require_cuda(dL_dloss, ws_inclusive_scan, wts_inclusive_scan, ws, deltas, ts, rays_a)
    R, S = rays_a.shape[0], ws.shape[0]
    out = torch.zeros(S, dtype=torch.float32, device=ws.device)
    dL_dloss = dL_dloss.contiguous()           # (require_cuda has checked it already; held by name for the duration of the launch)
    with torch.cuda.device(ws.device):
        call("ngp_distortion_loss_bw", ptr(dL_dloss), ptr(ws_inclusive_scan), ptr(wts_inclusive_scan), ptr(ws),
             ptr(deltas), ptr(ts), ptr(rays_a), R, S, ptr(out), stream())
    return out
