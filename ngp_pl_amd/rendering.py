"""`render()` of the reference (/root/reference/models/rendering.py:11-163): same signature, same
kwargs (test_time, exp_step_factor, T_threshold, max_samples, random_bg, exposure,
output_radiance, to_cpu, to_numpy), same result dictionary.
"""
import ctypes as C

import torch

from . import _lib, vren
from ._lib import call, ptr, stream
from .custom_functions import RayAABBIntersector, RayMarcher, VolumeRenderer

from .stepper import MAX_SAMPLES        # noqa: E402  (1024, rendering.py:7)
NEAR_DISTANCE = 0.01


_BG = {}


def _background(exp_step_factor, device, random_bg=False):
    """white for synthetic scenes (rendering.py:111-112,153-154), random or black for real ones; the two constant colours are
    cached per device (a torch.ones per call is a kernel launch)."""
    if exp_step_factor != 0 and random_bg:
        return torch.rand(3, device=device)
    key = (exp_step_factor == 0, str(device))
    bg = _BG.get(key)
    if bg is None:
        bg = _BG[key] = torch.ones(3, device=device) if exp_step_factor == 0 else torch.zeros(3, device=device)
    return bg


@torch.autocast("cuda")
def render(model, rays_o, rays_d, **kwargs):
    """rays (R,3)/(R,3) -> dict with rgb (R,3), depth (R), opacity (R) and, in training, ws, deltas,
    ts, rays_a, rm_samples, vr_samples; at test time total_samples (rendering.py:11-43)."""
    rays_o = rays_o.contiguous(); rays_d = rays_d.contiguous()
    test_time = kwargs.get("test_time", False)
    native_test = test_time and getattr(model, "fused", False) and model.rgb_act == "Sigmoid" and rays_o.is_cuda and \
        not any(isinstance(v, torch.Tensor) for v in kwargs.values())
    if native_test or (not test_time and _fused_train(model, rays_o, rays_d, kwargs)):
        rays_o, rays_d = rays_o.float(), rays_d.float()
        hits_t = None
        if test_time or not _native_train(model, kwargs):       # (the native stepper's march starts with this prologue itself)
            # rendering.py:27-29 (one box, one hit, near clamp) as ONE launch; (R,1,2) like the operator's output.  (The operator
            # chain below costs four launches and -- the boolean-mask assignment goes through nonzero() -- a host synchronisation
            # per call: 3 % of a 1.5 ms frame on the trained field.)
            hits_t = torch.empty(rays_o.shape[0], 1, 2, dtype=torch.float32, device=rays_o.device)
            with _lib.device_guard(rays_o.device):
                call("ngp_ray_aabb_near", ptr(rays_o), ptr(rays_d), ptr(model.center), ptr(model.half_size), NEAR_DISTANCE,
                     rays_o.shape[0], ptr(hits_t), stream())
        if test_time:
            fn = _render_test_native if kwargs.get("host_loop", False) else _render_test_device
        else:
            fn = _render_train
    else:
        _, hits_t, _ = RayAABBIntersector.apply(rays_o, rays_d, model.center, model.half_size, 1)
        t1 = hits_t[:, 0, 0]
        hits_t[(t1 >= 0) & (t1 < NEAR_DISTANCE), 0, 0] = NEAR_DISTANCE
        fn = _render_test if test_time else _render_train
    results = fn(model, rays_o, rays_d, hits_t, **kwargs)
    if kwargs.get("to_cpu", False):
        for k, v in results.items():
            if torch.is_tensor(v):
                v = v.cpu()
                if kwargs.get("to_numpy", False):
                    v = v.numpy()
            results[k] = v
    return results


@torch.no_grad()
def _render_test(model, rays_o, rays_d, hits_t, **kwargs):
    """Iterative march / infer / composite with alive-ray compaction (rendering.py:46-118)."""
    esf = kwargs.get("exp_step_factor", 0.)
    n_rays, device = len(rays_o), rays_o.device
    opacity = torch.zeros(n_rays, device=device)
    depth = torch.zeros(n_rays, device=device)
    rgb = torch.zeros(n_rays, 3, device=device)
    hits = hits_t[:, 0].contiguous()          # (R,2), advanced in place by the marcher
    alive = torch.arange(n_rays, device=device)
    min_samples = 1 if esf == 0 else 4
    samples = 0
    total_samples = 0
    max_samples = kwargs.get("max_samples", MAX_SAMPLES)
    T_threshold = kwargs.get("T_threshold", 1e-4)
    while samples < max_samples:
        n_alive = len(alive)
        if n_alive == 0:
            break
        n_step = max(min(n_rays // n_alive, 64), min_samples)
        samples += n_step
        xyzs, dirs, deltas, ts, n_eff = vren.raymarching_test(
            rays_o, rays_d, hits, alive, model.density_bitfield, model.cascades, model.scale, esf,
            model.grid_size, MAX_SAMPLES, n_step)
        total_samples += n_eff.sum()
        xyzs = xyzs.view(-1, 3); dirs = dirs.view(-1, 3)
        valid = ~torch.all(dirs == 0, dim=1)
        if valid.sum() == 0:
            break
        sigmas = torch.zeros(len(xyzs), device=device)
        rgbs = torch.zeros(len(xyzs), 3, device=device)
        s, c = model(xyzs[valid], dirs[valid], **kwargs)
        sigmas[valid] = s.float(); rgbs[valid] = c.float()
        vren.composite_test_fw(sigmas.view(-1, n_step), rgbs.view(-1, n_step, 3), deltas, ts, hits, alive,
                               T_threshold, n_eff, opacity, depth, rgb)
        alive = alive[alive >= 0]
    bg = _background(esf, device)
    return {"opacity": opacity, "depth": depth, "rgb": rgb + bg * (1 - opacity)[:, None], "total_samples": total_samples}


@torch.no_grad()
def _render_test_native(model, rays_o, rays_d, hits_t, **kwargs):
    """The same loop as `_render_test` (rendering.py:46-118: same N_samples schedule, same kernels'
    arithmetic, same results) with the torch glue removed: no boolean masks (the field is evaluated
    on the zero-padded slots too and composite_test only reads the first N_eff of a ray), alive-ray
    compaction and the sample count on device, one 4-byte host read per iteration instead of two
    syncs and ~60 small launches."""
    esf = kwargs.get("exp_step_factor", 0.)
    n_rays, dev = len(rays_o), rays_o.device
    opacity = torch.zeros(n_rays, device=dev); depth = torch.zeros(n_rays, device=dev); rgb = torch.zeros(n_rays, 3, device=dev)
    hits = hits_t[:, 0].contiguous()
    alive = torch.arange(n_rays, device=dev)
    min_samples = 1 if esf == 0 else 4
    max_samples = kwargs.get("max_samples", MAX_SAMPLES)
    T_threshold = kwargs.get("T_threshold", 1e-4)
    enc, net = model.xyz_encoder, model.rgb_net
    eh, rh = enc._half.get(enc.params), net._half.get(net.params)
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    count_host = torch.empty(1, dtype=torch.int32, pin_memory=True)
    samples, n_alive = 0, n_rays
    f32 = dict(dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        while samples < max_samples and n_alive > 0:
            N = max(min(n_rays // n_alive, 64), min_samples)
            samples += N
            M = n_alive * N
            xyzs = torch.empty(M, 3, **f32); dirs = torch.empty(M, 3, **f32); deltas = torch.empty(M, **f32); ts = torch.empty(M, **f32)
            n_eff = torch.empty(n_alive, dtype=torch.int32, device=dev)
            call("ngp_raymarching_test", ptr(rays_o), ptr(rays_d), ptr(hits), ptr(alive), ptr(model.density_bitfield), model.cascades,
                 float(model.scale), float(esf), model.grid_size, MAX_SAMPLES, N, n_alive, ptr(xyzs), ptr(dirs), ptr(deltas), ptr(ts),
                 ptr(n_eff), stream())
            feats = torch.empty(16, M, 2, dtype=torch.float16, device=dev)
            sigmas = torch.empty(M, **f32); rgbs = torch.empty(M, 3, **f32)
            call("ngp_hashgrid_fwd", ptr(xyzs), ptr(model.xyz_min), ptr(model.xyz_max), ptr(eh[enc.n_mlp:]), C.byref(enc.meta), M, ptr(feats), stream())
            call("ngp_field_fwd", ptr(feats), ptr(dirs), ptr(eh), ptr(rh), M, ptr(sigmas), ptr(rgbs), None, stream())       # inference: h stays in registers
            call("ngp_composite_test_fw", ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(ts), ptr(alive), float(T_threshold), ptr(n_eff), n_alive, N,
                 ptr(opacity), ptr(depth), ptr(rgb), stream())
            alive_new = torch.empty(n_alive, dtype=torch.int64, device=dev)
            count = torch.zeros(1, dtype=torch.int32, device=dev)
            call("ngp_compact_alive", ptr(alive), ptr(n_eff), n_alive, ptr(alive_new), ptr(count), ptr(total), stream())
            count_host.copy_(count, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            n_alive = int(count_host[0])
            alive = alive_new[:n_alive]
    bg = _background(esf, dev)
    return {"opacity": opacity, "depth": depth, "rgb": rgb + bg * (1 - opacity)[:, None], "total_samples": total[0]}


@torch.no_grad()
def _render_test_device(model, rays_o, rays_d, hits_t, **kwargs):
    """`_render_test` as ONE call into the library (`ngp_render_test_frame`): the loop of
    rendering.py:46-118 with the alive count, N_samples and the batch sizes kept on the device, so
    no iteration waits for the host.  `chunk_scale=1, probe_cap=0` (default) keeps the reference's
    chunking and is bit-identical to `_render_test_native`; larger `chunk_scale` / a `probe_cap`
    emit the same samples per ray in fewer, better balanced iterations (results equal up to the
    float rounding of where the composite re-reads T = 1 - opacity)."""
    esf = float(kwargs.get("exp_step_factor", 0.))
    n_rays, dev = len(rays_o), rays_o.device
    chunk_scale = int(kwargs.get("chunk_scale", 1))
    probe_cap = int(kwargs.get("probe_cap", 0))
    enc, net = model.xyz_encoder, model.rgb_net
    eh, rh = enc._half.get(enc.params), net._half.get(net.params)
    nbytes = _lib.lib().ngp_render_test_workspace_bytes(n_rays, chunk_scale, esf)
    ws = getattr(model, "_render_ws", None)
    if ws is None or ws.numel() < nbytes or ws.device != dev:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        model._render_ws = ws
    opacity = torch.empty(n_rays, device=dev); depth = torch.empty(n_rays, device=dev); rgb = torch.empty(n_rays, 3, device=dev)
    total = torch.empty(1, dtype=torch.int64, device=dev)
    hits = hits_t[:, 0].contiguous()
    bg = (C.c_float * 3)(*([1.0, 1.0, 1.0] if esf == 0 else [0.0, 0.0, 0.0]))     # rendering.py:112-116
    n_it = C.c_int32(0)
    with _lib.device_guard(dev):
        call("ngp_render_test_frame", ptr(rays_o), ptr(rays_d), ptr(hits), ptr(model.density_bitfield), model.cascades,
             float(model.scale), esf, model.grid_size, int(kwargs.get("max_samples", MAX_SAMPLES)), float(kwargs.get("T_threshold", 1e-4)),
             ptr(model.xyz_min), ptr(model.xyz_max), ptr(eh[enc.n_mlp:]), C.byref(enc.meta), ptr(eh), ptr(rh),
             n_rays, chunk_scale, probe_cap, bg, ptr(ws), nbytes, ptr(opacity), ptr(depth), ptr(rgb), ptr(total),
             C.byref(n_it), stream())
    return {"opacity": opacity, "depth": depth, "rgb": rgb, "total_samples": total[0], "n_iterations": n_it.value}


class _FusedTrainRender(torch.autograd.Function):
    """`__render_rays_train` (rendering.py:121-163) as ONE autograd node when the model is in its fused configuration:
    march -> hash grid -> MLPs -> composite -> background blend in the forward; in the backward the composite gradient,
    then the field backward and the table backward on the ACTIVE samples only (those up to each ray's early stop; the rest
    have exactly zero gradient) -- the same kernels in the same order as the native step (trainer.Trainer.step), so that
    the reference-shaped surface render() + NeRFLoss + autograd + FusedAdam runs what `Trainer.step` runs.  Results and
    gradients equal the node-by-node path (RayMarcher -> NGP.forward -> VolumeRenderer) up to the summation order of
    the table gradient (tests/test_train_gpu.py::test_fused_render_node_matches_the_operator_chain)."""

    @staticmethod
    def forward(ctx, enc_params, rgb_params, model, rays_o, rays_d, hits_t, esf, T_threshold, bg, noise=None):
        from . import tcnn
        n, dev = rays_o.shape[0], rays_o.device
        enc, net = model.xyz_encoder, model.rgb_net
        f32 = dict(dtype=torch.float32, device=dev); f16 = dict(dtype=torch.float16, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            sq = stream()
            if noise is None:
                noise = torch.rand(n, **f32)                          # custom_functions.py:83
            else:                                                     # caller-supplied jitter (tests: the same draw as another path)
                noise = noise.to(**f32).contiguous()
            rays_a = torch.empty(n, 3, dtype=torch.int64, device=dev)
            scratch = torch.empty(max(n, 1) * MAX_SAMPLES, **f32)
            counter = _pinned_counter()
            counter.np[0] = -1
            call("ngp_raymarching_train_count", ptr(rays_o), ptr(rays_d), ptr(hits_t), ptr(model.density_bitfield), model.cascades,
                 float(model.scale), float(esf), ptr(noise), model.grid_size, MAX_SAMPLES, n, ptr(rays_a), counter.ptr, ptr(scratch), sq)
            eh, rh = enc._half.get(enc_params), net._half.get(rgb_params)      # the casts overlap the march on the device
            done = torch.cuda.Event(); done.record()
            # the reference syncs here too (raymarching.cu:298: counter.item()); bounded poll
            _lib.poll_event(done, "ngp_raymarching_train_count inside render() (%d rays)" % n)
            S = int(counter.np[0])
            xyzs = torch.empty(S, 3, **f32); dirs = torch.empty(S, 3, **f32); deltas = torch.empty(S, **f32); ts = torch.empty(S, **f32)
            call("ngp_raymarching_train_write", ptr(rays_o), ptr(rays_d), ptr(rays_a), ptr(scratch), float(model.scale), float(esf),
                 model.grid_size, MAX_SAMPLES, n, ptr(xyzs), ptr(dirs), ptr(deltas), ptr(ts), sq)
            feats = torch.empty(16, S, 2, **f16)
            h = torch.empty(S, 16, **f16) if _lib.field_bwd_uses_h() else None      # (the one-launch backward recomputes it)
            sigmas = torch.empty(S, **f32); rgbs = torch.empty(S, 3, **f32)
            if S > 0:
                call("ngp_hashgrid_fwd", ptr(xyzs), ptr(model.xyz_min), ptr(model.xyz_max), ptr(eh[enc.n_mlp:]), C.byref(enc.meta), S, ptr(feats), sq)
                call("ngp_field_fwd", ptr(feats), ptr(dirs), ptr(eh), ptr(rh), S, ptr(sigmas), ptr(rgbs), ptr(h), sq)
            total = torch.empty(n, dtype=torch.int64, device=dev)
            opacity = torch.empty(n, **f32); depth = torch.empty(n, **f32); rgb = torch.empty(n, 3, **f32); ws = torch.empty(S, **f32)
            ray_offs = torch.empty(n, **i32); n_active = torch.empty(1, **i32)
            call("ngp_composite_train_fw", ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(ts), ptr(rays_a), float(T_threshold), n, S,
                 ptr(total), ptr(opacity), ptr(depth), ptr(rgb), ptr(ws), ptr(ray_offs), sq)
            call("ngp_active_scan", ptr(ray_offs), n, ptr(n_active), sq)
            rgb_out = torch.empty(n, 3, **f32)
            call("ngp_bg_blend", ptr(rgb), ptr(opacity), ptr(bg), n, ptr(rgb_out), sq)       # rendering.py:161
        ctx.model, ctx.S, ctx.T_threshold = model, S, T_threshold
        ctx.save_for_backward(rays_a, xyzs, dirs, deltas, ts, feats, h, sigmas, rgbs, ws, opacity, depth, rgb, ray_offs, n_active, bg)
        vr_samples = total.sum()
        rm_samples = torch.tensor(S, dtype=torch.int32)
        ctx.mark_non_differentiable(rays_a, deltas, ts, vr_samples, rm_samples)
        return vr_samples, opacity, depth, rgb_out, ws, rays_a, deltas, ts, rm_samples

    @staticmethod
    def backward(ctx, _g_vr, g_opacity, g_depth, g_rgb, g_ws, _g_rays_a, _g_deltas, _g_ts, _g_rm):
        from . import tcnn
        model, S = ctx.model, ctx.S
        rays_a, xyzs, dirs, deltas, ts, feats, h, sigmas, rgbs, ws, opacity, depth, rgb, ray_offs, n_active, bg = ctx.saved_tensors
        enc, net = model.xyz_encoder, model.rgb_net
        n, dev = rays_a.shape[0], rays_a.device
        none5 = (None,) * 8
        if S == 0:
            if model.native_grads:
                return (None, None) + none5
            return (torch.zeros_like(enc.params), torch.zeros_like(net.params)) + none5
        f32 = dict(dtype=torch.float32, device=dev); f16 = dict(dtype=torch.float16, device=dev)
        g_rgb = torch.zeros(n, 3, **f32) if g_rgb is None else g_rgb.float().contiguous()
        g_opacity_in = None if g_opacity is None else g_opacity.float().contiguous()
        g_opacity = torch.empty(n, **f32)
        with torch.cuda.device(dev):
            call("ngp_bg_blend_bw", ptr(g_rgb), ptr(g_opacity_in), ptr(bg), n, ptr(g_opacity), stream())     # through rgb + bg (1 - opacity)
        g_depth = _zeros(n, dev) if g_depth is None else g_depth.float().contiguous()
        g_ws = None if g_ws is None else g_ws.float().contiguous()
        scale = tcnn.LOSS_SCALE
        lib = _lib.lib()
        with torch.cuda.device(dev):
            sq = stream()
            eh, rh = enc._half.get(enc.params), net._half.get(net.params)
            nbytes = lib.ngp_hashgrid_bwd_binned_workspace_bytes(C.byref(enc.meta), S) if tcnn.binned_enabled() else 0
            dL_dsigmas = torch.empty(S, **f32); dL_drgbs = torch.empty(S, 3, **f32)
            active = torch.empty(S, dtype=torch.int32, device=dev)
            x_act = torch.empty(S, 3, **f32) if nbytes else None
            call("ngp_composite_train_bw", ptr(g_opacity), ptr(g_depth), ptr(g_rgb), ptr(g_ws), ptr(sigmas), ptr(rgbs), ptr(ws), ptr(deltas),
                 ptr(ts), ptr(rays_a), ptr(opacity), ptr(depth), ptr(rgb), float(ctx.T_threshold), n, S, ptr(dL_dsigmas), ptr(dL_drgbs),
                 ptr(ray_offs), ptr(active), ptr(xyzs) if nbytes else None, ptr(x_act), sq)
            n_part = call("ngp_field_bwd_partials", S)
            partials = torch.empty(n_part * (enc.n_mlp + net.params.numel()), **f32)
            dh = torch.empty(S, 16, **f16) if h is not None else None
            dfeats = torch.empty(16, S, 2, **f16)
            call("ngp_field_bwd", ptr(feats), ptr(dirs), ptr(h), ptr(eh), ptr(rh), ptr(dL_dsigmas), ptr(dL_drgbs), scale, S,
                 ptr(active), ptr(n_active), ptr(dh), ptr(dfeats), ptr(partials), sq)
            g16 = model._grid_grad16(dev)
            if nbytes:
                bin_ws = tcnn.binned_workspace(dev, nbytes)
                call("ngp_hashgrid_bwd_binned", ptr(x_act), ptr(model.xyz_min), ptr(model.xyz_max), ptr(dfeats), C.byref(enc.meta), S,
                     None, ptr(n_active), ptr(bin_ws), bin_ws.numel(), ptr(g16), sq)
            else:
                call("ngp_hashgrid_bwd_sliced", ptr(xyzs), ptr(model.xyz_min), ptr(model.xyz_max), ptr(dfeats), C.byref(enc.meta), S,
                     ptr(active), ptr(n_active), ptr(g16), sq)
            p_density = partials[:n_part * enc.n_mlp]; p_rgb = partials[n_part * enc.n_mlp:]
            if model.native_grads:
                model.hand_over_native(dict(grid16=g16, density_partials=p_density, rgb_partials=p_rgb, n_partials=n_part, scale=scale))
                return (None, None) + none5
            g_enc = torch.empty_like(enc.params)
            g_enc[:enc.n_mlp] = tcnn.reduce_partials(p_density, n_part, enc.n_mlp) / scale
            call("ngp_cast_f16_to_f32", ptr(g16), enc.n_grid, 1.0 / scale, ptr(g_enc[enc.n_mlp:]), sq)
            g_rgbw = tcnn.reduce_partials(p_rgb, n_part, net.params.numel()) / scale
        return (g_enc, g_rgbw) + none5


class _NativeTrainRender(torch.autograd.Function):
    """`__render_rays_train` (rendering.py:121-163) through the native stepper (csrc/stepper.hip): the forward half of the step
    is ONE library call (AABB + jitter + march, sample expansion, hash grid, field, composite, live-sample offsets, background
    blend), the backward half two (composite + field backward on the live samples; table backward) -- the launch sequence of
    `Trainer.step`, split where the reference's API splits it.  Used when the model is in its fused configuration AND leaves its
    gradients in native buffers for optim.FusedAdam (model.native_grads: one backward per forward).  The returned per-ray /
    per-sample tensors are views of the model's step buffers: valid until its next training-branch render()."""

    @staticmethod
    def forward(ctx, enc_params, rgb_params, model, rays_o, rays_d, esf, T_threshold, bg, next_rays):
        from .stepper import RenderStepper
        n, dev = rays_o.shape[0], rays_o.device
        rs = getattr(model, "_render_stepper", None)
        if rs is None:
            rs = model._render_stepper = RenderStepper(model)
        with _lib.device_guard(dev):
            B, h = rs.prepare(n, esf, T_threshold, bg)
            mq = stream()
            sq = mq
            ro_p, rd_p = rays_o.data_ptr(), rays_d.data_ptr()
            if rs.pending is None or rs.pending[:2] != (ro_p, rd_p):
                if rs.pending is not None:
                    call("ngp_stepper_drop_pending", h)
                call("ngp_stepper_march", h, ro_p, rd_p, mq, mq)
            rs.pending = None
            no_p = nd_p = None
            if next_rays is not None and next_rays[0].shape[0] == n:       # the caller knows its next batch: march it under this step
                if rs.side is None:
                    from .trainer import marching_stream
                    rs.side = marching_stream(dev)
                sq = rs.side.cuda_stream
                nxt = (next_rays[0].float().contiguous(), next_rays[1].float().contiguous())
                no_p, nd_p = nxt[0].data_ptr(), nxt[1].data_ptr()
            S_c = C.c_int32(0)
            call("ngp_stepper_render_forward", h, ro_p, rd_p, no_p, nd_p, B.p["rgb_out"], mq, sq, C.byref(S_c))
            if no_p is not None:
                rs.pending = (no_p, nd_p, nxt)
            S = S_c.value
            k = call("ngp_stepper_last_set", h)
        rs.generation += 1
        ctx.model, ctx.rs, ctx.generation, ctx.S = model, rs, rs.generation, S
        f32 = torch.float32
        opacity, depth = B.opacity, B.depth
        rgb_out = B.fixed("rgb_out", f32, n, 3)
        ws, deltas, ts = B.prefix("ws", f32, S), B.prefix(B.sample_name("deltas", k), f32, S), B.prefix(B.sample_name("ts", k), f32, S)
        rays_a = B.fixed("rays_a%d" % k, torch.int64, n, 3)
        vr_samples = B.total.sum()
        rm_samples = torch.tensor(S, dtype=torch.int32)
        ctx.mark_non_differentiable(rays_a, deltas, ts, vr_samples, rm_samples)
        return vr_samples, opacity, depth, rgb_out, ws, rays_a, deltas, ts, rm_samples

    @staticmethod
    def backward(ctx, _g_vr, g_opacity, g_depth, g_rgb, g_ws, _g_rays_a, _g_deltas, _g_ts, _g_rm):
        from . import tcnn
        model, rs, S = ctx.model, ctx.rs, ctx.S
        none7 = (None,) * 7
        if ctx.generation != rs.generation:
            raise RuntimeError("backward of a render() whose step buffers have been reused by a later training-branch render() of the "
                               "same model: with model.native_grads every forward needs its backward before the next one "
                               "(set model.native_grads = False for several forwards per backward)")
        enc, net = model.xyz_encoder, model.rgb_net
        if S == 0:
            return (None, None) + none7
        B, h = rs.buf, rs.handle
        n, dev = B.n, B.arena.device
        f32 = dict(dtype=torch.float32, device=dev)
        g_rgb = torch.zeros(n, 3, **f32) if g_rgb is None else g_rgb.float().contiguous()
        g_opacity = None if g_opacity is None else g_opacity.float().contiguous()
        g_depth = None if g_depth is None else g_depth.float().contiguous()
        g_ws = None if g_ws is None else g_ws.float().contiguous()
        np_c = C.c_int32(0)
        with _lib.device_guard(dev):
            mq = stream()
            call("ngp_stepper_render_backward", h, ptr(g_rgb), ptr(g_opacity), ptr(g_depth), ptr(g_ws), tcnn.LOSS_SCALE, mq, C.byref(np_c))
            call("ngp_stepper_table_backward", h, 1, 0, mq)
        n_part = np_c.value
        g16 = model._grid_grad16(dev)
        p_density, p_rgb = B.partial_rows(n_part, enc.n_mlp)
        model.hand_over_native(dict(grid16=g16, density_partials=p_density, rgb_partials=p_rgb, n_partials=n_part, scale=tcnn.LOSS_SCALE,
                                    stepper=h))          # (FusedAdam asks it for the step's overflow flag and the dynamic loss scale)
        return (None, None) + none7


_ZEROS = {}


def _zeros(n, dev):
    """A cached all-zero (n) f32 seed (dL/ddepth when the loss has no depth term): read-only by convention."""
    z = _ZEROS.get((n, dev))
    if z is None:
        z = _ZEROS[(n, dev)] = torch.zeros(n, dtype=torch.float32, device=dev)
    return z


class _PinnedCounter:
    def __init__(self):
        self.t = torch.zeros(2, dtype=torch.int32).pin_memory()
        self.np = self.t.numpy()
        self.ptr = self.t.data_ptr()


_COUNTER = None


def _pinned_counter():
    """{S, R} of the march lands in pinned (device-mapped) host memory: no copy kernel between the march and the host read."""
    global _COUNTER
    if _COUNTER is None:
        _COUNTER = _PinnedCounter()
    return _COUNTER


def _fused_train(model, rays_o, rays_d, kwargs):
    """The training branch runs as one autograd node when the model is in its fused configuration, the rays carry no gradient
    (no pose optimisation) and no per-ray tensor kwargs (exposure) have to be expanded per sample."""
    return getattr(model, "fused", False) and getattr(model, "fused_render", True) and model.rgb_act == "Sigmoid" and rays_o.is_cuda and \
        not (torch.is_grad_enabled() and (rays_o.requires_grad or rays_d.requires_grad)) and \
        not any(isinstance(v, torch.Tensor) for k, v in kwargs.items() if k != "noise")


def _native_train(model, kwargs):
    """The training branch goes through the native stepper when the gradients stay in native buffers for optim.FusedAdam (one
    backward per forward, see _NativeTrainRender), autograd is recording, and the caller supplies no jitter of its own.
    NGP_NATIVE_RENDER=0 keeps the launch-by-launch node."""
    import os
    return getattr(model, "native_grads", False) and torch.is_grad_enabled() and "noise" not in kwargs and \
        os.environ.get("NGP_NATIVE_RENDER", "1") != "0"


def _render_train(model, rays_o, rays_d, hits_t, **kwargs):
    """march -> field -> composite (rendering.py:121-163)."""
    esf = kwargs.get("exp_step_factor", 0.)
    results = {}
    if _fused_train(model, rays_o, rays_d, kwargs):
        bg = _background(esf, rays_o.device, kwargs.get("random_bg", False))
        if _native_train(model, kwargs):
            (results["vr_samples"], results["opacity"], results["depth"], results["rgb"], results["ws"], results["rays_a"],
             results["deltas"], results["ts"], results["rm_samples"]) = _NativeTrainRender.apply(
                model.xyz_encoder.params, model.rgb_net.params, model, rays_o.float().contiguous(), rays_d.float().contiguous(), esf,
                kwargs.get("T_threshold", 1e-4), bg, kwargs.get("next_rays"))
            return results
        (results["vr_samples"], results["opacity"], results["depth"], results["rgb"], results["ws"], results["rays_a"],
         results["deltas"], results["ts"], results["rm_samples"]) = _FusedTrainRender.apply(
            model.xyz_encoder.params, model.rgb_net.params, model, rays_o.float(), rays_d.float(), hits_t[:, 0].contiguous(), esf,
            kwargs.get("T_threshold", 1e-4), bg, kwargs.get("noise"))
        return results
    rays_a, xyzs, dirs, results["deltas"], results["ts"], results["rm_samples"] = RayMarcher.apply(
        rays_o, rays_d, hits_t[:, 0], model.density_bitfield, model.cascades, model.scale, esf,
        model.grid_size, MAX_SAMPLES)
    for k, v in kwargs.items():              # per-ray tensors (e.g. exposure) repeated per sample
        if isinstance(v, torch.Tensor):
            kwargs[k] = torch.repeat_interleave(v[rays_a[:, 0]], rays_a[:, 2], 0)
    sigmas, rgbs = model(xyzs, dirs, **kwargs)
    (results["vr_samples"], results["opacity"], results["depth"], results["rgb"], results["ws"]) = VolumeRenderer.apply(
        sigmas, rgbs.contiguous(), results["deltas"], results["ts"], rays_a, kwargs.get("T_threshold", 1e-4))
    results["rays_a"] = rays_a
    bg = _background(esf, rays_o.device, kwargs.get("random_bg", False))
    results["rgb"] = results["rgb"] + bg * (1 - results["opacity"])[:, None]
    return results
