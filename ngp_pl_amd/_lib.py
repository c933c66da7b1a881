"""ctypes binding of libngp_hip.so (C ABI: include/ngp_hip.h).

The product path has NO fallback: if the library is missing or a call fails this raises.
torch is imported first so that the library binds to the HIP runtime torch already loaded
(same SONAME libamdhip64.so.7) and torch's streams/pointers are valid inside it.
"""
import contextlib
import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NGP_HIP_LIB") or os.path.join(_HERE, "csrc", "libngp_hip.so")   # override: A/B builds while profiling

P = C.c_void_p
I = C.c_int
F = C.c_float
L = C.c_int64

NGP_MAX_LEVELS = 16


class GridMeta(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("n_features", C.c_int32),
                ("offset", C.c_uint32 * (NGP_MAX_LEVELS + 1)),
                ("resolution", C.c_uint32 * NGP_MAX_LEVELS),
                ("scale", C.c_float * NGP_MAX_LEVELS)]


class StepperConfig(C.Structure):
    """ngp_stepper_config (include/ngp_hip.h)."""
    _fields_ = [("center", P), ("half_size", P), ("xyz_min", P), ("xyz_max", P), ("density_bitfield", P),
                ("cascades", C.c_int32), ("grid_size", C.c_int32), ("scale", F), ("exp_step_factor", F), ("meta", GridMeta),
                ("enc_param", P), ("enc_half", P), ("enc_m", P), ("enc_v", P),
                ("rgb_param", P), ("rgb_half", P), ("rgb_m", P), ("rgb_v", P),
                ("n_grid", C.c_int64), ("n_density", C.c_int32), ("n_rgb", C.c_int32), ("grid_grad16", P),
                ("max_samples", C.c_int32), ("near_distance", F), ("T_threshold", F), ("lambda_opacity", F), ("lambda_distortion", F),
                ("bg", P), ("beta1", F), ("beta2", F), ("eps", F), ("weight_decay", F), ("noise_seed", C.c_uint64)]


class StepBuffersC(C.Structure):
    """ngp_step_buffers (include/ngp_hip.h)."""
    _fields_ = [("n_rays", C.c_int32), ("distortion", C.c_int32), ("cap", C.c_int64)] + \
        [(k, P) for k in ("xyzs", "dirs", "deltas", "ts", "feats", "h", "sigmas", "rgbs", "ws", "dL_dsigmas", "dL_drgbs", "active", "x_act",
                          "dh", "dfeats", "ws_incl", "wts_incl", "dL_dws",
                          "total", "opacity", "depth", "rgb", "dL_drgb", "dL_dopacity", "ray_offs", "dist", "zeros", "dist_seed")] + \
        [("hits_t", P * 2), ("rays_a", P * 2), ("noise", P * 2), ("scratch", P * 2), ("counter", P * 2),
         ("list_k", P), ("list_rest", P), ("two_round_counts", P), ("offs_k", P * 2),
         ("n_active", P), ("stats", P), ("partials", P), ("max_partials", C.c_int32),
         ("fw_ws", P), ("fw_bytes", C.c_size_t), ("bin_ws", P), ("bin_bytes", C.c_size_t), ("bin_max", C.c_int32)]


class GridPartials(C.Structure):
    """ngp_grid_partials (include/ngp_hip.h)."""
    _fields_ = [("n_levels", C.c_int32), ("reserved", C.c_int32), ("value_end", C.c_int64), ("offset", C.c_uint32 * (NGP_MAX_LEVELS + 1)),
                ("k_split", C.c_int32 * NGP_MAX_LEVELS), ("part_off", C.c_int64 * NGP_MAX_LEVELS), ("partial", P)]


ABI_VERSION = 6              # include/ngp_hip.h: ngp_abi_version()


class ExchangeConfig(C.Structure):
    """ngp_exchange_config (include/ngp_hip.h)."""
    _fields_ = [("mode", C.c_int32), ("n_chunks", C.c_int32), ("n_groups", C.c_int32), ("reserved", C.c_int32), ("piece", C.c_int64),
                ("grad_padded", P), ("table_padded", P), ("shard16", P), ("small", P), ("flags", P), ("step_state", P), ("stage", P)]


# name -> argtypes (every function returns int, except the two queries noted below)
_PROTOS = {
    "ngp_ray_aabb_intersect": [P, P, P, P, I, I, I, P, P, P, P],
    "ngp_ray_sphere_intersect": [P, P, P, P, I, I, I, P, P, P, P],
    "ngp_ray_aabb_near": [P, P, P, P, F, I, P, P],
    "ngp_ray_aabb_near_noise": [P, P, P, P, F, I, C.c_uint64, P, P, P],
    "ngp_morton3D": [P, I, P, P],
    "ngp_morton3D_invert": [P, I, P, P],
    "ngp_packbits": [P, I, I, F, P, P],
    "ngp_density_grid_update": [P, P, P, F, I, P, P],
    "ngp_packbits_auto": [P, I, P, F, P, P],
    "ngp_raymarching_train_count": [P, P, P, P, I, F, F, P, I, I, I, P, P, P, P],
    "ngp_raymarching_train_write": [P, P, P, P, F, F, I, I, I, P, P, P, P, P],
    "ngp_raymarching_train_write_k": [P, P, P, P, F, F, I, I, I, P, P, P, P, I, P, P, P],
    "ngp_raymarching_train_write_kc": [P, P, P, P, F, F, I, I, I, P, P, P, P, I, P, P, P, P],
    "ngp_raymarching_train_count_k": [P, P, P, P, I, F, F, P, I, I, I, P, P, P, I, P, P],
    "ngp_stepper_record_bytes": [I],
    "ngp_debug_hashgrid_fwd_map": [C.POINTER(GridMeta), I, P, I],
    "ngp_raymarching_test": [P, P, P, P, P, I, F, F, I, I, I, I, P, P, P, P, P, P],
    "ngp_composite_train_fw": [P, P, P, P, P, F, I, I, P, P, P, P, P, P, P],
    "ngp_composite_train_bw": [P] * 13 + [F, I, I, P, P, P, P, P, P, P],
    "ngp_active_scan": [P, I, P, P],
    "ngp_composite_train_fw_loss": [P, P, P, P, P, F, I, I, P, P, P, P, P, P, P, P, P, F, F, P, P, P, P, P, C.c_size_t, P],
    "ngp_composite_probe": [P, P, P, I, F, I, P, P, P],
    "ngp_composite_train_fw_loss_counts": [P, P, P, P, P, F, I, I, P, P, P, P, P, P, P, P, F, F, P, P, P, C.c_size_t, P],
    "ngp_composite_train_fw_blend": [P, P, P, P, P, F, I, I, P, P, P, P, P, P, P, P, P],
    "ngp_composite_train_bw_render": [P] * 13 + [F, I, I, P, P, P, P, P, P, P, P, P],
    "ngp_composite_train_bw_tail": [P] * 13 + [F, I, I, P, P, P, P, P, P, P, P, P, P, P, C.c_size_t, P],
    "ngp_composite_train_fw_loss_h": [P, P, P, P, P, F, I, I, P, P, P, P, P, P, P, P, P, P, F, F, P, P, P, P, P, C.c_size_t, P],
    "ngp_composite_test_fw": [P, P, P, P, P, F, P, I, I, P, P, P, P],
    "ngp_distortion_loss_fw": [P, P, P, P, I, I, P, P, P, P],
    "ngp_distortion_loss_bw": [P, P, P, P, P, P, P, I, I, P, P],
    "ngp_grid_meta_init": [C.POINTER(GridMeta), I, I, I, I, F],
    "ngp_hashgrid_fwd": [P, P, P, P, C.POINTER(GridMeta), I, P, P],
    "ngp_hashgrid_fwd_list": [P, P, P, P, C.POINTER(GridMeta), I, P, I, P, P, P],
    "ngp_field_fwd_list": [P, P, P, P, I, P, I, P, P, P, P, P],
    "ngp_hashgrid_bwd": [P, P, P, P, C.POINTER(GridMeta), I, P, I, P],
    "ngp_hashgrid_bwd_sliced": [P, P, P, P, C.POINTER(GridMeta), I, P, P, P, P],
    "ngp_active_samples": [P, P, I, P, P, P, P],
    "ngp_density_fwd": [P, P, I, P, P, P],
    "ngp_field_fwd": [P, P, P, P, I, P, P, P, P],
    "ngp_density_bwd": [P, P, P, P, F, I, P, P, P, P, P],
    "ngp_field_bwd_partials": [I],
    "ngp_field_bwd": [P, P, P, P, P, P, P, F, I, P, P, P, P, P, P],
    "ngp_field_bwd_guarded": [P, P, P, P, P, P, P, F, P, F, I, P, P, P, P, P, P, I, P],
    "ngp_adam_use_loss_scaler": [P, I, F, F, I, F, F],
    "ngp_field_bwd_two_launches": [P, P, P, P, P, P, P, F, I, P, P, P, P, P, P],
    "ngp_field_bwd_uses_h": [],
    "ngp_mlp_fwd": [P, P, I, I, I, I, I, P, P],
    "ngp_mlp_bwd_partials": [I],
    "ngp_mlp_bwd": [P, P, P, I, I, I, I, I, P, P, P],
    "ngp_sh4_fwd": [P, I, P, P],
    "ngp_feats_to_rowmajor": [P, I, I, P, P],
    "ngp_feats_from_rowmajor": [P, I, I, P, P],
    "ngp_adam_step": [P, P, P, I, P, P, L, F, F, F, F, F, I, F, P, P],
    "ngp_adam_step_partials": [P, P, P, I, P, P, I, F, F, F, F, F, I, F, P, P],
    "ngp_adam_step_field": [P, P, P, P, P, L, P, P, P, P, P, I, P, P, P, P, P, I, I, F, F, F, F, F, I, F, I, P, P, P],
    "ngp_adam_step_field_shard": [P, P, P, P, P, L, P, P, P, P, P, I, P, P, P, P, P, I, I, F, F, F, F, F, I, F, P, P, P, P],
    "ngp_adam_step_field_merge": [P, P, P, P, P, L, P, P, P, P, P, I, P, P, P, P, P, I, I, F, F, F, F, F, I, F, P, P, C.POINTER(GridPartials), P],
    "ngp_hashgrid_bwd_binned_deferred": [P, P, P, P, C.POINTER(GridMeta), I, P, P, P, C.c_size_t, P, C.POINTER(GridPartials), P],
    "ngp_stepper_backward_update": [P, F, I, F, P, P],
    "ngp_adam_step_field_pieces": [P, P, P, P, P, L, L, I, I, I, P, P, P, P, P, I, P, P, P, P, P, I, I, F, F, F, F, F, I, F, P, P, P, P],
    "ngp_comm_unique_id": [P],
    "ngp_comm_create": [P, I, I, C.POINTER(P)],
    "ngp_comm_destroy": [P],
    "ngp_comm_info": [P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(P)],
    "ngp_comm_all_reduce": [P, P, L, I, P],
    "ngp_comm_reduce_scatter": [P, P, P, L, I, P],
    "ngp_comm_all_gather": [P, P, P, L, I, P],
    "ngp_comm_broadcast": [P, P, L, I, P],
    "ngp_comm_exchange_slices": [P, P, P, C.c_int64, I, P],
    "ngp_comm_all_gather_direct": [P, P, C.c_int64, I, P],
    "ngp_sum_slices_f16": [P, P, I, I, C.c_int64, P, P],
    "ngp_stepper_set_exchange": [P, P, C.POINTER(ExchangeConfig)],
    "ngp_stepper_tail": [P, F, I, F, P],
    "ngp_stepper_exchange_times": [P, C.POINTER(C.c_float), C.POINTER(C.c_float)],
    "ngp_reduce_partials": [P, I, I, P, P],
    "ngp_found_inf": [P, I, L, P, I, P],
    "ngp_found_inf2": [P, I, L, P, I, L, P, P, P],
    "ngp_reduce_partials2": [P, I, P, I, I, P, P],
    "ngp_cast_f32_to_f16": [P, L, P, P],
    "ngp_cast_f16_to_f32": [P, L, F, P, P],
    "ngp_nerf_loss": [P, P, P, P, F, F, I, P, P, P, P, P],
    "ngp_nerf_loss_terms_fw": [P, P, P, F, I, P, P, P],
    "ngp_nerf_loss_terms_bw": [P, I, P, I, P, P, P, F, I, P, P, P],
    "ngp_bg_blend": [P, P, P, I, P, P],
    "ngp_bg_blend_bw": [P, P, P, I, P, P],
    "ngp_compact_alive": [P, P, I, P, P, P, P],
    "ngp_sample_rays": [P, P, P, I, I, I, C.c_uint64, P, P, P, P, P, P, P],
    "ngp_get_rays": [P, P, I, P, P, P],
    "ngp_abi_version": [],
    "ngp_march_guard_read": [P, I],
    "ngp_march_guard_first": [P],
    "ngp_stepper_create": [C.POINTER(StepperConfig), C.POINTER(StepBuffersC), C.POINTER(P)],
    "ngp_stepper_destroy": [P],
    "ngp_stepper_set_buffers": [P, C.POINTER(StepBuffersC)],
    "ngp_stepper_set_sample_sets": [P, P, P, P, P],
    "ngp_stepper_march": [P, P, P, P, P],
    "ngp_stepper_pending": [P, P, P],
    "ngp_stepper_last_set": [P],
    "ngp_stepper_two_rounds": [P],
    "ngp_stepper_drop_pending": [P],
    "ngp_stepper_front": [P, P, P, P, P, P, F, F, P, P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    "ngp_stepper_table_backward": [P, I, I, P],
    "ngp_stepper_render_forward": [P, P, P, P, P, P, P, P, C.POINTER(C.c_int32)],
    "ngp_stepper_render_backward": [P, P, P, P, P, F, P, C.POINTER(C.c_int32)],
    "ngp_stepper_update": [P, F, I, F, P, P, I, P, P, P],
    "ngp_stepper_host_times": [P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong), I],
    "ngp_stepper_before_update": [P, C.POINTER(C.c_void_p)],
    "ngp_stepper_set_loss_scaler": [P, F, F, F, I, P],
    "ngp_stepper_loss_scale": [P, C.POINTER(C.c_float), C.POINTER(C.c_int32), P],
    "ngp_stepper_timing": [P, I],
    "ngp_stepper_stage_times": [P, C.POINTER(C.c_float)],
    "ngp_hashgrid_fwd_n": [P, P, P, P, C.POINTER(GridMeta), I, P, P, P],
    "ngp_field_fwd_n": [P, P, P, P, I, P, P, P, P, P],
    "ngp_hashgrid_bwd_binned": [P, P, P, P, C.POINTER(GridMeta), I, P, P, P, C.c_size_t, P, P],
    "ngp_hashgrid_bwd_binned_group": [P, P, P, P, C.POINTER(GridMeta), I, P, P, P, C.c_size_t, P, I, I, P],
    "ngp_hashgrid_bwd_binned_group_entries": [C.POINTER(GridMeta), I, I, I, C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    "ngp_hashgrid_bwd_input": [P, P, P, P, P, C.POINTER(GridMeta), I, F, P, P],
    "ngp_sh4_bwd": [P, P, I, F, P, P],
    "ngp_density_fwd_scatter": [P, P, I, P, P, P],
    "ngp_occupancy_update_workspace_layout": [I, I, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)],
    "ngp_march_train_fused": [P, P, P, P, F, C.c_uint64, P, I, F, F, I, I, I, P, P, P, P, P, P, P, P, P, P, P],
    "ngp_mark_invisible_cells": [P, P, I, I, I, F, I, I, F, P, P, P],
    "ngp_occupancy_update": [P, P, I, I, F, F, F, P, I, C.c_uint64, P, P, P, C.POINTER(GridMeta), P, P, C.c_size_t, P],
    "ngp_debug_render_block_hops": [I],
    "ngp_debug_render_wave_rays": [I],
    "ngp_render_test_frame": [P, P, P, P, I, F, F, I, I, F, P, P, P, C.POINTER(GridMeta), P, P, I, I, I,
                              C.POINTER(C.c_float), P, C.c_size_t, P, P, P, P, C.POINTER(C.c_int32), P],
}
_COUNT_QUERIES = ("ngp_field_bwd_partials", "ngp_field_bwd_uses_h", "ngp_mlp_bwd_partials", "ngp_abi_version", "ngp_stepper_pending", "ngp_stepper_last_set", "ngp_stepper_two_rounds",
                  "ngp_stepper_record_bytes", "ngp_debug_hashgrid_fwd_map")

_lib = None


def lib():
    """The loaded library; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libngp_hip.so is missing (%s): run `python -m ngp_pl_amd.build` or "
                               "__graft_entry__.build(); there is no CPU/eager fallback" % LIB_PATH)
        h = C.CDLL(LIB_PATH)
        for name, argtypes in _PROTOS.items():
            f = getattr(h, name)
            f.argtypes = argtypes
            f.restype = I
        h.ngp_build_arch.argtypes = []
        h.ngp_build_arch.restype = C.c_char_p
        h.ngp_comm_last_error.argtypes = []
        h.ngp_comm_last_error.restype = C.c_char_p
        h.ngp_render_test_workspace_bytes.argtypes = [I, I, F]
        h.ngp_render_test_workspace_bytes.restype = C.c_size_t
        h.ngp_hashgrid_bwd_binned_workspace_bytes.argtypes = [C.POINTER(GridMeta), I]
        h.ngp_hashgrid_bwd_binned_workspace_bytes.restype = C.c_size_t
        h.ngp_composite_train_fw_loss_workspace_bytes.argtypes = [I]
        h.ngp_composite_train_fw_loss_workspace_bytes.restype = C.c_size_t
        h.ngp_occupancy_update_workspace_bytes.argtypes = [I, I]
        h.ngp_occupancy_update_workspace_bytes.restype = C.c_size_t
        if h.ngp_abi_version() != ABI_VERSION:                         # a stale .so next to newer Python (or the reverse)
            raise RuntimeError("%s has ABI version %d, this package binds version %d: rebuild the library (python -m ngp_pl_amd.build)" % (
                LIB_PATH, h.ngp_abi_version(), ABI_VERSION))
        for which, rec in ((0, StepperConfig), (1, StepBuffersC), (2, ExchangeConfig)):      # the mirrored records must be the library's own layout
            if h.ngp_stepper_record_bytes(which) != C.sizeof(rec):
                raise RuntimeError("%s is %d bytes here, %d in %s: rebuild the library (python -m ngp_pl_amd.build)" % (
                    rec.__name__, C.sizeof(rec), h.ngp_stepper_record_bytes(which), LIB_PATH))
        _lib = h
    return _lib


def exported_symbols():
    return list(_PROTOS) + ["ngp_build_arch", "ngp_comm_last_error", "ngp_render_test_workspace_bytes", "ngp_occupancy_update_workspace_bytes",
                                  "ngp_hashgrid_bwd_binned_workspace_bytes", "ngp_composite_train_fw_loss_workspace_bytes"]


class NgpError(RuntimeError):
    pass


def call(name, *args):
    """Invoke a C-ABI entry point; non-zero status raises."""
    rc = getattr(lib(), name)(*args)
    if name in _COUNT_QUERIES:
        return rc
    if rc != 0:
        kind = {-1: "NGP_EINVAL (bad argument)", -2: "NGP_EUNSUP (unsupported configuration)",
                -3: "NGP_ETIMEOUT (a device result did not arrive within NGP_SPIN_TIMEOUT_S)"}.get(rc, "hipError_t %d" % rc)
        if rc == -4:
            kind = "NGP_ECOMM (%s)" % (lib().ngp_comm_last_error() or b"?").decode(errors="replace")
        raise NgpError("%s failed: %s" % (name, kind))
    return 0


_FIELD_BWD_USES_H = None


def field_bwd_uses_h():
    """False when ngp_field_bwd neither reads the forward's h_out nor writes dh_scratch (the one-launch kernel: both may be NULL,
    the forward then skips the (S,16) store); True in the two-launch A/B build."""
    global _FIELD_BWD_USES_H
    if _FIELD_BWD_USES_H is None:
        _FIELD_BWD_USES_H = bool(call("ngp_field_bwd_uses_h"))
    return _FIELD_BWD_USES_H


SPIN_TIMEOUT_S = float(os.environ.get("NGP_SPIN_TIMEOUT_S", "30"))


class DeviceTimeout(NgpError):
    """A host poll on a device result ran out of time: the kernel named in the message did not finish."""


def poll_event(event, what, word=None, timeout_s=None):
    """Busy-polls `event` (a torch.cuda.Event) -- and, first, the pinned count word `word` (a numpy view that holds -1 until the
    kernel writes it) -- like the loops it replaces (no interrupt-driven sleep: those wake tens of microseconds late), but
    with a deadline: after `timeout_s` (NGP_SPIN_TIMEOUT_S, default 30 s) it raises DeviceTimeout naming the kernel instead
    of spinning for ever on a march that does not come back."""
    import time
    spins, t_end = 0, None
    limit = SPIN_TIMEOUT_S if timeout_s is None else timeout_s
    if word is not None:
        while word[0] < 0:
            spins += 1
            if spins & 1023 == 0:
                if event.query():
                    break
                if spins > 50000:
                    time.sleep(0)                    # a long wait: offer the core to other threads
                now = time.perf_counter()
                if t_end is None:
                    t_end = now + limit
                elif now > t_end:
                    raise DeviceTimeout("%s: no result after %.0f s (the kernel is still running or the device is gone)" % (what, limit))
    while not event.query():
        spins += 1
        if spins & 1023 == 0:
            now = time.perf_counter()
            if t_end is None:
                t_end = now + limit
            elif now > t_end:
                raise DeviceTimeout("%s: event not reached after %.0f s (the kernel is still running or the device is gone)" % (what, limit))


def march_guard_counts(reset=False):
    """Counts of tripped termination guards in the marching kernels since the last reset (synchronises): all zero for rays
    that come from an AABB / sphere intersection."""
    buf = (C.c_uint32 * 4)()
    call("ngp_march_guard_read", C.cast(buf, P), 1 if reset else 0)
    return list(buf)


def march_guard_first():
    """The first probe that tripped the absorbed-step guard: dict of t, t_target, face distances, ray origin / direction."""
    buf = (C.c_float * 12)()
    call("ngp_march_guard_first", C.cast(buf, P))
    v = list(buf)
    return dict(t=v[0], t_target=v[1], t_faces=v[2:5], origin=v[5:8], direction=v[8:11], dt_lo=v[11])


_NO_GUARD = contextlib.nullcontext()


def device_guard(dev):
    """`with torch.cuda.device(dev)` only when dev is not the current device already (one process per GPU sets its device once; the
    guard's bookkeeping costs 10-20 us of host time per use, and the reference-shaped step path used six of them)."""
    idx = dev.index
    if idx is None or torch.cuda.current_device() == idx:
        return _NO_GUARD
    return torch.cuda.device(dev)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


try:
    _raw_stream, _cur_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice
except AttributeError:              # another torch build: the public API (12 us per call here: device-index and availability checks in Python)
    _raw_stream = _cur_device = None


def stream():
    """Raw handle of torch's current stream on the current device.  `torch.cuda.current_stream().cuda_stream` measured 12 us per call
    (tools/api_cprofile.py: four calls per step on the reference-shaped path); the C entry point behind it takes 0.3 us."""
    if _raw_stream is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    """Mirrors CHECK_INPUT of the reference (include/utils.h:4-6): device + contiguity."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("tensor must be a CUDA (HIP) tensor")
        if not t.is_contiguous():
            raise RuntimeError("tensor must be contiguous")
