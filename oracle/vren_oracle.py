"""TEST INFRASTRUCTURE -- numpy front end of the CPU oracle and of the CPU-compiled reference.

`Oracle()`       wraps oracle/libngp_oracle.so  (our plain-C restatement, oracle/ngp_oracle.c).
`Reference(fma)` wraps oracle/_ref/libvren_ref_{fma,nofma}.so (the reference's own .cu sources
                 compiled for the host by oracle/build_ref.sh).
Both expose the 12 `vren` entry points (/root/reference/models/csrc/binding.cpp:234-250) with
numpy arrays in / numpy arrays out, same names and argument order as the pybind module.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
C = ctypes
_f = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32 = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64 = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_u8 = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _fp(a):
    """float32 array or NULL."""
    return None if a is None else _c(a, np.float32).ctypes.data_as(C.POINTER(C.c_float))


class _Vren:
    """Shared numpy API; subclasses bind `self._fn(name)`."""

    prefix = None
    lib = None

    def _fn(self, name, argtypes, restype=None):
        f = getattr(self.lib, self.prefix + name)
        f.argtypes = argtypes
        f.restype = restype
        return f

    # -- helpers (raymarching.cu:62-161) --------------------------------------------------
    def morton3D(self, coords):
        coords = _c(coords, np.int32)
        n = coords.shape[0]
        out = np.zeros(n, np.int32)
        self._fn("morton3D", [_i32, C.c_int, _i32])(coords, n, out)
        return out

    def morton3D_invert(self, indices):
        indices = _c(indices, np.int32)
        n = indices.shape[0]
        out = np.zeros((n, 3), np.int32)
        self._fn("morton3D_invert", [_i32, C.c_int, _i32])(indices, n, out)
        return out

    def packbits(self, density_grid, density_threshold, density_bitfield):
        grid = _c(density_grid, np.float32).reshape(-1)
        assert density_bitfield.dtype == np.uint8 and density_bitfield.flags.c_contiguous
        self._fn("packbits", [_f, C.c_int, C.c_float, _u8])(grid, density_bitfield.size, density_threshold, density_bitfield)

    # -- intersection (intersection.cu) ---------------------------------------------------
    def _intersect(self, name, rays_o, rays_d, centers, extents, max_hits):
        rays_o, rays_d = _c(rays_o, np.float32), _c(rays_d, np.float32)
        centers, extents = _c(centers, np.float32), _c(extents, np.float32)
        n, v = rays_o.shape[0], centers.shape[0]
        cnt = np.zeros(n, np.int32)
        hits_t = np.zeros((n, max_hits, 2), np.float32)
        hits_idx = np.zeros((n, max_hits), np.int64)
        self._fn(name, [_f, _f, _f, _f, C.c_int, C.c_int, C.c_int, _i32, _f, _i64])(
            rays_o, rays_d, centers, extents, n, v, max_hits, cnt, hits_t, hits_idx)
        return cnt, hits_t, hits_idx

    def ray_aabb_intersect(self, rays_o, rays_d, centers, half_sizes, max_hits):
        return self._intersect("ray_aabb_intersect", rays_o, rays_d, centers, half_sizes, max_hits)

    def ray_sphere_intersect(self, rays_o, rays_d, centers, radii, max_hits):
        return self._intersect("ray_sphere_intersect", rays_o, rays_d, centers, radii, max_hits)

    # -- marching (raymarching.cu:163-454) -------------------------------------------------
    def raymarching_train(self, rays_o, rays_d, hits_t, density_bitfield, cascades, scale, exp_step_factor,
                          noise, grid_size, max_samples):
        """Returns rays_a, xyzs, dirs, deltas, ts, counter with the sample arrays already cut to
        counter[0] rows (what RayMarcher.forward does, custom_functions.py:91-96)."""
        rays_o, rays_d, hits_t = _c(rays_o, np.float32), _c(rays_d, np.float32), _c(hits_t, np.float32)
        noise, bitfield = _c(noise, np.float32), _c(density_bitfield, np.uint8)
        n = rays_o.shape[0]
        cap = n * max_samples
        rays_a = np.zeros((n, 3), np.int64)
        xyzs = np.zeros((cap, 3), np.float32); dirs = np.zeros((cap, 3), np.float32)
        deltas = np.zeros(cap, np.float32); ts = np.zeros(cap, np.float32)
        counter = np.zeros(2, np.int32)
        S = self._train(rays_o, rays_d, hits_t, bitfield, cascades, scale, exp_step_factor, noise, grid_size,
                        max_samples, n, cap, rays_a, xyzs, dirs, deltas, ts, counter)
        assert S >= 0
        return rays_a, xyzs[:S].copy(), dirs[:S].copy(), deltas[:S].copy(), ts[:S].copy(), counter

    def raymarching_test(self, rays_o, rays_d, hits_t, alive_indices, density_bitfield, cascades, scale,
                         exp_step_factor, grid_size, max_samples, N_samples):
        """hits_t (R,2) float32 C-contiguous is advanced IN PLACE."""
        rays_o, rays_d = _c(rays_o, np.float32), _c(rays_d, np.float32)
        assert hits_t.dtype == np.float32 and hits_t.flags.c_contiguous
        alive, bitfield = _c(alive_indices, np.int64), _c(density_bitfield, np.uint8)
        na = alive.shape[0]
        xyzs = np.zeros((na, N_samples, 3), np.float32); dirs = np.zeros((na, N_samples, 3), np.float32)
        deltas = np.zeros((na, N_samples), np.float32); ts = np.zeros((na, N_samples), np.float32)
        n_eff = np.zeros(na, np.int32)
        self._test(rays_o, rays_d, hits_t, alive, bitfield, cascades, scale, exp_step_factor, grid_size, max_samples,
                   N_samples, na, xyzs, dirs, deltas, ts, n_eff)
        return xyzs, dirs, deltas, ts, n_eff

    # -- compositing (volumerendering.cu) --------------------------------------------------
    def composite_train_fw(self, sigmas, rgbs, deltas, ts, rays_a, T_threshold):
        sigmas, rgbs, deltas, ts = (_c(a, np.float32) for a in (sigmas, rgbs, deltas, ts))
        rays_a = _c(rays_a, np.int64)
        R, S = rays_a.shape[0], sigmas.shape[0]
        total = np.zeros(R, np.int64); opacity = np.zeros(R, np.float32); depth = np.zeros(R, np.float32)
        rgb = np.zeros((R, 3), np.float32); ws = np.zeros(S, np.float32)
        self._fn("composite_train_fw", [_f, _f, _f, _f, _i64, C.c_float, C.c_int, C.c_int, _i64, _f, _f, _f, _f])(
            sigmas, rgbs, deltas, ts, rays_a, T_threshold, R, S, total, opacity, depth, rgb, ws)
        return total, opacity, depth, rgb, ws

    def composite_train_bw(self, dL_dopacity, dL_ddepth, dL_drgb, dL_dws, sigmas, rgbs, ws, deltas, ts, rays_a,
                           opacity, depth, rgb, T_threshold):
        arrs = [_c(a, np.float32) for a in (dL_dopacity, dL_ddepth, dL_drgb, dL_dws, sigmas, rgbs, ws, deltas, ts)]
        rays_a = _c(rays_a, np.int64)
        opacity, depth, rgb = (_c(a, np.float32) for a in (opacity, depth, rgb))
        R, S = rays_a.shape[0], arrs[4].shape[0]
        dsig = np.zeros(S, np.float32); drgbs = np.zeros((S, 3), np.float32)
        self._fn("composite_train_bw", [_f] * 9 + [_i64, _f, _f, _f, C.c_float, C.c_int, C.c_int, _f, _f])(
            *arrs, rays_a, opacity, depth, rgb, T_threshold, R, S, dsig, drgbs)
        return dsig, drgbs

    def composite_test_fw(self, sigmas, rgbs, deltas, ts, hits_t, alive_indices, T_threshold, N_eff_samples,
                          opacity, depth, rgb):
        """alive_indices, opacity, depth, rgb are updated IN PLACE (must be C-contiguous arrays)."""
        sigmas, rgbs, deltas, ts = (_c(a, np.float32) for a in (sigmas, rgbs, deltas, ts))
        n_eff = _c(N_eff_samples, np.int32)
        na, ns = sigmas.shape
        self._ctest(sigmas, rgbs, deltas, ts, _c(hits_t, np.float32), alive_indices, T_threshold, n_eff, na, ns,
                    opacity, depth, rgb)

    # -- distortion loss (losses.cu) -------------------------------------------------------
    def distortion_loss_fw(self, ws, deltas, ts, rays_a):
        ws, deltas, ts = (_c(a, np.float32) for a in (ws, deltas, ts))
        rays_a = _c(rays_a, np.int64)
        R, S = rays_a.shape[0], ws.shape[0]
        loss = np.zeros(R, np.float32); a = np.zeros(S, np.float32); b = np.zeros(S, np.float32)
        self._fn("distortion_loss_fw", [_f, _f, _f, _i64, C.c_int, C.c_int, _f, _f, _f])(ws, deltas, ts, rays_a, R, S, loss, a, b)
        return loss, a, b

    def distortion_loss_bw(self, dL_dloss, ws_inclusive_scan, wts_inclusive_scan, ws, deltas, ts, rays_a):
        arrs = [_c(a, np.float32) for a in (dL_dloss, ws_inclusive_scan, wts_inclusive_scan, ws, deltas, ts)]
        rays_a = _c(rays_a, np.int64)
        R, S = rays_a.shape[0], arrs[3].shape[0]
        out = np.zeros(S, np.float32)
        self._fn("distortion_loss_bw", [_f] * 6 + [_i64, C.c_int, C.c_int, _f])(*arrs, rays_a, R, S, out)
        return out


class Oracle(_Vren):
    """Our plain-C restatement (oracle/ngp_oracle.c).  fma=True mimics nvcc's contraction."""
    prefix = "oracle_"

    def __init__(self, fma=True):
        path = os.path.join(HERE, "libngp_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle not built: run `make -C oracle` (or __graft_entry__.build())")
        self.lib = C.CDLL(path)
        self.fma = bool(fma)

    def _set(self):
        self.lib.oracle_set_fma(int(self.fma))

    def _fn(self, name, argtypes, restype=None):
        self._set()
        return super()._fn(name, argtypes, restype)

    def _train(self, ro, rd, ht, bf, cascades, scale, esf, noise, G, ms, n, cap, rays_a, xyzs, dirs, deltas, ts, counter):
        f = self._fn("raymarching_train", [_f, _f, _f, _u8, C.c_int, C.c_float, C.c_float, _f, C.c_int, C.c_int, C.c_int,
                                           C.c_longlong, _i64, _f, _f, _f, _f, _i32], C.c_longlong)
        return f(ro, rd, ht, bf, cascades, scale, esf, noise, G, ms, n, cap, rays_a, xyzs, dirs, deltas, ts, counter)

    def _test(self, ro, rd, ht, alive, bf, cascades, scale, esf, G, ms, ns, na, xyzs, dirs, deltas, ts, n_eff):
        f = self._fn("raymarching_test", [_f, _f, _f, _i64, _u8, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int,
                                          C.c_int, _f, _f, _f, _f, _i32])
        f(ro, rd, ht, alive, bf, cascades, scale, esf, G, ms, ns, na, xyzs, dirs, deltas, ts, n_eff)

    def _ctest(self, sig, rgbs, deltas, ts, hits_t, alive, T, n_eff, na, ns, opacity, depth, rgb):
        f = self._fn("composite_test_fw", [_f, _f, _f, _f, _i64, C.c_float, _i32, C.c_int, C.c_int, _f, _f, _f])
        f(sig, rgbs, deltas, ts, alive, T, n_eff, na, ns, opacity, depth, rgb)


class Reference(_Vren):
    """The reference's own kernels compiled for the CPU (oracle/build_ref.sh)."""
    prefix = "ref_"

    def __init__(self, fma=True):
        path = os.path.join(HERE, "_ref", "libvren_ref_%s.so" % ("fma" if fma else "nofma"))
        if not os.path.exists(path):
            raise RuntimeError("reference build missing: run oracle/build_ref.sh where /root/reference exists")
        import torch  # noqa: F401  (libtorch must be loaded first; the .so links against it)
        self.lib = C.CDLL(path)
        self.fma = bool(fma)

    @staticmethod
    def available(fma=True):
        return os.path.exists(os.path.join(HERE, "_ref", "libvren_ref_%s.so" % ("fma" if fma else "nofma")))

    def _train(self, ro, rd, ht, bf, cascades, scale, esf, noise, G, ms, n, cap, rays_a, xyzs, dirs, deltas, ts, counter):
        f = self._fn("raymarching_train", [_f, _f, _f, _u8, C.c_int, C.c_float, C.c_float, _f, C.c_int, C.c_int, C.c_int,
                                           C.c_longlong, C.c_longlong, _i64, _f, _f, _f, _f, _i32], C.c_longlong)
        return f(ro, rd, ht, bf, cascades, scale, esf, noise, G, ms, n, cap, bf.size, rays_a, xyzs, dirs, deltas, ts, counter)

    def _test(self, ro, rd, ht, alive, bf, cascades, scale, esf, G, ms, ns, na, xyzs, dirs, deltas, ts, n_eff):
        f = self._fn("raymarching_test", [_f, _f, _f, _i64, _u8, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_longlong, _f, _f, _f, _f, _i32])
        f(ro, rd, ht, alive, bf, cascades, scale, esf, G, ms, ns, na, ro.shape[0], bf.size, xyzs, dirs, deltas, ts, n_eff)

    def _ctest(self, sig, rgbs, deltas, ts, hits_t, alive, T, n_eff, na, ns, opacity, depth, rgb):
        f = self._fn("composite_test_fw", [_f, _f, _f, _f, _f, _i64, C.c_float, _i32, C.c_int, C.c_int, C.c_int, _f, _f, _f])
        f(sig, rgbs, deltas, ts, hits_t, alive, T, n_eff, na, ns, opacity.shape[0], opacity, depth, rgb)
