"""Buffers and handles of the native step (csrc/stepper.hip): `StepBuffers` = every buffer a step touches as ONE arena per batch
size, handed to the library as an `ngp_step_buffers` record; `RenderStepper` = the per-model stepper behind `render()`'s training
branch (rendering.py:121-163), which runs the same launch sequence as `Trainer.step` split where the reference's API splits it:
forward half inside `render()`, backward half inside the autograd node's backward."""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import call, ptr, stream

MAX_SAMPLES = 1024          # rendering.py:7 (ngp_pl_amd.rendering re-exports it)


def _align(x, a=256):
    return (x + a - 1) // a * a


class StepBuffers:
    """Every buffer a step touches, allocated ONCE per batch size: the sample-proportional ones at the worst
    case S = R * MAX_SAMPLES (308 B per sample slot: 2.6 GB for 8192 rays -- under 1 % of the 288 GB of HBM; only
    the first S rows of each are ever touched), two sets of march records (the march of batch k+1 runs while
    step k still reads its own), pinned count words, the table-backward workspace.  After construction a step
    performs no hipMalloc / hipHostMalloc / hipFuncSetAttribute: the first timed step costs what the 10 000th does."""

    PER_SAMPLE = (("xyzs", 12), ("dirs", 12), ("deltas", 4), ("ts", 4), ("feats", 64), ("h", 32), ("sigmas", 4), ("rgbs", 12),
                  ("ws", 4), ("dL_dsigmas", 4), ("dL_drgbs", 12), ("active", 4), ("x_act", 12), ("dh", 32), ("dfeats", 64),
                  # the packed samples a second time: a prefetched march expands into the set the running step does not read
                  # (ngp_stepper_set_sample_sets)
                  ("xyzs1", 12), ("dirs1", 12), ("deltas1", 4), ("ts1", 4))
    DISTORTION = (("ws_incl", 4), ("wts_incl", 4), ("dL_dws", 4))
    PER_RAY = (("total", 8), ("opacity", 4), ("depth", 4), ("rgb", 12), ("dL_drgb", 12), ("dL_dopacity", 4), ("ray_offs", 4),
               ("dist", 4), ("zeros", 4), ("dist_seed", 4), ("rgb_out", 12))
    MARCH = (("hits_t", 8), ("rays_a", 24), ("noise", 4), ("scratch", 4 * MAX_SAMPLES), ("offs_k", 4))
    MAX_PARTIALS = 256

    def __init__(self, model, n_rays, distortion, binned):
        enc, net = model.xyz_encoder, model.rgb_net
        dev = model.center.device
        self.n, self.cap = n_rays, n_rays * MAX_SAMPLES
        lib = _lib.lib()
        self.n_mlp_params = enc.n_mlp + net.params.numel()
        off, total = {}, 0

        def add(name, nbytes):
            nonlocal total
            off[name] = total
            total += _align(nbytes)
        for name, b in self.PER_SAMPLE + (self.DISTORTION if distortion else ()):
            add(name, b * self.cap)
        for name, b in self.PER_RAY:
            add(name, b * n_rays)
        for k in (0, 1):
            for name, b in self.MARCH:
                add("%s%d" % (name, k), b * n_rays)
        add("n_active", 4); add("stats", 8)
        add("list_k", 4 * 64 * n_rays); add("list_rest", 4 * self.cap); add("two_round_counts", 16)
        add("partials", self.MAX_PARTIALS * self.n_mlp_params * 4)
        self.fw_bytes = int(lib.ngp_composite_train_fw_loss_workspace_bytes(n_rays))
        add("fw_ws", self.fw_bytes)
        # the binned table backward takes batches up to its chunk directory (1 M samples); larger ones (only the first
        # steps of the occupancy warm-up) go to the one-pass sliced kernel, which needs no workspace
        self.bin_max, self.bin_bytes = 0, 0
        if binned:
            # workspace for up to 288 samples per ray (the first steps of a run, when every cell still counts as occupied, march ~245),
            # at most the kernels' 4608 chunks of 1024: larger batches fall to the one-pass kernel
            s = min(self.cap, -(-288 * n_rays // 1024) * 1024, 4608 * 1024)
            while s > 0 and not lib.ngp_hashgrid_bwd_binned_workspace_bytes(C.byref(enc.meta), s):
                s -= 1024
            self.bin_max = max(s, 0)
            self.bin_bytes = int(lib.ngp_hashgrid_bwd_binned_workspace_bytes(C.byref(enc.meta), self.bin_max)) if self.bin_max else 0
            add("bin_ws", self.bin_bytes)
        self.arena = torch.empty(total, dtype=torch.uint8, device=dev)
        base = self.arena.data_ptr()
        assert base % 256 == 0
        self.p = {k: base + v for k, v in off.items()}
        self.off = off

        def view(name, dtype, *shape):
            n = 1
            for d in shape:
                n *= d
            return self.arena[off[name]:off[name] + n * torch.empty(0, dtype=dtype).element_size()].view(dtype).view(*shape)
        self.view = view
        self._flat = {}              # name -> the whole buffer as a flat tensor (cached: a prefix of it is one slice, ~3 us)
        self._cached = {}            # (name, dtype, shape) -> fixed-size view
        f32 = torch.float32
        self.total = view("total", torch.int64, n_rays); self.opacity = view("opacity", f32, n_rays)
        self.depth = view("depth", f32, n_rays); self.rgb = view("rgb", f32, n_rays, 3)
        self.stats = view("stats", f32, 2); self.dist = view("dist", f32, n_rays)
        self.n_active = view("n_active", torch.int32, 1)
        view("zeros", f32, n_rays).zero_()               # dL/ddepth: the loss has no depth term
        # The two-round forward leaves the samples behind a ray's stop unevaluated; the composite masks them by position (composite.hip:
        # chunk_transmittance), so their contents cannot matter -- zero-filled once all the same, so that a debugger or a new kernel
        # that does read them sees sigma = 0, rgb = 0 or an earlier step's values, never allocator bytes.
        view("sigmas", f32, self.cap).zero_(); view("rgbs", f32, self.cap, 3).zero_()
        self.dist_seed_val = None
        self.noise = [view("noise%d" % k, f32, n_rays) for k in (0, 1)]
        # {S, R} of a march is written by its scan kernel straight into pinned (device-mapped) host memory
        view("two_round_counts", torch.int32, 4).zero_()
        # {S, R, live samples of the step that consumed the march, -} per record set
        self.counter_host = [torch.zeros(4, dtype=torch.int32).pin_memory() for _ in (0, 1)]
        self.counter_np = [t.numpy() for t in self.counter_host]
        self.counter_p = [t.data_ptr() for t in self.counter_host]
        self.next_set = 0
        self.ready = [torch.cuda.Event() for _ in (0, 1)]
        self.done = [torch.cuda.Event() for _ in (0, 1)]

    def prefix(self, name, dtype, n):
        """The first n elements of per-sample buffer `name` (what a step with n samples wrote)."""
        flat = self._flat.get((name, dtype))
        if flat is None:
            nbytes = dict(self.PER_SAMPLE + self.DISTORTION)[name] * self.cap
            flat = self._flat[(name, dtype)] = self.arena[self.off[name]:self.off[name] + nbytes].view(dtype)
        return flat[:n]

    def fixed(self, name, dtype, *shape):
        """view() for shapes that do not change from step to step: built once."""
        key = (name, dtype, shape)
        v = self._cached.get(key)
        if v is None:
            v = self._cached[key] = self.view(name, dtype, *shape)
        return v

    def partial_rows(self, n_part, n_density):
        """(density rows, rgb rows) views of the MLP weight-gradient partials the field backward wrote (n_part rows each)."""
        key = ("partials", n_part)
        v = self._cached.get(key)
        if v is None:
            o = self.off["partials"]
            v = self._cached[key] = (self.view("partials", torch.float32, n_part * n_density),
                                     self.arena[o + 4 * n_part * n_density:o + 4 * n_part * self.n_mlp_params].view(torch.float32))
        return v

    def c_struct(self, distortion):
        """The same buffers as an ngp_step_buffers record for the native stepper."""
        P = self.p
        c = _lib.StepBuffersC()
        c.n_rays, c.distortion, c.cap = self.n, 1 if distortion else 0, self.cap
        for name in ("xyzs", "dirs", "deltas", "ts", "feats", "h", "sigmas", "rgbs", "ws", "dL_dsigmas", "dL_drgbs", "active", "x_act", "dh",
                     "dfeats", "total", "opacity", "depth", "rgb", "dL_drgb", "dL_dopacity", "ray_offs", "dist", "zeros", "dist_seed",
                     "n_active", "stats", "partials", "fw_ws", "list_k", "list_rest", "two_round_counts"):
            setattr(c, name, P[name])
        if distortion:
            c.ws_incl, c.wts_incl, c.dL_dws = P["ws_incl"], P["wts_incl"], P["dL_dws"]
        for k in (0, 1):
            c.hits_t[k], c.rays_a[k], c.noise[k], c.scratch[k] = P["hits_t%d" % k], P["rays_a%d" % k], P["noise%d" % k], P["scratch%d" % k]
            c.counter[k] = self.counter_p[k]
            c.offs_k[k] = P["offs_k%d" % k]
        c.max_partials, c.fw_bytes = self.MAX_PARTIALS, self.fw_bytes
        c.bin_ws, c.bin_bytes, c.bin_max = (P["bin_ws"] if self.bin_max else None), self.bin_bytes, self.bin_max
        return c

    def attach_sample_sets(self, handle):
        """Hands the second set of packed-sample buffers to a stepper (NGP_TWO_SAMPLE_SETS=0: one set, expansion on the main stream).
        Which set a step read: `sample_set(handle)`."""
        self.two_sets = os.environ.get("NGP_TWO_SAMPLE_SETS", "1") != "0"
        if self.two_sets:
            P = self.p
            call("ngp_stepper_set_sample_sets", handle, P["xyzs1"], P["dirs1"], P["deltas1"], P["ts1"])
        return self.two_sets

    def sample_name(self, name, k):
        """Buffer name of packed-sample array `name` (xyzs / dirs / deltas / ts) in sample set k (ngp_stepper_last_set)."""
        return name + "1" if (k == 1 and getattr(self, "two_sets", False)) else name

    def sample_views(self, S, k=0):
        """Tensor views of the last step's packed samples (debugging / tests; the step itself uses raw pointers).  k: the sample
        set the step read (ngp_stepper_last_set() of a stepper with two sets, else 0)."""
        f32 = torch.float32
        nm = lambda a: self.sample_name(a, k)       # noqa: E731
        return dict(xyzs=self.view(nm("xyzs"), f32, S, 3), dirs=self.view(nm("dirs"), f32, S, 3), deltas=self.view(nm("deltas"), f32, S),
                    ts=self.view(nm("ts"), f32, S), sigmas=self.view("sigmas", f32, S), rgbs=self.view("rgbs", f32, S, 3),
                    ws=self.view("ws", f32, S))




class RenderStepper:
    """The native stepper behind `render()`'s training branch for ONE model: buffers per batch size, the library handle, the
    generation counter that ties a backward to its forward (the per-sample results live in the step buffers: they are valid until
    the model's next training-branch render)."""

    def __init__(self, model):
        self.model = model
        self.buf = None
        self.handle = None
        self.key = None
        self.generation = 0
        self.side = None
        self.pending = None          # (ptr_o, ptr_d, tensors) of a prefetched march
        self.scaler_on, self.scale_saved = False, None
        dev = model.center.device
        self.bg_buf = torch.zeros(3, device=dev)
        self.bg_src = None

    def _destroy(self):
        if self.handle is not None:
            if getattr(self, "scaler_on", False):
                self.scale_saved = self.loss_scale_state()[0]          # carried over the rebuild
            _lib.lib().ngp_stepper_destroy(self.handle)
            self.handle = self.key = self.pending = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:            # noqa: BLE001 -- interpreter shutdown
            pass

    def prepare(self, n_rays, esf, T_threshold, bg):
        """Buffers for n_rays and a library handle that matches the model's current pointers and this call's recipe."""
        from . import tcnn
        m = self.model
        enc, net = m.xyz_encoder, m.rgb_net
        if self.buf is None or self.buf.n != n_rays:
            self._destroy()
            torch.cuda.synchronize()
            self.buf = None
            self.buf = StepBuffers(m, n_rays, False, tcnn.binned_enabled())
        if bg is not self.bg_src:                    # the two constant colours are cached objects: a copy only when it changes
            self.bg_buf.copy_(bg); self.bg_src = bg
        eh, rh = enc._half.get(enc.params), net._half.get(net.params)
        g16 = m._grid_grad16(enc.params.device)
        key = (eh.data_ptr(), rh.data_ptr(), g16.data_ptr(), m.density_bitfield.data_ptr(), m.center.data_ptr(), float(esf), float(T_threshold))
        if self.handle is None or key != self.key:
            self._destroy()
            c = _lib.StepperConfig()
            c.center, c.half_size, c.xyz_min, c.xyz_max = ptr(m.center), ptr(m.half_size), ptr(m.xyz_min), ptr(m.xyz_max)
            c.density_bitfield = ptr(m.density_bitfield)
            c.cascades, c.grid_size, c.scale, c.exp_step_factor = m.cascades, m.grid_size, float(m.scale), float(esf)
            c.meta = enc.meta
            c.enc_half, c.rgb_half = ptr(eh), ptr(rh)                              # no f32 masters / moments: this stepper never updates
            c.n_grid, c.n_density, c.n_rgb, c.grid_grad16 = enc.n_grid, enc.n_mlp, net.params.numel(), ptr(g16)
            from .rendering import NEAR_DISTANCE
            c.max_samples, c.near_distance, c.T_threshold = MAX_SAMPLES, NEAR_DISTANCE, float(T_threshold)
            c.lambda_opacity, c.lambda_distortion, c.bg = 0.0, 0.0, ptr(self.bg_buf)
            c.beta1, c.beta2, c.eps, c.weight_decay = 0.9, 0.999, 1e-15, 0.0
            c.noise_seed = int(os.environ.get("NGP_NOISE_SEED", "20240924")) ^ 0x5DEECE66D
            bc = self.buf.c_struct(False)
            h = C.c_void_p()
            call("ngp_stepper_create", C.byref(c), C.byref(bc), C.byref(h))
            self.handle, self.key = h, key
            self.buf.attach_sample_sets(h)
            # the dynamic loss scale (GradScaler's rule on the device, include/ngp_hip.h): this node only runs with native gradients,
            # i.e. with this package's FusedAdam behind it, which hands the scale to its launch (optim.FusedAdam._step_native)
            cfg = getattr(m, "native_loss_scaler", None)
            if cfg:
                call("ngp_stepper_set_loss_scaler", h, float(self.scale_saved or cfg["init_scale"]), float(cfg["growth_factor"]), float(cfg["backoff_factor"]),
                     int(cfg["growth_interval"]), stream())
            self.scaler_on = bool(cfg)
        return self.buf, self.handle

    def loss_scale_state(self):
        """(scale, clean steps) of the node's dynamic loss scale; (1.0, 0) when off.  Syncs."""
        if self.handle is None or not self.scaler_on:
            return 1.0, 0
        sc, tr = C.c_float(0.0), C.c_int32(0)
        call("ngp_stepper_loss_scale", self.handle, C.byref(sc), C.byref(tr), stream())
        return float(sc.value), int(tr.value)
