"""Lightning-free driver of one training step, mirroring NeRFSystem.training_step
(/root/reference/train.py:159-185) + configure_optimizers (:112-139) for the default recipe
(Synthetic-NeRF: scale 0.5, white background, no distortion loss, no pose optimisation).

  * `step()`          -- the hot path: the native stepper (csrc/stepper.hip), three library calls per step, no autograd
                         graph, native gradient buffers consumed by the fused Adam.  The ray march of step k+1 only needs
                         the occupancy bitfield, so it runs on a SECOND HIP stream concurrently with step k's
                         encode / MLP / backward kernels and its count lands in pinned host memory before step k+1 starts.
  * `step_autograd()` -- the same maths through render() + NeRFLoss + torch autograd, i.e. what the reference's
                         train.py drives.
  * `_exchange_and_update()` -- the host-side restatement of the step's tail for the torch.distributed exchanges
                         (ddp.GradientExchange / ShardedExchange); device agnostic, driven with CPU tensors by the gloo tests.
"""
import contextlib
import ctypes as C
import math
import os

import torch

from . import _lib, tcnn
from ._lib import call, ptr, stream
from .losses import NeRFLoss
from .optim import FusedAdam, cosine_lr
from .rendering import MAX_SAMPLES, NEAR_DISTANCE, render


from .stepper import StepBuffers, _align        # noqa: E402,F401  (the step's buffers: shared with rendering.py's native render node)


_MARCH_STREAMS = {}


def marching_stream(dev):
    """THE marching stream of a device: one per process, shared by every Trainer / render stepper on that device (they run one
    after another).  torch hands out its 32 pooled high-priority streams round-robin, and which hardware queue a stream lands on
    depends on how many were handed out before: with one stream per Trainer the SAME workload ran at 0.39 or 0.73 ms per step from
    one instance to the next inside one process (round 5, tools/loop_variance.py)."""
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    st = _MARCH_STREAMS.get(key)
    if st is None or os.environ.get("NGP_SHARED_SIDE", "1") == "0":
        st = torch.cuda.Stream(device=dev, priority=int(os.environ.get("NGP_MARCH_PRIORITY", "-1")))
        _MARCH_STREAMS[key] = st
    return st


class Trainer:
    def __init__(self, model, lr=1e-2, num_epochs=30, steps_per_epoch=1000, T_threshold=1e-4,
                 lambda_opacity=1e-3, grad_scale=1.0, warmup_steps=256, update_interval=16, overlap_march=True,
                 lambda_distortion=0.0, binned_backward=None, erode=False, native_step=None, loss_scaler=True):
        self.model = model
        if not hasattr(model, "density_grid"):
            model.register_training_buffers()
        self.opt = FusedAdam(model, lr=lr, eps=1e-15)
        self.base_lr, self.num_epochs, self.steps_per_epoch = lr, num_epochs, steps_per_epoch
        self.T_threshold, self.lambda_opacity = T_threshold, lambda_opacity
        self.warmup_steps, self.update_interval = warmup_steps, update_interval
        self.grad_scale = grad_scale
        # Dynamic loss scale of the native step: torch.cuda.amp.GradScaler's rule on the device, on top of tiny-cuda-nn's fixed 128 --
        # what Lightning's precision=16 gives the reference (train.py:274).  True: GradScaler's defaults (init 65536, growth 2 every
        # 2000 clean steps, backoff 0.5); a dict(init_scale=, growth_factor=, backoff_factor=, growth_interval=) overrides them; False /
        # None: the fixed scale alone (rounds 1-5: half of a step's feature gradients flush to zero in f16, profiles/r06_loss_scale.txt).
        # Steps whose update runs through Python hooks (grad_hook / mlp_grad_hook / a replaced optimizer step) keep the fixed scale.
        self.loss_scaler = None
        if loss_scaler:
            self.loss_scaler = dict(init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000)
            if isinstance(loss_scaler, dict):
                self.loss_scaler.update(loss_scaler)
        model.native_loss_scaler = dict(self.loss_scaler) if self.loss_scaler else None       # render()'s native node (step_autograd): the same rule
        self._scaler_active = False          # what the current stepper is configured with
        self._scaler_saved = None            # scale carried over a rebuild of the stepper
        self.global_step = 0
        self.exp_step_factor = 1 / 256 if model.scale > 0.5 else 0.0      # train.py:95-96
        dev = model.center.device
        self.bg = torch.ones(3, device=dev) if self.exp_step_factor == 0 else None
        self.lambda_distortion = lambda_distortion                          # opt.py:25-29 suggests 1e-3 for real scenes
        self.loss_fn = NeRFLoss(lambda_opacity=lambda_opacity, lambda_distortion=lambda_distortion)
        self._dist_seed = None
        # table backward: the binned variant (exact fixed-point sums, deterministic; measured 5 % faster per step) unless
        # NGP_BINNED_BWD=0 / binned_backward=False selects the one-pass sliced kernel
        self.binned_backward = bool(int(os.environ.get("NGP_BINNED_BWD", "1"))) if binned_backward is None else binned_backward
        self._buf = None                 # StepBuffers of the current batch size
        self._grid_step = -1             # global_step the occupancy grid was last brought up to date for
        self.loss_scale = tcnn.LOSS_SCALE   # factor on dL/dsigma, dL/drgb inside the f16 backward; lowered to 128 / world under DDP
        # train.py:160-163: erode = (dataset_name == 'colmap'), i.e. the unbounded real scenes; needs NGP.mark_invisible_cells
        self.erode = erode
        # The marching stream must land on a hardware queue that runs NEXT TO the main stream's or nothing overlaps (a
        # default-priority stream was observed to share the main stream's queue once RCCL had created its own: step 0.52 -> 0.89
        # ms): a high-priority stream, and ONE per process (marching_stream above).
        self.side = marching_stream(dev) if (overlap_march and dev.type == "cuda") else None
        # seed of the march's jitter draws (custom_functions.py:83: every rank's torch.rand_like draws from its own generator): the rank
        # is mixed in so that data-parallel ranks do not jitter ray slot r alike at every step
        rank = 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank = torch.distributed.get_rank()
        self.noise_seed = (int(os.environ.get("NGP_NOISE_SEED", "20240924")) + 0x632BE59BD9B4E019 * rank) & 0xFFFFFFFFFFFFFFFF
        # where in the step the next batch's march is enqueued is the native stepper's business (NGP_MARCH_AT, csrc/stepper.hip;
        # default: behind the field forward.  profiles/archive_r01_r04/r02_march_sweep.txt, r03_march_sweep.txt: wherever it lands the march costs
        # the kernels next to it 17-20 us)
        self.last = {}
        self.grad_hook = None    # called between backward and optimizer (multi-GPU gradient all-reduce)
        self.mlp_grad_hook = None  # called once the MLP gradients exist, before the hash-grid backward is enqueued
        self.group_hook = None     # called behind each launch group of the table backward: (group, n_groups, entry_begin, entry_end)
        self.bwd_groups = 1        # launch groups of the table backward when a group_hook is installed
        self.update_hook = None    # replaces the whole-table Adam: (lr, step, grad_scale, found_inf, stream) -- ddp.ShardedExchange
        self.native_exchange = None  # ddp.NativeExchange: the step's tail (exchange + update) is ONE library call, ngp_stepper_tail
        self._group_cache = {}
        self.events = None       # not None: stage timing is on (bench.py's roofline leg reads stage_times_ms() after each step)
        self.native_step = True  # the step is enqueued by the native stepper (the Python-enqueued copy of it was removed in round 5)
        if native_step is False:
            raise ValueError("Trainer(native_step=False): the Python-enqueued step was removed; the native stepper is the step")
        self._stepper = None         # ngp_stepper handle
        self._stepper_key = None     # what it was built for (pointers, recipe)
        self._pending_key = None     # (rays_o ptr, rays_d ptr) of the batch whose march the native stepper holds
        self._pending_keep = None    # ... and the tensors themselves (alive until consumed)
        self._timing_on = False
        self._late_march = os.environ.get("NGP_MARCH_AT", "mlp_fwd") in ("hashgrid_bwd", "adam")

    # -- stage timing ----------------------------------------------------------------------------
    STAGES = ("march_write", "hashgrid_fwd", "mlp_fwd", "composite_fw+loss", "composite_bw", "mlp_bwd", "hashgrid_bwd", "adam",
              "march_count(side stream)")

    def stage_times_ms(self):
        """Elapsed time between consecutive stage marks of the last profiled step (syncs)."""
        if self._stepper is None:
            return []
        ms = (C.c_float * len(self.STAGES))()
        call("ngp_stepper_stage_times", self._stepper, ms)
        return [(name, float(t)) for name, t in zip(self.STAGES, ms) if t >= 0]

    # -- pieces --------------------------------------------------------------------------------
    def _device_guard(self, dev):
        """`with torch.cuda.device(dev)` only when dev is not the current device already (one process per GPU sets its device
        once; the guard's bookkeeping is ~10 us of host time per step otherwise)."""
        if dev.index is None or torch.cuda.current_device() == dev.index:
            return contextlib.nullcontext()
        return torch.cuda.device(dev)

    def _maybe_update_grid(self):
        if self.global_step % self.update_interval == 0:                  # train.py:160-163
            self.model.update_density_grid(0.01 * MAX_SAMPLES / 3 ** 0.5, warmup=self.global_step < self.warmup_steps,
                                           erode=self.erode)

    def buffers(self, n_rays):
        """The step's preallocated buffers for batches of n_rays (built on first use, rebuilt if the batch size changes)."""
        if self._buf is None or self._buf.n != n_rays:
            if self._stepper is not None:
                call("ngp_stepper_drop_pending", self._stepper)
                self._pending_key = self._pending_keep = None
            torch.cuda.synchronize()
            self._buf = None                                   # release the old arena before the new one is requested
            self._buf = StepBuffers(self.model, n_rays, self.lambda_distortion > 0, self.binned_backward)
            if self._stepper is not None:
                bc = self._buf.c_struct(self.lambda_distortion > 0)
                call("ngp_stepper_set_buffers", self._stepper, C.byref(bc))
                self._buf.attach_sample_sets(self._stepper)
        return self._buf

    def host_times(self, reset=True):
        """(wait_ms, enqueue_ms) per step the native stepper's entry points spent on the host since the last reset: polling for the
        march's sample count (device-bound) / everything else (launches, events, checks)."""
        if self._stepper is None:
            return None
        w, e, n = C.c_double(), C.c_double(), C.c_longlong()
        call("ngp_stepper_host_times", self._stepper, C.byref(w), C.byref(e), C.byref(n), 1 if reset else 0)
        k = max(n.value, 1)
        return {"wait_ms_per_step": w.value / k * 1e3, "enqueue_ms_per_step": e.value / k * 1e3, "steps": n.value}

    def last_march_noise(self):
        """The jitter values (R) the march of the last stepped batch used (a view of the step buffers; tests hand it to another path)."""
        return self._buf.noise[call("ngp_stepper_last_set", self._stepper)]

    @property
    def has_pending(self):
        """A march of the next batch has been enqueued ahead of its step."""
        return self._pending_key is not None

    # -- the native stepper ----------------------------------------------------------------------
    def _native_stepper(self, B):
        """The ngp_stepper for the model's current buffers (created on first use; rebuilt when a pointer it holds moved)."""
        m = self.model
        enc, net = m.xyz_encoder, m.rgb_net
        eh, rh = enc._half.get(enc.params), net._half.get(net.params)
        (em, ev), (rm, rv) = self.opt.moments("enc"), self.opt.moments("rgb")
        g16 = m._grid_grad16(enc.params.device)
        key = (enc.params.data_ptr(), eh.data_ptr(), em.data_ptr(), ev.data_ptr(), net.params.data_ptr(), rh.data_ptr(), rm.data_ptr(),
               rv.data_ptr(), g16.data_ptr(), m.density_bitfield.data_ptr(), m.center.data_ptr(), self.lambda_distortion)
        if self._stepper is not None and key == self._stepper_key:
            return self._stepper
        self._destroy_stepper()
        c = _lib.StepperConfig()
        c.center, c.half_size, c.xyz_min, c.xyz_max = ptr(m.center), ptr(m.half_size), ptr(m.xyz_min), ptr(m.xyz_max)
        c.density_bitfield = ptr(m.density_bitfield)
        c.cascades, c.grid_size, c.scale, c.exp_step_factor = m.cascades, m.grid_size, float(m.scale), float(self.exp_step_factor)
        c.meta = enc.meta
        c.enc_param, c.enc_half, c.enc_m, c.enc_v = ptr(enc.params), ptr(eh), ptr(em), ptr(ev)
        c.rgb_param, c.rgb_half, c.rgb_m, c.rgb_v = ptr(net.params), ptr(rh), ptr(rm), ptr(rv)
        c.n_grid, c.n_density, c.n_rgb, c.grid_grad16 = enc.n_grid, enc.n_mlp, net.params.numel(), ptr(g16)
        c.max_samples, c.near_distance, c.T_threshold = MAX_SAMPLES, NEAR_DISTANCE, self.T_threshold
        c.lambda_opacity, c.lambda_distortion, c.bg = self.lambda_opacity, self.lambda_distortion, ptr(self.bg)
        b1, b2 = self.opt.betas
        c.beta1, c.beta2, c.eps, c.weight_decay = b1, b2, self.opt.eps, self.opt.weight_decay
        c.noise_seed = self.noise_seed
        bc = B.c_struct(self.lambda_distortion > 0)
        h = C.c_void_p()
        call("ngp_stepper_create", C.byref(c), C.byref(bc), C.byref(h))
        self._stepper, self._stepper_key = h, key
        B.attach_sample_sets(h)
        self._pending_key = self._pending_keep = None
        self._timing_on = False
        if self.native_exchange is not None:
            self.native_exchange.attach(h)
        return h

    def loss_scale_state(self):
        """(scale the next step multiplies tiny-cuda-nn's 128 by, clean steps since it last changed); (1.0, 0) while the scaler is off.  Syncs."""
        if self._stepper is None or not self._scaler_active:
            return (self._scaler_saved or 1.0), 0
        sc, tr = C.c_float(0.0), C.c_int32(0)
        call("ngp_stepper_loss_scale", self._stepper, C.byref(sc), C.byref(tr), stream())
        return float(sc.value), int(tr.value)

    def _configure_scaler(self, h, want):
        if want:
            cfg = dict(self.loss_scaler)
            if self._scaler_saved:
                cfg["init_scale"] = self._scaler_saved
            call("ngp_stepper_set_loss_scaler", h, float(cfg["init_scale"]), float(cfg["growth_factor"]), float(cfg["backoff_factor"]),
                 int(cfg["growth_interval"]), stream())
        else:
            if self._scaler_active:
                self._scaler_saved = self.loss_scale_state()[0]
            call("ngp_stepper_set_loss_scaler", h, 0.0, 2.0, 0.5, 2000, stream())
        self._scaler_active = want

    def _destroy_stepper(self):
        if self._stepper is not None:
            if self._scaler_active:
                self._scaler_saved = self.loss_scale_state()[0]
                self._scaler_active = False
            _lib.lib().ngp_stepper_destroy(self._stepper)
            self._stepper = self._stepper_key = None
            self._pending_key = self._pending_keep = None

    def __del__(self):
        try:
            self._destroy_stepper()
        except Exception:            # noqa: BLE001 -- interpreter shutdown
            pass

    @torch.no_grad()
    def _step_native(self, rays_o, rays_d, rgb_gt, next_batch):
        """`step` through the native stepper: march hand-over, front, [MLP-gradient hook], table backward, [grid-gradient hook],
        Adam -- three library calls without hooks."""
        m = self.model
        dev = rays_o.device
        enc, net = m.xyz_encoder, m.rgb_net
        with self._device_guard(dev):
            mq = stream()                       # raw handle of torch's current stream (0.3 us; torch.cuda.current_stream() costs 12)
            sq = self.side.cuda_stream if self.side is not None else mq
            n = rays_o.shape[0]
            if not (rays_o.is_contiguous() and rays_d.is_contiguous() and rgb_gt.is_contiguous()):
                rays_o, rays_d, rgb_gt = rays_o.contiguous(), rays_d.contiguous(), rgb_gt.contiguous()
            had_pending = self._pending_key is not None
            B = self.buffers(n)                      # (a batch-size change drops the pending march)
            h = self._native_stepper(B)
            timing = self.events is not None
            if timing != self._timing_on:
                call("ngp_stepper_timing", h, 1 if timing else 0); self._timing_on = timing
            ro_p, rd_p = rays_o.data_ptr(), rays_d.data_ptr()
            if self._pending_key != (ro_p, rd_p):
                if self._pending_key is not None:
                    call("ngp_stepper_drop_pending", h)
                if not had_pending or self._grid_step != self.global_step:
                    self._maybe_update_grid(); self._grid_step = self.global_step
                call("ngp_stepper_march", h, ro_p, rd_p, mq, sq)
            self._pending_key = self._pending_keep = None
            # march of the next batch: concurrent with this step unless the occupancy grid is due for an update first
            # (that needs this step's optimizer result)
            next_needs_update = (self.global_step + 1) % self.update_interval == 0
            prefetch = next_batch is not None and not next_needs_update and next_batch[0].shape[0] == n
            no_p = nd_p = None
            if next_batch is not None:
                if not (next_batch[0].is_contiguous() and next_batch[1].is_contiguous()):
                    next_batch = (next_batch[0].contiguous(), next_batch[1].contiguous())
                no_p, nd_p = next_batch[0].data_ptr(), next_batch[1].data_ptr()
            use_dist = self.lambda_distortion > 0
            if use_dist:
                seed_val = self.lambda_distortion / n * self.grad_scale
                if B.dist_seed_val != seed_val:
                    B.view("dist_seed", torch.float32, n).fill_(seed_val); B.dist_seed_val = seed_val
            # hooks (multi-GPU exchange through Python), an update hook, or an optimizer whose step() was replaced (tests capture the
            # gradients there): the tail runs through Python, at the fixed loss scale
            custom_opt = "step" in vars(self.opt) and not getattr(self.opt.step, "_wrapped_by_lr_sched", False)     # (an LR scheduler wraps step(): still ours)
            hooks = self.grad_hook is not None or self.mlp_grad_hook is not None or custom_opt
            want_scaler = self.loss_scaler is not None and (self.native_exchange is not None or not hooks)
            if want_scaler != self._scaler_active:
                self._configure_scaler(h, want_scaler)
            S_c, np_c = C.c_int32(0), C.c_int32(0)
            call("ngp_stepper_front", h, ro_p, rd_p, rgb_gt.data_ptr(), no_p if prefetch else None, nd_p if prefetch else None,
                 self.loss_scale, self.grad_scale, mq, sq, C.byref(S_c), C.byref(np_c))
            if prefetch:
                self._pending_key, self._pending_keep = (no_p, nd_p), next_batch
            S, n_part = S_c.value, np_c.value
            # (hooks: the tail runs through Python; otherwise table backward + Adam are two more library calls)
            if self.native_exchange is not None:
                # data parallel, enqueued by the library: MLP all-reduce, table backward in launch groups with the chunks of the
                # gradient handed to the communicator's stream behind them, non-finite checks, Adam, all-gather -- one call,
                # whether or not THIS rank's batch had samples (it joins every collective with zeros)
                epoch = self.global_step // self.steps_per_epoch
                lr = self.opt.param_groups[0]["lr"] = cosine_lr(self.base_lr, epoch, self.num_epochs)
                self.opt.t += 1
                call("ngp_stepper_tail", h, lr, self.opt.t, self.loss_scale * self.native_exchange.world * self.grad_scale, mq)
                self.native_exchange.stepped()
                enc._half.mark_fresh(enc.params); net._half.mark_fresh(net.params)
            elif S > 0:
                epoch = self.global_step // self.steps_per_epoch
                lr = self.opt.param_groups[0]["lr"] = cosine_lr(self.base_lr, epoch, self.num_epochs)
                if not hooks:
                    self.opt.t += 1
                    # table backward + fused Adam in ONE call (the dense levels' merge folded into the Adam launch).  The library's
                    # overflow guard decides on the device whether the step applies (a non-finite weight gradient: skipped, as
                    # GradScaler does for the reference, train.py:274); the count of APPLIED steps lives next to it (step_state)
                    if self.opt._step_state is None:
                        self.opt.ensure_step_state(self.opt.t - 1)
                    call("ngp_stepper_backward_update", h, lr, self.opt.t, self.loss_scale * self.grad_scale, self.opt._step_state.data_ptr(), mq)
                    enc._half.mark_fresh(enc.params); net._half.mark_fresh(net.params)
                else:
                    g16 = m._grid_grad16(dev)
                    native = dict(grid16=g16, density_partials=B.view("partials", torch.float32, n_part * enc.n_mlp),
                                  rgb_partials=B.arena[B.off["partials"] + 4 * n_part * enc.n_mlp:
                                                       B.off["partials"] + 4 * n_part * B.n_mlp_params].view(torch.float32),
                                  n_partials=n_part, scale=self.loss_scale)
                    m._native = native
                    if self.mlp_grad_hook is not None:
                        self.mlp_grad_hook()
                    ng = self.bwd_groups if (self.group_hook is not None and self.grad_hook is not None and 0 < S <= B.bin_max) else 1
                    for g in range(ng):
                        call("ngp_stepper_table_backward", h, ng, g, mq)
                        if ng > 1:
                            self.group_hook(g, ng, *self._group_entries(enc.meta, S, ng, g))
                    found_inf = self.grad_hook() if self.grad_hook is not None else None
                    nat = m._native
                    if self.update_hook is not None:
                        self.opt.t += 1
                        self.update_hook(lr, self.opt.t, nat["scale"] * self.grad_scale, found_inf, mq)
                    elif custom_opt:
                        self.opt.step(grad_scale=self.grad_scale, found_inf=found_inf, stream_handle=mq)
                    else:
                        self.opt.t += 1
                        call("ngp_stepper_update", h, lr, self.opt.t, nat["scale"] * self.grad_scale, ptr(nat["density_partials"]),
                             ptr(nat["rgb_partials"]), nat["n_partials"], ptr(found_inf), self.opt.step_state(found_inf), mq)
                        enc._half.mark_fresh(enc.params); net._half.mark_fresh(net.params)
                        m._native = None
            elif self.grad_hook is not None or self.mlp_grad_hook is not None:
                # no samples on THIS rank: the other ranks still expect it in the gradient collectives
                self._exchange_and_update(self.zero_native(dev), None, mq)
            if prefetch and self._late_march and not call("ngp_stepper_pending", h, no_p, nd_p):
                call("ngp_stepper_march", h, no_p, nd_p, mq, sq)       # a late placement (NGP_MARCH_AT=hashgrid_bwd / adam) whose stage did not run natively
            self.global_step += 1
            self.last = dict(stats=B.stats, rm_samples=S, total=B.total, n_rays=n, rgb=B.rgb, opacity=B.opacity,
                             distortion=B.dist if (S > 0 and use_dist) else None, n_active=B.n_active)
            if next_batch is not None and next_needs_update and next_batch[0].shape[0] == n:
                self._maybe_update_grid(); self._grid_step = self.global_step
                call("ngp_stepper_march", h, no_p, nd_p, mq, sq)
                self._pending_key, self._pending_keep = (no_p, nd_p), next_batch
        return self.last

    # -- the hot path --------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, rays_o, rays_d, rgb_gt, next_batch=None):
        """One optimisation step on a batch of rays.  `next_batch` = (rays_o, rays_d) of the
        following step, if known: its march overlaps this step's kernels.  The tensors in the returned
        record are views of the step's preallocated buffers: valid until the next step."""
        if not rays_o.is_cuda:
            raise RuntimeError("Trainer.step: rays on %s -- the product path has no CPU fallback" % rays_o.device)
        return self._step_native(rays_o, rays_d, rgb_gt, next_batch)

    def _group_entries(self, meta, S, n_groups, group):
        """Table-entry range [begin, end) that launch group `group` of `n_groups` completes (host-side plan, cached)."""
        key = (n_groups, group)
        if key not in self._group_cache:
            a, b = C.c_int64(), C.c_int64()
            call("ngp_hashgrid_bwd_binned_group_entries", C.byref(meta), 1, n_groups, group, C.byref(a), C.byref(b))
            self._group_cache[key] = (a.value, b.value)
        return self._group_cache[key]

    def zero_native(self, dev):
        """The native gradient record of a rank whose batch produced no samples: zeros, at this trainer's loss scale."""
        m = self.model
        enc, net = m.xyz_encoder, m.rgb_net
        f32 = dict(dtype=torch.float32, device=dev)
        return dict(grid16=m._grid_grad16(dev).zero_(), density_partials=torch.zeros(enc.n_mlp, **f32),
                    rgb_partials=torch.zeros(net.params.numel(), **f32), n_partials=1, scale=self.loss_scale)

    def _exchange_and_update(self, native, table_backward, mq):
        """Tail of the step once the MLP backward has left its partial sums in `native`:
        [MLP-gradient collective, asynchronous] -> table backward (fills native['grid16']) -> lr schedule ->
        [grid-gradient collective + non-finite check] -> fused Adam on the native buffers.
        The bracketed hooks are set under multi-GPU (ngp_pl_amd/ddp.py); device agnostic (tests/test_ddp_gloo.py drives
        it with CPU tensors)."""
        if getattr(self, "native_exchange", None) is not None:
            raise RuntimeError("a NativeExchange is installed: the step's tail is ngp_stepper_tail, not the hook-driven Python tail")
        self.model._native = native
        if self.mlp_grad_hook is not None:
            self.mlp_grad_hook()
        if table_backward is not None:
            table_backward()
        epoch = self.global_step // self.steps_per_epoch
        self.opt.param_groups[0]["lr"] = cosine_lr(self.base_lr, epoch, self.num_epochs)
        found_inf = None
        if self.grad_hook is not None:
            found_inf = self.grad_hook()
        if getattr(self, "update_hook", None) is not None:
            self.opt.t += 1
            self.update_hook(self.opt.param_groups[0]["lr"], self.opt.t, self.model._native["scale"] * self.grad_scale, found_inf, mq)
        else:
            self.opt.step(grad_scale=self.grad_scale, found_inf=found_inf, stream_handle=mq)

    def skipped_steps(self):
        """Optimizer steps the overflow guard (or a caller's skip flag) did not apply, as (MLP blocks, grid block) -- syncs."""
        a_mlp, a_grid = self.opt.applied_steps()
        return self.opt.t - a_mlp, self.opt.t - a_grid

    def metrics(self):
        """Host-side readout of the last step (syncs): loss, psnr, rm_s, vr_s as train.py:177-183 logs them."""
        st = self.last["stats"].tolist()
        n = self.last["n_rays"]
        mse = st[1] / (3 * n)
        if self.last.get("distortion") is not None:
            st[0] += self.lambda_distortion * float(self.last["distortion"].mean())
        return dict(loss=st[0], psnr=-10 * math.log10(max(mse, 1e-12)), rm_s=self.last["rm_samples"] / n,
                    vr_s=float(self.last["total"].sum().item()) / n)

    # -- the reference-shaped path ---------------------------------------------------------------
    def step_autograd(self, rays_o, rays_d, rgb_gt, noise=None, next_batch=None):
        """render() -> NeRFLoss -> backward -> FusedAdam, as train.py:159-185.  The GradScaler Lightning's precision=16 wraps around
        that step (train.py:274) lives on the device here: render()'s native node runs its f16 backward under the dynamic loss scale
        (`model.native_loss_scaler`), FusedAdam unscales, skips and updates the scale in its launch -- no host sync."""
        if self.grad_hook is not None or self.mlp_grad_hook is not None or self.update_hook is not None or self.native_exchange is not None:
            # the autograd surface issues no collectives: under an installed exchange the ranks would train independently and
            # silently diverge (Trainer.step is the data-parallel path; uninstall() the exchange for single-process use)
            raise RuntimeError("step_autograd() with a gradient exchange installed: the reference-shaped path is single-process; "
                               "use Trainer.step under data parallelism or exchange.uninstall(trainer) first")
        self._maybe_update_grid()
        kwargs = {"test_time": False}
        if self.exp_step_factor:
            kwargs["exp_step_factor"] = self.exp_step_factor
        if noise is not None:
            kwargs["noise"] = noise          # (R) jitter of the march instead of a fresh torch.rand draw
        if next_batch is not None and (self.global_step + 1) % self.update_interval != 0:
            kwargs["next_rays"] = (next_batch[0], next_batch[1])     # marched under this step (not across an occupancy update)
        results = render(self.model, rays_o, rays_d, **kwargs)
        loss_d = self.loss_fn(results, {"rgb": rgb_gt})
        loss = sum(lo.mean() for lo in loss_d.values())
        loss.backward()
        epoch = self.global_step // self.steps_per_epoch
        self.opt.param_groups[0]["lr"] = cosine_lr(self.base_lr, epoch, self.num_epochs)
        self.opt.step()
        self.global_step += 1
        return results, loss
