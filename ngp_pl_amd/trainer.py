"""Lightning-free driver of one training step, mirroring NeRFSystem.training_step
(/root/reference/train.py:159-185) + configure_optimizers (:112-139) for the default recipe
(Synthetic-NeRF: scale 0.5, white background, no distortion loss, no pose optimisation).

Two implementations of the same step:
  * `step()`          -- the hot path: direct calls into libngp_hip.so, no autograd graph, native
                         gradient buffers consumed by optim.FusedAdam.  12 kernel launches and one
                         8-byte host read (the packed sample count).  The ray march of step k+1
                         only needs the occupancy bitfield, so it runs on a SECOND HIP stream
                         concurrently with step k's encode/MLP/backward kernels (the march is a
                         latency-bound chain of dependent loads that occupies a fraction of the
                         CUs) and its count lands in pinned host memory before step k+1 starts.
  * `step_autograd()` -- the same maths through render() + NeRFLoss + torch autograd, i.e. what
                         the reference's train.py drives; used to check the hot path (tests).
"""
import ctypes as C
import math
import os

import torch

from . import _lib, tcnn
from ._lib import call, ptr, stream
from .losses import NeRFLoss
from .optim import FusedAdam, cosine_lr
from .rendering import MAX_SAMPLES, NEAR_DISTANCE, render


class Trainer:
    def __init__(self, model, lr=1e-2, num_epochs=30, steps_per_epoch=1000, T_threshold=1e-4,
                 lambda_opacity=1e-3, grad_scale=1.0, warmup_steps=256, update_interval=16, overlap_march=True,
                 lambda_distortion=0.0, binned_backward=None):
        self.model = model
        if not hasattr(model, "density_grid"):
            model.register_training_buffers()
        self.opt = FusedAdam(model, lr=lr, eps=1e-15)
        self.base_lr, self.num_epochs, self.steps_per_epoch = lr, num_epochs, steps_per_epoch
        self.T_threshold, self.lambda_opacity = T_threshold, lambda_opacity
        self.warmup_steps, self.update_interval = warmup_steps, update_interval
        self.grad_scale = grad_scale
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
        self._bin_ws = None
        self._fw_ws = None               # per-row loss terms of ngp_composite_train_fw_loss
        # The marching stream must land on its own hardware queue or nothing overlaps: HIP multiplexes streams onto
        # a few HSA queues (GPU_MAX_HW_QUEUES, default 4) round-robin, and once RCCL has created its streams a
        # default-priority stream was observed to share the main stream's queue (rocprofv3: every kernel on one
        # queue_id, step 0.52 -> 0.89 ms).  High-priority streams are served from a separate queue.
        self.side = torch.cuda.Stream(device=dev, priority=int(os.environ.get("NGP_MARCH_PRIORITY", "-1"))) if (overlap_march and dev.type == "cuda") else None
        self._pending = None     # marched-but-not-consumed batch
        # where in the step the next batch's march is enqueued (it starts behind whatever the main stream has queued by
        # then): "top" = next to the hash forward (default), "hashgrid_fwd" / "mlp_fwd" / "mlp_bwd" / "hashgrid_bwd" = behind
        # that stage.  NGP_MARCH_LATE=1 is "mlp_bwd" (next to the table backward's latency-bound slice owners).
        self.march_at = os.environ.get("NGP_MARCH_AT", "mlp_bwd" if int(os.environ.get("NGP_MARCH_LATE", "0")) else "top")
        if self.march_at not in ("top", "hashgrid_fwd", "mlp_fwd", "mlp_bwd", "hashgrid_bwd"):
            raise ValueError("NGP_MARCH_AT: unknown stage %r" % self.march_at)
        self.last = {}
        self.grad_hook = None    # called between backward and optimizer (multi-GPU gradient all-reduce)
        self.mlp_grad_hook = None  # called once the MLP gradients exist, before the hash-grid backward is enqueued
        self.events = None       # list of (stage, event) when stage timing is on (bench.py roofline)
        self.march_ms = None
        self._zeros = None

    # -- stage timing ----------------------------------------------------------------------------
    def _mark(self, name):
        if self.events is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()           # torch's current stream == the stream every kernel here is launched on
            self.events.append((name, e))

    def stage_times_ms(self):
        """Elapsed time between consecutive stage marks of the last profiled step (syncs)."""
        torch.cuda.synchronize()
        ev = self.events
        out = [(ev[i + 1][0], ev[i][1].elapsed_time(ev[i + 1][1])) for i in range(len(ev) - 1)]
        if self.march_ms is not None:
            out.append(("march_count(side stream)", self.march_ms[0].elapsed_time(self.march_ms[1])))
        return out

    # -- pieces --------------------------------------------------------------------------------
    def _maybe_update_grid(self):
        if self.global_step % self.update_interval == 0:                  # train.py:160-163
            self.model.update_density_grid(0.01 * MAX_SAMPLES / 3 ** 0.5, warmup=self.global_step < self.warmup_steps)

    def _march(self, rays_o, rays_d):
        """AABB + near clamp + pass 1 of the march (+ ray-ordered scan).  Enqueued on the side
        stream behind everything the main stream has queued so far; the packed sample count
        lands in pinned host memory."""
        m = self.model
        n, dev = rays_o.shape[0], rays_o.device
        hits_t = torch.empty(n, 2, dtype=torch.float32, device=dev)
        rays_a = torch.empty(n, 3, dtype=torch.int64, device=dev)
        scratch = torch.empty(n * MAX_SAMPLES, dtype=torch.float32, device=dev)
        # {S, R} is written by the scan kernel straight into pinned (device-mapped) host memory: no copy kernel and no
        # extra launch between the march and the event the host waits on
        counter_host = torch.empty(2, dtype=torch.int32, pin_memory=True)
        main = torch.cuda.current_stream()
        st = self.side if self.side is not None else main
        if st is not main:
            ready = torch.cuda.Event(); ready.record(main)
            st.wait_event(ready)
        sq = st.cuda_stream                      # raw handle once: torch.cuda.current_stream() costs ~8 us per call
        with torch.cuda.stream(st):
            t0 = t1 = None
            if self.events is not None:
                t0 = torch.cuda.Event(enable_timing=True); t0.record()
            noise = torch.rand(n, dtype=torch.float32, device=dev)        # jitter of the first sample (custom_functions.py:83); drawn on the marching stream
            call("ngp_ray_aabb_near", ptr(rays_o), ptr(rays_d), ptr(m.center), ptr(m.half_size), NEAR_DISTANCE, n, ptr(hits_t), sq)
            call("ngp_raymarching_train_count", ptr(rays_o), ptr(rays_d), ptr(hits_t), ptr(m.density_bitfield), m.cascades,
                 float(m.scale), self.exp_step_factor, ptr(noise), m.grid_size, MAX_SAMPLES, n, ptr(rays_a), ptr(counter_host),
                 ptr(scratch), sq)
            if self.events is not None:
                t1 = torch.cuda.Event(enable_timing=True); t1.record()
            done = torch.cuda.Event(); done.record()
        return dict(rays_o=rays_o, rays_d=rays_d, rays_a=rays_a, counter_host=counter_host, scratch=scratch,
                    hits_t=hits_t, noise=noise, done=done, timing=(t0, t1) if t0 is not None else None)

    # -- the hot path --------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, rays_o, rays_d, rgb_gt, next_batch=None):
        """One optimisation step on a batch of rays.  `next_batch` = (rays_o, rays_d) of the
        following step, if known: its march overlaps this step's kernels."""
        m = self.model
        dev = rays_o.device
        enc, net = m.xyz_encoder, m.rgb_net
        with torch.cuda.device(dev):
            main = torch.cuda.current_stream()
            mq = main.cuda_stream
            if self._pending is not None and self._pending["rays_o"] is rays_o:
                rec = self._pending
            else:
                self._maybe_update_grid()
                rec = self._march(rays_o.contiguous(), rays_d.contiguous())
            self._pending = None
            n = rays_o.shape[0]
            # march of the next batch: concurrent with this step unless the occupancy grid is due
            # for an update first (that needs this step's optimizer result).  Enqueued BEFORE the
            # host blocks on this batch's march: the marching stream then runs the marches back to
            # back instead of idling for a host round trip (wake-up + enqueue) between them.
            next_needs_update = (self.global_step + 1) % self.update_interval == 0
            prefetch = next_batch is not None and not next_needs_update

            def march_next_if_at(stage):
                if prefetch and self.march_at == stage and self._pending is None:
                    self._pending = self._march(next_batch[0], next_batch[1])
            march_next_if_at("top")
            # the step's only host wait: the march of THIS batch.  Polled, not Event.synchronize(): the blocking wait
            # sleeps on an interrupt and wakes tens of microseconds late, which left the main stream idle at every step
            done = rec["done"]
            while not done.query():
                pass
            S = int(rec["counter_host"][0])
            # no main.wait_event(done): the host has just observed the event, so everything enqueued from here on is
            # ordered behind the march already; the barrier packet measured ~20 us of idle main stream per step
            self.march_ms = rec["timing"]
            if self.events is not None:
                self.events = []
            self._mark("start")
            f32 = dict(dtype=torch.float32, device=dev)
            f16 = dict(dtype=torch.float16, device=dev)
            xyzs = torch.empty(S, 3, **f32); dirs = torch.empty(S, 3, **f32)
            deltas = torch.empty(S, **f32); ts = torch.empty(S, **f32)
            call("ngp_raymarching_train_write", ptr(rec["rays_o"]), ptr(rec["rays_d"]), ptr(rec["rays_a"]), ptr(rec["scratch"]),
                 float(m.scale), self.exp_step_factor, m.grid_size, MAX_SAMPLES, n, ptr(xyzs), ptr(dirs), ptr(deltas), ptr(ts), mq)
            self._mark("march_write")
            rays_a = rec["rays_a"]
            eh, rh = enc._half.get(enc.params), net._half.get(net.params)
            feats = torch.empty(16, S, 2, **f16); h = torch.empty(S, 16, **f16)
            sigmas = torch.empty(S, **f32); rgbs = torch.empty(S, 3, **f32)
            total = torch.empty(n, dtype=torch.int64, device=dev)
            opacity = torch.empty(n, **f32); depth = torch.empty(n, **f32); rgb = torch.empty(n, 3, **f32); ws = torch.empty(S, **f32)
            stats = torch.empty(2, **f32)                  # loss, sum of squared error (written by ngp_composite_train_fw_loss)
            dL_drgb = torch.empty(n, 3, **f32); dL_dopacity = torch.empty(n, **f32)
            if self._zeros is None or self._zeros.shape[0] != n:
                self._zeros = torch.zeros(n, **f32)        # dL/ddepth: the loss has no depth term
            dL_ddepth = self._zeros
            ray_offs = torch.empty(n, dtype=torch.int32, device=dev); n_active = torch.empty(1, dtype=torch.int32, device=dev)
            dL_dsigmas = torch.empty(S, **f32); dL_drgbs = torch.empty(S, 3, **f32)
            if S > 0:
                call("ngp_hashgrid_fwd", ptr(xyzs), ptr(m.xyz_min), ptr(m.xyz_max), ptr(eh[enc.n_mlp:]), C.byref(enc.meta), S, ptr(feats), mq)
                self._mark("hashgrid_fwd")
                march_next_if_at("hashgrid_fwd")
                call("ngp_field_fwd", ptr(feats), ptr(dirs), ptr(eh), ptr(rh), S, ptr(sigmas), ptr(rgbs), ptr(h), mq)
                self._mark("mlp_fwd")
                march_next_if_at("mlp_fwd")
            # composite + per-ray loss seeds, then one small kernel: offsets of the live samples and the loss sums
            if self._fw_ws is None or self._fw_ws.numel() < 8 * (n + 3):
                self._fw_ws = torch.empty(8 * (n + 3), dtype=torch.uint8, device=dev)
            call("ngp_composite_train_fw_loss", ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(ts), ptr(rays_a), self.T_threshold, n, S,
                 ptr(total), ptr(opacity), ptr(depth), ptr(rgb), ptr(ws), ptr(ray_offs), ptr(n_active), ptr(rgb_gt), ptr(self.bg),
                 self.lambda_opacity, self.grad_scale, ptr(stats), ptr(stats[1:]), ptr(dL_drgb), ptr(dL_dopacity),
                 ptr(self._fw_ws), self._fw_ws.numel(), mq)
            self._mark("composite_fw+loss")
            if S > 0:
                # backward only over the samples up to each ray's early stop (the rest have zero gradient):
                # composite_fw counted them per ray, the scan above placed them, composite_bw lists them
                active = torch.empty(S, dtype=torch.int32, device=dev)
                dL_dws = dist = None
                if self.lambda_distortion > 0:
                    # losses.py:6-37,58-59: lambda * distortion per ray, mean over rays; its gradient enters the composite as dL/dws
                    dist = torch.empty(n, **f32); ws_incl = torch.empty(S, **f32); wts_incl = torch.empty(S, **f32)
                    call("ngp_distortion_loss_fw", ptr(ws), ptr(deltas), ptr(ts), ptr(rays_a), n, S, ptr(dist), ptr(ws_incl), ptr(wts_incl), mq)
                    seed_val = self.lambda_distortion / n * self.grad_scale
                    if self._dist_seed is None or self._dist_seed[0] != (n, seed_val):
                        self._dist_seed = ((n, seed_val), torch.full((n,), seed_val, **f32))
                    dL_dws = torch.empty(S, **f32)
                    call("ngp_distortion_loss_bw", ptr(self._dist_seed[1]), ptr(ws_incl), ptr(wts_incl), ptr(ws), ptr(deltas), ptr(ts),
                         ptr(rays_a), n, S, ptr(dL_dws), mq)
                # the binned table backward reads the live samples' positions as a stream: composite_bw copies them in list order
                nbytes = _lib.lib().ngp_hashgrid_bwd_binned_workspace_bytes(C.byref(enc.meta), S) if self.binned_backward else 0
                x_act = torch.empty(S, 3, **f32) if nbytes else None      # 0: batch too large for the binned variant (occupancy warm-up)
                call("ngp_composite_train_bw", ptr(dL_dopacity), ptr(dL_ddepth), ptr(dL_drgb), ptr(dL_dws), ptr(sigmas), ptr(rgbs), ptr(ws),
                     ptr(deltas), ptr(ts), ptr(rays_a), ptr(opacity), ptr(depth), ptr(rgb), self.T_threshold, n, S,
                     ptr(dL_dsigmas), ptr(dL_drgbs), ptr(ray_offs), ptr(active), ptr(xyzs) if nbytes else None, ptr(x_act), mq)
                self._mark("composite_bw")
                n_part = call("ngp_field_bwd_partials", S)
                partials = torch.empty(n_part * (enc.n_mlp + net.params.numel()), **f32)
                dh = torch.empty(S, 16, **f16); dfeats = torch.empty(16, S, 2, **f16)
                call("ngp_field_bwd", ptr(feats), ptr(dirs), ptr(h), ptr(eh), ptr(rh), ptr(dL_dsigmas), ptr(dL_drgbs), tcnn.LOSS_SCALE, S,
                     ptr(active), ptr(n_active), ptr(dh), ptr(dfeats), ptr(partials), mq)
                self._mark("mlp_bwd")
                march_next_if_at("mlp_bwd")
                g16 = m._grid_grad16(dev)
                m._native = dict(grid16=g16, density_partials=partials[:n_part * enc.n_mlp], rgb_partials=partials[n_part * enc.n_mlp:],
                                 n_partials=n_part, scale=tcnn.LOSS_SCALE)
                if self.mlp_grad_hook is not None:
                    self.mlp_grad_hook()
                if nbytes:
                    if self._bin_ws is None or self._bin_ws.numel() < nbytes:
                        self._bin_ws = torch.empty(int(nbytes * 1.25), dtype=torch.uint8, device=dev)
                    call("ngp_hashgrid_bwd_binned", ptr(x_act), ptr(m.xyz_min), ptr(m.xyz_max), ptr(dfeats), C.byref(enc.meta), S,
                         None, ptr(n_active), ptr(self._bin_ws), self._bin_ws.numel(), ptr(g16), mq)
                else:
                    call("ngp_hashgrid_bwd_sliced", ptr(xyzs), ptr(m.xyz_min), ptr(m.xyz_max), ptr(dfeats), C.byref(enc.meta), S,
                         ptr(active), ptr(n_active), ptr(g16), mq)
                self._mark("hashgrid_bwd")
                march_next_if_at("hashgrid_bwd")
                epoch = self.global_step // self.steps_per_epoch
                self.opt.param_groups[0]["lr"] = cosine_lr(self.base_lr, epoch, self.num_epochs)
                if self.grad_hook is not None:
                    self.grad_hook()
                self.opt.step(grad_scale=self.grad_scale, stream_handle=mq)
                self._mark("adam")
            elif self.grad_hook is not None or self.mlp_grad_hook is not None:
                # no samples on THIS rank: the other ranks still expect it in the gradient collectives (DDP semantics:
                # every rank joins every all-reduce), so it contributes zeros and applies the averaged update like them
                g16 = m._grid_grad16(dev).zero_()
                m._native = dict(grid16=g16, density_partials=torch.zeros(enc.n_mlp, **f32), rgb_partials=torch.zeros(net.params.numel(), **f32),
                                 n_partials=1, scale=tcnn.LOSS_SCALE)
                if self.mlp_grad_hook is not None:
                    self.mlp_grad_hook()
                epoch = self.global_step // self.steps_per_epoch
                self.opt.param_groups[0]["lr"] = cosine_lr(self.base_lr, epoch, self.num_epochs)
                if self.grad_hook is not None:
                    self.grad_hook()
                self.opt.step(grad_scale=self.grad_scale, stream_handle=mq)
            if prefetch and self._pending is None:       # a later stage was asked for and this batch had no samples
                self._pending = self._march(next_batch[0], next_batch[1])
            self.global_step += 1
            self.last = dict(stats=stats, rm_samples=S, total=total, n_rays=n, rgb=rgb, opacity=opacity,
                             distortion=dist if S > 0 else None)
            if next_batch is not None and next_needs_update:
                self._maybe_update_grid()
                self._mark("grid_update")
                self._pending = self._march(next_batch[0], next_batch[1])
        return self.last

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
    def step_autograd(self, rays_o, rays_d, rgb_gt):
        """render() -> NeRFLoss -> backward -> FusedAdam, as train.py:159-185 (no GradScaler: the
        tcnn modules carry their own loss scale)."""
        self._maybe_update_grid()
        kwargs = {"test_time": False}
        if self.exp_step_factor:
            kwargs["exp_step_factor"] = self.exp_step_factor
        results = render(self.model, rays_o, rays_d, **kwargs)
        loss_d = self.loss_fn(results, {"rgb": rgb_gt})
        loss = sum(lo.mean() for lo in loss_d.values())
        loss.backward()
        epoch = self.global_step // self.steps_per_epoch
        self.opt.param_groups[0]["lr"] = cosine_lr(self.base_lr, epoch, self.num_epochs)
        self.opt.step()
        self.global_step += 1
        return results, loss
