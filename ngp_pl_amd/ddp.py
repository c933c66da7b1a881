"""Multi-GPU gradient exchange of the native step: what Lightning's DDPPlugin does for the reference
(/root/reference/train.py:268-272, opt.py:42: one process per GPU, gradients averaged over ranks by an
all-reduce every step), applied to the native gradient buffers the fused backward leaves behind.

  * every rank marches / encodes / backpropagates its own batch of rays (per-rank sampler seeds, as every
    rank's dataloader workers draw their own np.random.choice, datasets/base.py:25-29);
  * the MLP gradients (10 240 f32, summed from the per-workgroup partial rows by two native launches) are
    all-reduced asynchronously as soon as the MLP backward has produced them, underneath the table backward;
  * the packed-f16 grid gradient (22.9 MB; DDP would move 45.7 MB of f32) is all-reduced as a SUM of
    gradients that every rank produced at loss scale 128 / world: the sum then sits at the single-GPU loss
    scale 128 x mean, i.e. the f16 headroom (65504 / 128) and the underflow floor do not depend on the world
    size.  The table backward accumulates in 64-bit fixed point and rounds to f16 once, so each rank's
    addend carries one f16 rounding; the ring adds world - 1 more (tiny-cuda-nn's own f16 atomics round
    after EVERY corner update);
  * the table backward runs in `n_groups` launch groups (ngp_hashgrid_bwd_binned_group) that each complete one contiguous
    range of the gradient table; the all-reduce of a finished range is issued asynchronously right behind its group, i.e.
    it runs on RCCL's stream while the slice owners of the next group still occupy the main stream (DDP's bucketed overlap,
    with the buckets aligned to what the backward finishes when); the last range and whatever is still in flight are
    waited for in front of the optimizer;
  * GradScaler's non-finite check runs on the reduced buffers (`ngp_found_inf`) and feeds the fused Adam's
    skip flag: a sum that overflowed on any rank is seen by all ranks alike (they hold the same sum), so the
    ranks stay in lock step without a second collective.

The host logic is device agnostic (the gloo test drives it with CPU tensors); kernels are used when the
buffers live on a GPU.
"""
import torch

from . import tcnn


class GradientExchange:
    def __init__(self, model, dist, world, group=None, n_groups=1, ranges=None):
        """`n_groups`: launch groups of the table backward = pieces of the grid-gradient exchange (1, the default: one
        all-reduce behind the whole backward.  Measured on MI355X with a 1-rank process group (profiles/r02_pg1_timeline.txt):
        every cross-stream hand-over -- the event the collective's stream waits on, the event the main stream waits on --
        idles the main stream for ~20 us, so a piece only pays where the transfer it hides is much longer than that).  `ranges`: their table-entry ranges [(begin, end), ...]; taken from the library's plan for the
        model's grid when not given.  EVERY rank issues the same sequence of collectives every step -- MLP block, then the
        ranges in order -- whichever path its own step took (binned groups, the one-pass fallback for oversized batches, or
        no samples at all): ranges a rank's backward did not hand over piecewise are issued in front of the optimizer."""
        self.model, self.dist, self.world, self.group = model, dist, world, group
        self.loss_scale = tcnn.LOSS_SCALE / world
        if ranges is None:
            ranges = self._plan_ranges(model, n_groups) if n_groups > 1 else []
        self.ranges = [tuple(r) for r in ranges]
        self.n_groups = max(len(self.ranges), 1)
        self._small = None
        self._flag = None
        self._work = None
        self._issued = 0           # ranges of this step already in flight
        self._works = []

    @staticmethod
    def _plan_ranges(model, n_groups):
        import ctypes as C
        from ._lib import call
        meta = model.xyz_encoder.meta
        out = []
        for g in range(n_groups):
            a, b = C.c_int64(), C.c_int64()
            call("ngp_hashgrid_bwd_binned_group_entries", C.byref(meta), 1, n_groups, g, C.byref(a), C.byref(b))
            out.append((a.value, b.value))
        return out

    def install(self, trainer):
        """Hooks into Trainer.step: MLP collective behind the MLP backward, grid collective + inf check behind the
        table backward; the backward runs at loss scale 128 / world."""
        trainer.loss_scale = self.loss_scale
        trainer.mlp_grad_hook = self.reduce_mlp
        trainer.grad_hook = self.reduce_grid
        if len(self.ranges) > 1:
            trainer.bwd_groups = len(self.ranges)
            trainer.group_hook = self.reduce_piece
        return self

    def uninstall(self, trainer):
        """Back to a single-process step (bench.py: the rank-0-only legs behind the timed region issue no collectives)."""
        trainer.loss_scale = tcnn.LOSS_SCALE
        trainer.mlp_grad_hook = trainer.grad_hook = trainer.group_hook = None
        trainer.bwd_groups = 1
        self._works, self._issued, self._work = [], 0, None

    def broadcast_parameters(self):
        """DDP's constructor broadcast: every rank starts from rank 0's parameters."""
        for p in self.model.parameters():
            if p.numel():                       # `dir_encoder.params` is empty (the SH encoding has no parameters)
                self.dist.broadcast(p.data, 0, group=self.group)
        for mod in (self.model.xyz_encoder, self.model.rgb_net):
            mod._half.invalidate()              # the f16 working copies follow the broadcast

    # -- hooks ---------------------------------------------------------------------------------
    def reduce_mlp(self):
        nat = self.model._native
        if nat is None:
            return
        enc, net = self.model.xyz_encoder, self.model.rgb_net
        n_d, n_r, n_part = enc.n_mlp, net.params.numel(), nat["n_partials"]
        dp, rp = nat["density_partials"], nat["rgb_partials"]
        if self._small is None or self._small.device != dp.device:
            self._small = torch.empty(n_d + n_r, dtype=torch.float32, device=dp.device)
        small = self._small
        if dp.is_cuda:                          # one launch into one buffer (the torch sum/sum/cat costs ~40 us of GPU time)
            from ._lib import call, ptr, stream
            call("ngp_reduce_partials2", ptr(dp), n_d, ptr(rp), n_r, n_part, ptr(small), stream())
        else:
            small[:n_d] = dp.view(n_part, n_d).sum(0); small[n_d:] = rp.view(n_part, n_r).sum(0)
        self._work = self.dist.all_reduce(small, group=self.group, async_op=True)

    def reduce_piece(self, group, n_groups, entry_begin, entry_end):
        """Behind launch group `group` of the table backward: all-reduce the table range it completed (2 features per entry)."""
        nat = self.model._native
        if nat is None:
            return
        if n_groups != len(self.ranges) or group != self._issued or (entry_begin, entry_end) != self.ranges[group]:
            raise RuntimeError("table backward group %d/%d [%d, %d) does not match the exchange plan %r" % (
                group, n_groups, entry_begin, entry_end, self.ranges))
        self._issue_next(nat["grid16"])

    def _issue_next(self, g16):
        a, b = self.ranges[self._issued]
        self._issued += 1
        if b > a:
            self._works.append(self.dist.all_reduce(g16[2 * a:2 * b], group=self.group, async_op=True))

    def reduce_grid(self):
        """Returns the device int32 found_inf flag (None on CPU tensors when everything is finite)."""
        nat = self.model._native
        if nat is None:
            return None
        enc = self.model.xyz_encoder
        if self._work is None:
            self.reduce_mlp()
        g16 = nat["grid16"]
        if self.ranges:
            if self.ranges[0][0] != 0 or self.ranges[-1][1] * 2 != g16.numel():
                raise RuntimeError("exchange plan %r does not cover the gradient table (%d entries)" % (self.ranges, g16.numel() // 2))
            while self._issued < len(self.ranges):        # ranges this rank's backward did not hand over piecewise
                self._issue_next(g16)
            for work in self._works:
                work.wait()
            self._works, self._issued = [], 0
        else:
            self.dist.all_reduce(g16, group=self.group)
        self._work.wait(); self._work = None
        small = self._small
        nat["density_partials"] = small[:enc.n_mlp]
        nat["rgb_partials"] = small[enc.n_mlp:]
        nat["n_partials"] = 1
        nat["scale"] = nat["scale"] * self.world          # sum of `world` gradients at scale 128 / world == 128 x their mean
        if g16.is_cuda:
            from ._lib import call, ptr, stream
            if self._flag is None or self._flag.device != g16.device:
                self._flag = torch.zeros(8, dtype=torch.int32, device=g16.device)     # two flags (16 bytes apart), used alternately
                self._flag_step = 0
            cur, nxt = self._flag[4 * (self._flag_step & 1):], self._flag[4 * ((self._flag_step + 1) & 1):]
            self._flag_step += 1
            call("ngp_found_inf2", ptr(g16), 0, g16.numel(), ptr(small), 1, small.numel(), ptr(cur), ptr(nxt), stream())
            return cur
        bad = not (bool(torch.isfinite(g16.float()).all()) and bool(torch.isfinite(small).all()))
        return torch.ones(1, dtype=torch.int32) if bad else None
