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
        all-reduce behind the whole backward.  Measured on MI355X with a 1-rank process group (profiles/archive_r01_r04/r02_pg1_timeline.txt):
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
        self.timing = False        # record device events around the exchange (bench.py's exchange_ms)
        self._t_events = []

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

    def info(self):
        """Who carries the collectives, as the communicator itself reports it (bench.py puts this in an N-GPU line)."""
        return {"comm": "torch.distributed (%s)" % self.dist.get_backend(self.group), "comm_ranks": int(self.dist.get_world_size(self.group))}

    def broadcast_parameters(self):
        """DDP's constructor broadcast: every rank starts from rank 0's parameters."""
        for p in self.model.parameters():
            if p.numel():                       # `dir_encoder.params` is empty (the SH encoding has no parameters)
                self.dist.broadcast(p.data, 0, group=self.group)
        for mod in (self.model.xyz_encoder, self.model.rgb_net):
            mod._half.invalidate()              # the f16 working copies follow the broadcast

    # -- timing --------------------------------------------------------------------------------
    def _t_begin(self, tensor):
        if self.timing and tensor.is_cuda:
            e = torch.cuda.Event(enable_timing=True); e.record()
            self._t_open = e

    def _t_end(self, tensor):
        if self.timing and tensor.is_cuda and getattr(self, "_t_open", None) is not None:
            e = torch.cuda.Event(enable_timing=True); e.record()
            self._t_events.append((self._t_open, e)); self._t_open = None

    def exchange_ms(self):
        """Mean device time per step between the start of the grid-gradient collective and the point where the optimizer may run
        (all-reduce exchange) or has run and the table is gathered (sharded exchange), over the steps recorded while `timing`."""
        if not self._t_events:
            return None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self._t_events]
        self._t_events = []
        return sum(ms) / len(ms)

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
        self._t_begin(g16)
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
            self._t_end(g16)
            return cur
        bad = not (bool(torch.isfinite(g16.float()).all()) and bool(torch.isfinite(small).all()))
        return torch.ones(1, dtype=torch.int32) if bad else None


class ShardedExchange(GradientExchange):
    """The exchange that scales: reduce-scatter of the packed-f16 grid gradient -> Adam on THIS rank's 1/world share of (f32 master,
    m, v) -> all-gather of the updated f16 working table.  The same bytes cross each link as with the all-reduce (a ring all-reduce
    IS a reduce-scatter + all-gather: 2 (world - 1) / world x 22.9 MB per rank and step), but the optimizer pass -- 52 us of an
    HBM-bound stream over 11.4 M parameters on one GPU -- is divided by the world size, and what is gathered is the table the next
    forward reads anyway.  The MLP blocks (10 240 parameters) keep the small asynchronous all-reduce and are updated by every rank.

    Semantics preserved from DDP (train.py:268-272): mean gradient over ranks, identical parameters on every rank after every
    step (tests/test_ddp_gloo.py: bit-identical f16 tables on all ranks, equal to the single-process update).  Two deliberate
    differences: (1) a rank's f32 master / m / v are only current inside its own pieces -- `gather_master()` makes the f32 master
    whole again (checkpoints; `state_dict()` callers); (2) the non-finite check decides per share (owner's reduced pieces) and per MLP
    block (the all-reduced sums every rank holds): every parameter is decided by one flag all its updaters agree on, so the ranks
    stay in lock step without a flag collective; GradScaler would skip the whole step.

    Layout (shared with the native tail, csrc/stepper.hip): the table is exchanged in `n_chunks` CHUNKS of world x `piece` f16 values,
    `piece` = ceil(n_grid / (n_chunks x world x 8)) x 8 (16-byte multiples); rank r owns values [c x chunk + r x piece, + piece) of
    every chunk c.  Gradient buffer and f16 working copy are re-seated on storages padded to n_chunks x chunk (the padding stays
    zero).  With n_chunks > 1 a chunk can be handed to the collective as soon as the table backward has completed it, while the
    backward of the following levels still runs; n_chunks = 1 is one reduce-scatter over the whole table.  Expected link traffic per
    step at world = 8: 2 x 7/8 x 22.9 MB = 40 MB per rank over the 7 xGMI links (~5.7 MB per link and direction each way)."""

    def __init__(self, model, dist, world, rank, group=None, adam=None, n_chunks=1):
        super().__init__(model, dist, world, group=group, n_groups=1)
        self.rank = rank
        enc = model.xyz_encoder
        self.n_grid = enc.n_grid
        if not 1 <= n_chunks <= 8:
            raise ValueError("n_chunks must be in 1..8")
        self.n_chunks = n_chunks
        self.piece = -(-self.n_grid // (n_chunks * world * 8)) * 8
        self.chunk = world * self.piece
        self.padded = n_chunks * self.chunk
        # this rank's pieces as [lo, hi) ranges of table values, clipped to the table
        self.pieces = []
        for c in range(n_chunks):
            lo = min(c * self.chunk + rank * self.piece, self.n_grid)
            self.pieces.append((lo, min(lo + self.piece, self.n_grid)))
        self.shard_len = self.piece                                  # (n_chunks = 1: the rank's shard, as before)
        self.lo, self.hi = self.pieces[0]
        self._shard16 = None
        self._flag_shard = None
        self._adam = adam if adam is not None else self._adam_kernel        # tests on CPU tensors inject a restatement

    # -- storage ---------------------------------------------------------------------------------
    def _seat(self, dev):
        """Gradient buffer and f16 working copy on storages padded to n_chunks x world x piece (once)."""
        m = self.model
        enc = m.xyz_encoder
        padded = self.padded
        if getattr(self, "_g_big", None) is None or self._g_big.device != dev:
            self._g_big = torch.zeros(padded, dtype=torch.float16, device=dev)
            m._g16 = self._g_big[:self.n_grid]
        if getattr(self, "_h_big", None) is None or self._h_big.device != dev:
            self._h_big = torch.zeros(enc.n_mlp + padded, dtype=torch.float16, device=dev)
            if hasattr(enc, "_half"):                    # the model's f16 working copy moves onto the padded storage
                self._h_big[:enc.n_mlp + self.n_grid].copy_(enc._half.get(enc.params))
                enc._half.t = self._h_big[:enc.n_mlp + self.n_grid]
                enc._half.mark_fresh(enc.params)
        if self._shard16 is None or self._shard16.device != dev:
            self._shard16 = torch.zeros(self.n_chunks * self.piece, dtype=torch.float16, device=dev)

    def install(self, trainer):
        super().install(trainer)
        trainer.update_hook = self.update
        trainer.bwd_groups, trainer.group_hook = 1, None
        self._trainer = trainer
        dev = self.model.xyz_encoder.params.device
        self._seat(dev)
        if getattr(trainer, "_stepper", None) is not None:
            trainer._destroy_stepper()          # it holds the old gradient / working-copy pointers
        return self

    def uninstall(self, trainer):
        """Back to whole-table updates: every rank gets the whole f32 master AND the whole Adam moments (a rank's m / v are only
        current inside its own pieces; whole-table Adam afterwards would otherwise resume with stale moments everywhere else)."""
        super().uninstall(trainer)
        trainer.update_hook = None
        self.gather_master()
        opt = getattr(trainer, "opt", None)
        if opt is not None and hasattr(opt, "moments"):
            for t in opt.moments("enc"):
                self._gather_shards(t)

    def broadcast_parameters(self):
        """DDP's constructor broadcast.  In sharded mode a rank's f32 master is only current inside its own pieces, and the f16
        working copies are re-cast from the master afterwards: make the master whole first."""
        if getattr(self, "_stepped", False):
            self.gather_master()
        super().broadcast_parameters()        # (the invalidated f16 copy is re-cast in place: it stays on the padded storage)

    # -- hooks -----------------------------------------------------------------------------------
    def reduce_grid(self):
        """Reduce-scatter of the grid gradient; returns (flag_mlp, flag_shard) device flags (None on CPU tensors when finite)."""
        nat = self.model._native
        if nat is None:
            return None
        enc = self.model.xyz_encoder
        if self._work is None:
            self.reduce_mlp()
        g16 = nat["grid16"]
        self._seat(g16.device)
        self._t_begin(g16)
        if g16.data_ptr() != self._g_big.data_ptr():
            self._g_big[:self.n_grid].copy_(g16)          # a producer that did not write into the seated buffer (zero_native)
        P, C = self.piece, self.chunk
        for c in range(self.n_chunks):
            self.dist.reduce_scatter_tensor(self._shard16[c * P:(c + 1) * P], self._g_big[c * C:(c + 1) * C], group=self.group)
        self._work.wait(); self._work = None
        small = self._small
        nat["density_partials"], nat["rgb_partials"], nat["n_partials"] = small[:enc.n_mlp], small[enc.n_mlp:], 1
        nat["scale"] = nat["scale"] * self.world
        if g16.is_cuda:
            from ._lib import call, ptr, stream
            if self._flag is None or self._flag.device != g16.device:
                self._flag = torch.zeros(16, dtype=torch.int32, device=g16.device)      # {mlp, shard} x two alternating sets, 16 bytes apart
                self._flag_step = 0
            k = self._flag_step & 1
            self._flag_step += 1
            mlp_cur, mlp_nxt = self._flag[8 * k:], self._flag[8 * (1 - k):]
            sh_cur, sh_nxt = self._flag[8 * k + 4:], self._flag[8 * (1 - k) + 4:]
            call("ngp_found_inf2", ptr(small), 1, small.numel(), None, 0, 0, ptr(mlp_cur), ptr(mlp_nxt), stream())
            call("ngp_found_inf2", ptr(self._shard16), 0, self._shard16.numel(), None, 0, 0, ptr(sh_cur), ptr(sh_nxt), stream())
            return mlp_cur, sh_cur
        bad_mlp = not bool(torch.isfinite(small).all())
        bad_sh = not bool(torch.isfinite(self._shard16.float()).all())
        one = torch.ones(1, dtype=torch.int32)
        return (one if bad_mlp else None), (one if bad_sh else None)

    def update(self, lr, step, grad_scale, found_inf, stream_handle=None):
        """Adam on this rank's pieces + the MLP blocks, then the all-gather of the f16 table (Trainer calls this instead of the
        whole-table update).  grad_scale: the factor the reduced gradients carry."""
        nat = self.model._native
        flag_mlp, flag_shard = found_inf if found_inf is not None else (None, None)
        self._adam(lr, step, grad_scale, nat, flag_mlp, flag_shard, stream_handle)
        self._stepped = True
        enc = self.model.xyz_encoder
        table = self._h_big[enc.n_mlp:]
        P, C = self.piece, self.chunk
        for c in range(self.n_chunks):
            t = table[c * C:(c + 1) * C]
            self.dist.all_gather_into_tensor(t, t[self.rank * P:(self.rank + 1) * P], group=self.group)
        self._t_end(table)
        self.model._native = None

    def _adam_kernel(self, lr, step, grad_scale, nat, flag_mlp, flag_shard, stream_handle):
        from ._lib import call, ptr, stream
        tr = self._trainer
        m = self.model
        enc, net = m.xyz_encoder, m.rgb_net
        (em, ev), (rm, rv) = tr.opt.moments("enc"), tr.opt.moments("rgb")
        b1, b2 = tr.opt.betas
        ne = enc.n_mlp
        p_enc, p_half, p_m, p_v = enc.params.data_ptr(), self._h_big.data_ptr(), em.data_ptr(), ev.data_ptr()
        sq = stream_handle if stream_handle is not None else stream()
        mlp = (p_enc, p_half, ptr(nat["density_partials"]), p_m, p_v, ne,
               net.params.data_ptr(), net._half.t.data_ptr(), ptr(nat["rgb_partials"]), rm.data_ptr(), rv.data_ptr(), net.params.numel(),
               nat["n_partials"], lr, b1, b2, tr.opt.eps, tr.opt.weight_decay, step, grad_scale, ptr(flag_mlp), ptr(flag_shard),
               tr.opt.step_state(flag_mlp), sq)
        if self.n_chunks == 1:
            lo, n = self.lo, self.hi - self.lo
            # (a rank whose shard is empty -- rank * shard_len >= n_grid: small tables, large worlds -- passes n = 0: the MLP blocks only)
            call("ngp_adam_step_field_shard", p_enc + 4 * (ne + lo), p_half + 2 * (ne + lo), ptr(self._shard16), p_m + 4 * (ne + lo), p_v + 4 * (ne + lo), n, *mlp)
        else:
            call("ngp_adam_step_field_pieces", p_enc + 4 * ne, p_half + 2 * ne, ptr(self._shard16), p_m + 4 * ne, p_v + 4 * ne, self.n_grid,
                 self.piece, self.n_chunks, self.world, self.rank, *mlp)
        enc._half.mark_fresh(enc.params); net._half.mark_fresh(net.params)

    def gather_master(self):
        """All ranks' f32 master pieces -> every rank's `xyz_encoder.params` (checkpointing / leaving the sharded mode)."""
        self._gather_shards(self.model.xyz_encoder.params.data)

    def _gather_shards(self, p):
        """p = [density MLP (n_mlp) | grid (n_grid)] f32 whose grid part is current per piece: all-gather the pieces in place."""
        enc = self.model.xyz_encoder
        buf = torch.zeros(self.padded, dtype=p.dtype, device=p.device)
        buf[:self.n_grid] = p[enc.n_mlp:]
        P, C = self.piece, self.chunk
        for c in range(self.n_chunks):
            t = buf[c * C:(c + 1) * C]
            self.dist.all_gather_into_tensor(t, t[self.rank * P:(self.rank + 1) * P].clone(), group=self.group)
        p[enc.n_mlp:] = buf[:self.n_grid]


class DirectExchange(ShardedExchange):
    """ShardedExchange with its two ring collectives replaced by point-to-point transfers (round 5): xGMI is a full mesh of
    point-to-point links, so a ring reduce-scatter / all-gather is bound by ONE link while N - 1 sit idle.  Here every rank sends
    slice q of its gradient straight to rank q and receives the N - 1 slices of its own share (N - 1 links at once), adds the N
    slices itself -- in rank order, in f32, one rounding to f16: deterministic, where the ring adds f16 in ring order -- runs Adam on
    its share and sends the updated f16 share to every peer.  The host-side mirror of ngp_stepper_tail's mode 2 (`NativeExchange(mode=
    "direct")`: ngp_comm_exchange_slices / ngp_sum_slices_f16 / ngp_comm_all_gather_direct); over torch.distributed it is made of
    isend / irecv pairs, which gloo has: tests/test_ddp_gloo.py drives it with CPU tensors at world 2-8."""

    def __init__(self, model, dist, world, rank, group=None, adam=None):
        super().__init__(model, dist, world, rank, group=group, adam=adam, n_chunks=1)
        self._stage = None

    def _p2p(self, sends, recvs):
        """All transfers of one phase: (tensor, peer) lists; posted together, waited together (an RCCL group on the device)."""
        ops = [self.dist.P2POp(self.dist.irecv, t, q, self.group) for t, q in recvs] + \
              [self.dist.P2POp(self.dist.isend, t, q, self.group) for t, q in sends]
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()

    def reduce_grid(self):
        nat = self.model._native
        if nat is None:
            return None
        enc = self.model.xyz_encoder
        if self._work is None:
            self.reduce_mlp()
        g16 = nat["grid16"]
        self._seat(g16.device)
        self._t_begin(g16)
        if g16.data_ptr() != self._g_big.data_ptr():
            self._g_big[:self.n_grid].copy_(g16)
        P, W, r = self.piece, self.world, self.rank
        if self._stage is None or self._stage.device != g16.device:
            self._stage = torch.zeros(W * P, dtype=torch.float16, device=g16.device)
        # slice q of my gradient -> rank q; rank q's slice of MY share -> stage[q]
        self._p2p([(self._g_big[q * P:(q + 1) * P], q) for q in range(W) if q != r],
                  [(self._stage[q * P:(q + 1) * P], q) for q in range(W) if q != r])
        if g16.is_cuda:
            from ._lib import call, ptr, stream
            call("ngp_sum_slices_f16", ptr(self._g_big[r * P:(r + 1) * P]), ptr(self._stage), W, r, P, ptr(self._shard16), stream())
        else:
            acc = torch.zeros(P, dtype=torch.float32)
            for q in range(W):                                  # rank order, f32, one rounding
                acc += (self._g_big[r * P:(r + 1) * P] if q == r else self._stage[q * P:(q + 1) * P]).float()
            self._shard16.copy_(acc.half())
        self._work.wait(); self._work = None
        small = self._small
        nat["density_partials"], nat["rgb_partials"], nat["n_partials"] = small[:enc.n_mlp], small[enc.n_mlp:], 1
        nat["scale"] = nat["scale"] * self.world
        if g16.is_cuda:
            from ._lib import call, ptr, stream
            if self._flag is None or self._flag.device != g16.device:
                self._flag = torch.zeros(16, dtype=torch.int32, device=g16.device)
                self._flag_step = 0
            k = self._flag_step & 1
            self._flag_step += 1
            mlp_cur, mlp_nxt = self._flag[8 * k:], self._flag[8 * (1 - k):]
            sh_cur, sh_nxt = self._flag[8 * k + 4:], self._flag[8 * (1 - k) + 4:]
            call("ngp_found_inf2", ptr(small), 1, small.numel(), None, 0, 0, ptr(mlp_cur), ptr(mlp_nxt), stream())
            call("ngp_found_inf2", ptr(self._shard16), 0, self._shard16.numel(), None, 0, 0, ptr(sh_cur), ptr(sh_nxt), stream())
            return mlp_cur, sh_cur
        bad_mlp = not bool(torch.isfinite(small).all())
        bad_sh = not bool(torch.isfinite(self._shard16.float()).all())
        one = torch.ones(1, dtype=torch.int32)
        return (one if bad_mlp else None), (one if bad_sh else None)

    def update(self, lr, step, grad_scale, found_inf, stream_handle=None):
        nat = self.model._native
        flag_mlp, flag_shard = found_inf if found_inf is not None else (None, None)
        self._adam(lr, step, grad_scale, nat, flag_mlp, flag_shard, stream_handle)
        self._stepped = True
        enc = self.model.xyz_encoder
        table = self._h_big[enc.n_mlp:]
        P, W, r = self.piece, self.world, self.rank
        mine = table[r * P:(r + 1) * P]
        self._p2p([(mine, q) for q in range(W) if q != r], [(table[q * P:(q + 1) * P], q) for q in range(W) if q != r])
        self._t_end(table)
        self.model._native = None


class NativeExchange(ShardedExchange):
    """The same exchange ENQUEUED BY THE LIBRARY (csrc/comm.hip + `ngp_stepper_tail` in csrc/stepper.hip): its own RCCL communicator
    and stream, no Python and no torch.distributed between the field backward and the gathered table.  Per step the trainer makes
    ONE call (`ngp_stepper_tail`) behind `ngp_stepper_front`; the main stream records events for the communicator's stream and
    waits once.  `mode`: "sharded" (reduce-scatter -> Adam on the rank's pieces -> all-gather of the f16 table) or "allreduce" (the
    reference's semantics literally: gradient all-reduce only, whole-table Adam on every rank).  `n_chunks` pieces of the grid
    exchange, handed over behind the `n_groups` launch groups of the table backward that complete them.  "direct" (round 5): the
    sharded schedule with point-to-point transfers over all xGMI links at once in place of the two ring collectives and the N
    slices added in rank order in f32 (see DirectExchange); one chunk.

    torch.distributed (`dist`) is used for what happens once: carrying the communicator id to the ranks, DDP's constructor broadcast
    of the parameters, and making master / moments whole again when the exchange is taken off."""

    def __init__(self, model, dist, world, rank, mode="sharded", n_chunks=1, n_groups=None, group=None):
        super().__init__(model, dist, world, rank, group=group, n_chunks=n_chunks)
        if mode not in ("sharded", "allreduce", "direct"):
            raise ValueError("mode must be 'sharded', 'allreduce' or 'direct'")
        if mode == "direct" and n_chunks != 1:
            raise ValueError("the direct exchange moves each rank's share in one piece: n_chunks must be 1")
        self.mode = mode
        self.n_groups = n_chunks if n_groups is None else n_groups
        self.comm = None
        self._cfg = None
        self._handle = None
        self._samples = []

    _MODES = {"allreduce": 0, "sharded": 1, "direct": 2}

    def _make_comm(self, dev):
        """Every rank leaves through the SAME sequence of collectives whatever happens on rank 0: if the unique id cannot be made
        there (RCCL not loadable), rank 0 still broadcasts -- an id record whose status byte says so -- and all ranks raise together
        (a rank that raised BEFORE the broadcast would leave the others waiting in it)."""
        import ctypes as C
        from ._lib import call
        ident = torch.zeros(129, dtype=torch.uint8)             # 128 bytes of id + 1 status byte (1 = valid)
        failure = None
        if self.rank == 0:
            try:
                buf = (C.c_ubyte * 128)()
                call("ngp_comm_unique_id", C.cast(buf, C.c_void_p))
                ident = torch.tensor(list(buf) + [1], dtype=torch.uint8)
            except Exception as e:          # noqa: BLE001 -- reported after the broadcast, on every rank
                failure = e
        if self.world > 1:
            carrier = ident.to(dev) if self.dist.get_backend(self.group) == "nccl" else ident
            self.dist.broadcast(carrier, 0, group=self.group)
            ident = carrier.cpu()
        if int(ident[128]) != 1:
            raise RuntimeError("rank 0 could not create the RCCL unique id%s" % (": %s" % failure if failure is not None else ""))
        raw = (C.c_ubyte * 128)(*ident[:128].tolist())
        h = C.c_void_p()
        with torch.cuda.device(dev):
            call("ngp_comm_create", C.cast(raw, C.c_void_p), self.world, self.rank, C.byref(h))
        self.comm = h

    def info(self):
        """World size, rank and RCCL version as the library's own communicator reports them (ngp_comm_info -> ncclCommCount /
        ncclCommUserRank / ncclGetVersion): the proof in an N-GPU line that RCCL saw N ranks."""
        import ctypes as C
        from ._lib import call
        if self.comm is None:
            return {"comm": "library-owned RCCL communicator (not created)"}
        w, r, v, st = C.c_int32(), C.c_int32(), C.c_int32(), C.c_void_p()
        call("ngp_comm_info", self.comm, C.byref(w), C.byref(r), C.byref(v), C.byref(st))
        return {"comm": "library-owned RCCL communicator (csrc/comm.hip)", "rccl_ranks": int(w.value), "rccl_rank": int(r.value),
                "rccl_version": int(v.value), "exchange_mode": self.mode}

    def install(self, trainer):
        from . import _lib
        dev = self.model.xyz_encoder.params.device
        if dev.type != "cuda":
            raise RuntimeError("NativeExchange needs the model on a GPU (the gloo tests drive ShardedExchange, its host-side mirror)")
        if not getattr(trainer, "native_step", False):
            # only the native stepper's tail (ngp_stepper_tail) issues this exchange's collectives: with the Python-enqueued step the
            # ranks would run whole-table Adam on their local gradients and diverge silently
            raise RuntimeError("NativeExchange needs Trainer(native_step=True) (the default): ngp_stepper_tail is what enqueues the "
                               "exchange; use ShardedExchange / GradientExchange with the Python-enqueued step")
        trainer.loss_scale = self.loss_scale
        trainer.mlp_grad_hook = trainer.grad_hook = trainer.group_hook = trainer.update_hook = None
        trainer.bwd_groups = 1
        trainer.native_exchange = self
        self._trainer = trainer
        self._seat(dev)
        enc, net = self.model.xyz_encoder, self.model.rgb_net
        self._small = torch.zeros(enc.n_mlp + net.params.numel(), dtype=torch.float32, device=dev)
        self._flag = torch.zeros(16, dtype=torch.int32, device=dev)
        state = trainer.opt.ensure_step_state()
        if self.comm is None:
            self._make_comm(dev)
        c = _lib.ExchangeConfig()
        self._stage = torch.zeros(self.world * self.piece, dtype=torch.float16, device=dev) if self.n_chunks == 1 else None
        c.mode, c.n_chunks, c.n_groups, c.piece = self._MODES[self.mode], self.n_chunks, self.n_groups, self.piece
        c.stage = self._stage.data_ptr() if self._stage is not None else None
        c.grad_padded, c.table_padded = self._g_big.data_ptr(), self._h_big.data_ptr() + 2 * enc.n_mlp
        c.shard16, c.small, c.flags, c.step_state = self._shard16.data_ptr(), self._small.data_ptr(), self._flag.data_ptr(), state.data_ptr()
        self._cfg = c
        if getattr(trainer, "_stepper", None) is not None:
            trainer._destroy_stepper()          # it holds the old gradient / working-copy pointers; the next step rebuilds and attaches
        return self

    def attach(self, handle):
        """Called by the trainer right after it (re)built its native stepper."""
        import ctypes as C
        from ._lib import call
        call("ngp_stepper_set_exchange", handle, self.comm, C.byref(self._cfg))
        self._handle = handle

    def _make_whole(self, trainer):
        """Sharded mode leaves master / moments current per piece only: gather them (torch.distributed; the device is idle first)."""
        torch.cuda.synchronize()
        if self.mode in ("sharded", "direct") and getattr(self, "_stepped", False):
            self.gather_master()
            for t in trainer.opt.moments("enc"):
                self._gather_shards(t)
            torch.cuda.synchronize()
            self._stepped = False

    def switch_mode(self, mode):
        """sharded <-> allreduce on the same communicator and buffers (bench.py measures both)."""
        import ctypes as C
        from ._lib import call
        if mode == self.mode:
            return
        if mode not in self._MODES or (mode == "direct" and self._stage is None):
            raise ValueError("cannot switch to mode %r (the direct exchange needs n_chunks == 1)" % (mode,))
        self._make_whole(self._trainer)
        self.mode = mode
        self._cfg.mode = self._MODES[mode]
        if self._handle is not None:
            call("ngp_stepper_set_exchange", self._handle, self.comm, C.byref(self._cfg))

    def uninstall(self, trainer):
        from ._lib import call
        self._make_whole(trainer)
        if getattr(trainer, "_stepper", None) is not None and self._handle is not None:
            call("ngp_stepper_set_exchange", trainer._stepper, None, None)
        self._handle = None
        trainer.native_exchange = None
        trainer.loss_scale = tcnn.LOSS_SCALE

    def close(self):
        from . import _lib
        if self.comm is not None:
            _lib.lib().ngp_comm_destroy(self.comm)
            self.comm = None

    def stepped(self):
        self._stepped = True

    # -- timing --------------------------------------------------------------------------------
    def sample_times(self):
        """(exchange_ms, exposed_ms) of the last step's tail, recorded while the trainer's stage timing is on (syncs)."""
        import ctypes as C
        from ._lib import call
        if self._handle is None:
            return
        a, b = C.c_float(), C.c_float()
        call("ngp_stepper_exchange_times", self._handle, C.byref(a), C.byref(b))
        if a.value >= 0:
            self._samples.append((a.value, b.value))

    def exchange_ms(self):
        if not self._samples:
            return None
        s, self._samples = self._samples, []
        self._last_exposed = sum(x[1] for x in s) / len(s)
        return sum(x[0] for x in s) / len(s)

    def exposed_ms(self):
        return getattr(self, "_last_exposed", None)
