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
  * GradScaler's non-finite check runs on the reduced buffers (`ngp_found_inf`) and feeds the fused Adam's
    skip flag: a sum that overflowed on any rank is seen by all ranks alike (they hold the same sum), so the
    ranks stay in lock step without a second collective.

The host logic is device agnostic (the gloo test drives it with CPU tensors); kernels are used when the
buffers live on a GPU.
"""
import torch

from . import tcnn


class GradientExchange:
    def __init__(self, model, dist, world, group=None):
        self.model, self.dist, self.world, self.group = model, dist, world, group
        self.loss_scale = tcnn.LOSS_SCALE / world
        self._small = None
        self._flag = None
        self._work = None

    def install(self, trainer):
        """Hooks into Trainer.step: MLP collective behind the MLP backward, grid collective + inf check behind the
        table backward; the backward runs at loss scale 128 / world."""
        trainer.loss_scale = self.loss_scale
        trainer.mlp_grad_hook = self.reduce_mlp
        trainer.grad_hook = self.reduce_grid
        return self

    def broadcast_parameters(self):
        """DDP's constructor broadcast: every rank starts from rank 0's parameters."""
        for p in self.model.parameters():
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
        if dp.is_cuda:                          # two launches into one buffer (torch sum/sum/cat costs ~40 us of GPU time)
            from ._lib import call, ptr, stream
            call("ngp_reduce_partials", ptr(dp), n_part, n_d, ptr(small), stream())
            call("ngp_reduce_partials", ptr(rp), n_part, n_r, ptr(small[n_d:]), stream())
        else:
            small[:n_d] = dp.view(n_part, n_d).sum(0); small[n_d:] = rp.view(n_part, n_r).sum(0)
        self._work = self.dist.all_reduce(small, group=self.group, async_op=True)

    def reduce_grid(self):
        """Returns the device int32 found_inf flag (None on CPU tensors when everything is finite)."""
        nat = self.model._native
        if nat is None:
            return None
        enc = self.model.xyz_encoder
        if self._work is None:
            self.reduce_mlp()
        g16 = nat["grid16"]
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
                self._flag = torch.zeros(1, dtype=torch.int32, device=g16.device)
            call("ngp_found_inf", ptr(g16), 0, g16.numel(), ptr(self._flag), 1, stream())
            call("ngp_found_inf", ptr(small), 1, small.numel(), ptr(self._flag), 0, stream())
            return self._flag
        bad = not (bool(torch.isfinite(g16.float()).all()) and bool(torch.isfinite(small).all()))
        return torch.ones(1, dtype=torch.int32) if bad else None
