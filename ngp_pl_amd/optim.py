"""Fused Adam for the NGP field: apex FusedAdam's update (the reference's optimizer,
/root/reference/train.py:131: lr 1e-2, eps 1e-15, adam_w_mode, bias correction) applied by
ngp_adam_step directly to the native gradient buffers the fused backward leaves behind
(packed-f16 grid gradient, f32 per-workgroup MLP partials): unscale + Adam + f32->f16 parameter
cast + gradient zeroing in ONE pass over the 11.4 M parameters and ONE launch for the grid table and
both MLP blocks (ngp_adam_step_field).
"""
import contextlib
import math

import torch

from . import tcnn
from ._lib import call, ptr, stream


class FusedAdam:
    def __init__(self, model, lr=1e-2, betas=(0.9, 0.999), eps=1e-15, weight_decay=0.0):
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.param_groups = [{"lr": lr}]
        self.t = 0
        enc, net = model.xyz_encoder, model.rgb_net
        self.state = {}
        for name, p in (("enc", enc.params), ("rgb", net.params)):
            self.state[name] = (torch.zeros_like(p.data), torch.zeros_like(p.data))
        # make sure the f16 working copies exist; from now on this optimizer keeps them fresh
        enc._half.get(enc.params); net._half.get(net.params)
        model.native_grads = True

    def zero_grad(self):
        pass   # ngp_adam_step zeroes what it consumes; the sliced grid backward overwrites

    @torch.no_grad()
    def step(self, grad_scale=1.0, found_inf=None, stream_handle=None):
        """grad_scale: extra factor the caller put on the loss (GradScaler); the kernels' own loss
        scale is taken from the native record.  found_inf: device int32 flag (non-zero: skip)."""
        model = self.model
        nat = model._native
        if nat is None:
            raise RuntimeError("FusedAdam.step() needs a backward of NGP's fused field (model.native_grads=True)")
        enc, net = model.xyz_encoder, model.rgb_net
        self.t += 1
        lr = self.param_groups[0]["lr"]
        b1, b2 = self.betas
        total_scale = nat["scale"] * grad_scale
        sq = stream_handle if stream_handle is not None else stream()
        m, v = self.state["enc"]
        rm, rv = self.state["rgb"]
        ne = enc.n_mlp
        # raw pointers by arithmetic (a tensor slice costs ~4 us of host time, this call passes 6 of them)
        p_enc, p_half, p_m, p_v = enc.params.data_ptr(), enc._half.t.data_ptr(), m.data_ptr(), v.data_ptr()
        guard = torch.cuda.device(enc.params.device) if stream_handle is None else contextlib.nullcontext()     # a raw stream handle names its device
        with guard:
            call("ngp_adam_step_field", p_enc + 4 * ne, p_half + 2 * ne, ptr(nat["grid16"]), p_m + 4 * ne, p_v + 4 * ne, enc.n_grid,
                 p_enc, p_half, ptr(nat["density_partials"]), p_m, p_v, ne,
                 net.params.data_ptr(), net._half.t.data_ptr(), ptr(nat["rgb_partials"]), rm.data_ptr(), rv.data_ptr(), net.params.numel(),
                 nat["n_partials"], lr, b1, b2, self.eps, self.weight_decay, self.t, total_scale, 0, ptr(found_inf), sq)      # 0: every table backward of this package overwrites the gradient
        model._native = None


def cosine_lr(base_lr, epoch, num_epochs, eta_min_ratio=1 / 30):
    """torch CosineAnnealingLR(opt, num_epochs, lr/30) stepped per epoch (train.py:135-137)."""
    eta_min = base_lr * eta_min_ratio
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch / num_epochs)) / 2
