"""Loss glue of the reference (/root/reference/losses.py:6-60) on `ngp_pl_amd.vren`."""
import torch
from torch import nn

from . import vren


class DistortionLoss(torch.autograd.Function):
    """Mip-NeRF 360 distortion loss in its DVGO-v2 prefix-sum form (losses.py:6-37).
    ws, deltas, ts (S), rays_a (R,3) -> loss (R)."""

    @staticmethod
    def forward(ctx, ws, deltas, ts, rays_a):
        loss, ws_incl, wts_incl = vren.distortion_loss_fw(ws.contiguous(), deltas, ts, rays_a)
        ctx.save_for_backward(ws_incl, wts_incl, ws, deltas, ts, rays_a)
        return loss

    @staticmethod
    def backward(ctx, dL_dloss):
        ws_incl, wts_incl, ws, deltas, ts, rays_a = ctx.saved_tensors
        return vren.distortion_loss_bw(dL_dloss.contiguous(), ws_incl, wts_incl, ws.contiguous(), deltas, ts, rays_a), None, None, None


class NeRFLoss(nn.Module):
    """rgb MSE + opacity entropy (+ distortion), per-element terms in a dict (losses.py:40-60)."""

    def __init__(self, lambda_opacity=1e-3, lambda_distortion=1e-3):
        super().__init__()
        self.lambda_opacity = lambda_opacity
        self.lambda_distortion = lambda_distortion

    def forward(self, results, target, **kwargs):
        d = {"rgb": (results["rgb"] - target["rgb"]) ** 2}
        o = results["opacity"] + 1e-10
        d["opacity"] = self.lambda_opacity * (-o * torch.log(o))
        if self.lambda_distortion > 0:
            d["distortion"] = self.lambda_distortion * DistortionLoss.apply(
                results["ws"], results["deltas"], results["ts"], results["rays_a"])
        return d
