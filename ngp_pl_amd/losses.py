"""Training losses of the hot path, API-compatible with the reference's `losses.py`
(NeRFLoss at losses.py:40-60, DistortionLoss at losses.py:6-37), on top of `ngp_pl_amd.vren`.

`NeRFLoss(lambda_opacity, lambda_distortion)(results, target)` returns the same dictionary of
per-element terms; the native step computes the default recipe (rgb + opacity terms, mean-reduced,
with analytic backward seeds) inside the compositing kernel (`ngp_composite_train_fw_loss`;
`ngp_nerf_loss` is the stand-alone form).
"""
import torch
from torch import nn

from . import vren

_EPS = 1e-10


def squared_error(pred_rgb, gt_rgb):
    """(R,3) photometric term."""
    diff = pred_rgb - gt_rgb
    return diff * diff


def opacity_entropy(opacity, weight):
    """-o log o, pushes a ray's opacity towards 0 or 1 (floaters)."""
    o = opacity + _EPS
    return -weight * o * torch.log(o)


def mean_loss_and_seeds(rgb, opacity, gt_rgb, bg=None, lambda_opacity=1e-3, grad_scale=1.0):
    """What the native step's compositing kernel produces for the default recipe, written with torch (fp32, any
    device): the scalar the trainer minimises, `sum_k mean(term_k)` over the rgb and opacity terms (train.py:173) with
    the background blended in first (rendering.py:153-161, `rgb + bg * (1 - opacity)`), the sum of squared errors
    (PSNR bookkeeping, train.py:177-180) and the analytic backward seeds dL/d(rgb) (R,3), dL/d(opacity) (R), both
    multiplied by `grad_scale`.  Used as the plain-torch statement of `ngp_composite_train_fw_loss`'s loss half."""
    n_rays = rgb.shape[0]
    transparency = (1.0 - opacity).unsqueeze(1)
    blended = rgb if bg is None else rgb + bg.view(1, 3) * transparency
    residual = blended - gt_rgb
    o = opacity + _EPS
    log_o = torch.log(o)
    sq_err = (residual * residual).sum()
    loss = sq_err / (3 * n_rays) + (-lambda_opacity * o * log_o).sum() / n_rays
    d_rgb = residual * (2.0 / (3 * n_rays))
    d_opacity = -lambda_opacity * (log_o + 1.0) / n_rays
    if bg is not None:
        d_opacity = d_opacity - (d_rgb * bg.view(1, 3)).sum(1)
    return loss, sq_err, d_rgb * grad_scale, d_opacity * grad_scale


class DistortionLoss(torch.autograd.Function):
    """Mip-NeRF 360's distortion regulariser evaluated with DVGO-v2's prefix sums.

    Arguments: ws (S) sample weights, deltas (S) interval lengths, ts (S) interval mid-points,
    rays_a (R,3) rows of [ray_idx, start_idx, N_samples].  Output: (R) loss per ray.  Only `ws`
    receives a gradient."""

    @staticmethod
    def forward(ctx, ws, deltas, ts, rays_a):
        ws = ws.contiguous()
        per_ray, w_cum, wt_cum = vren.distortion_loss_fw(ws, deltas, ts, rays_a)
        ctx.save_for_backward(w_cum, wt_cum, ws, deltas, ts, rays_a)
        return per_ray

    @staticmethod
    def backward(ctx, grad_per_ray):
        w_cum, wt_cum, ws, deltas, ts, rays_a = ctx.saved_tensors
        g = vren.distortion_loss_bw(grad_per_ray.contiguous(), w_cum, wt_cum, ws, deltas, ts, rays_a)
        return g, None, None, None


class _LossTerms(torch.autograd.Function):
    """rgb (R,3), opacity (R), gt (R,3) -> (rgb - gt)^2 (R,3), lambda * -(o + eps) log(o + eps) (R): the two unreduced terms of
    losses.py:47-56 in one launch forward and one backward (ngp_nerf_loss_terms_fw/_bw)."""

    @staticmethod
    def forward(ctx, rgb, opacity, gt, lambda_opacity):
        from ._lib import call, device_guard, ptr, stream
        rgb = rgb.float().contiguous(); opacity = opacity.float().contiguous(); gt = gt.float().contiguous()
        n = rgb.shape[0]
        sq = torch.empty_like(rgb); ent = torch.empty_like(opacity)
        with device_guard(rgb.device):
            call("ngp_nerf_loss_terms_fw", ptr(rgb), ptr(opacity), ptr(gt), float(lambda_opacity), n, ptr(sq), ptr(ent), stream())
        ctx.save_for_backward(rgb, opacity, gt)
        ctx.lambda_opacity = float(lambda_opacity)
        return sq, ent

    @staticmethod
    def backward(ctx, g_sq, g_ent):
        from ._lib import call, device_guard, ptr, stream
        rgb, opacity, gt = ctx.saved_tensors
        n = rgb.shape[0]
        # `.mean().backward()` hands back expanded (stride-0) views of one scalar: pass the scalar, do not materialise it
        def seed(g):
            if g.dtype == torch.float32 and g.numel() > 0 and all(st == 0 for st in g.stride()):
                return g, 1
            return g.float().contiguous(), 0
        (g_sq, sq_scalar), (g_ent, ent_scalar) = seed(g_sq), seed(g_ent)
        g_rgb = torch.empty_like(rgb); g_op = torch.empty_like(opacity)
        with device_guard(rgb.device):
            call("ngp_nerf_loss_terms_bw", ptr(g_sq), sq_scalar, ptr(g_ent), ent_scalar, ptr(rgb), ptr(opacity), ptr(gt), ctx.lambda_opacity, n,
                 ptr(g_rgb), ptr(g_op), stream())
        return g_rgb, g_op, None, None


class NeRFLoss(nn.Module):
    def __init__(self, lambda_opacity=1e-3, lambda_distortion=1e-3):
        super().__init__()
        self.lambda_opacity, self.lambda_distortion = lambda_opacity, lambda_distortion

    def forward(self, results, target, **kwargs):
        rgb, opacity, gt = results["rgb"], results["opacity"], target["rgb"]
        if rgb.is_cuda and rgb.dim() == 2 and rgb.shape[1] == 3 and gt.shape == rgb.shape and not gt.requires_grad:
            sq, ent = _LossTerms.apply(rgb, opacity, gt, self.lambda_opacity)
            terms = {"rgb": sq, "opacity": ent}
        else:
            terms = {
                "rgb": squared_error(rgb, gt),
                "opacity": opacity_entropy(opacity, self.lambda_opacity),
            }
        if self.lambda_distortion > 0:
            per_ray = DistortionLoss.apply(results["ws"], results["deltas"], results["ts"], results["rays_a"])
            terms["distortion"] = self.lambda_distortion * per_ray
        return terms
