"""Ray construction helpers with the names the reference's training loop uses
(/root/reference/datasets/ray_utils.py: get_ray_directions :11-47, get_rays :50-74, axisangle_to_R :77-106; used at
train.py:78-91).  Pure torch, differentiable where the reference's are (pose optimisation, `--optimize_ext`:
train.py:86-89 rotates the camera frame by axisangle_to_R(dR) and shifts it by dT before forming the rays, and the
gradients that reach them come out of RayMarcher.backward).  Checked on the CPU against vectors produced by the
reference's own functions (tests/golden/make_ray_golden.py).
"""
import torch

from .synthetic import get_ray_directions, get_rays   # noqa: F401  (same contract as ray_utils.py:11-74)


def axisangle_to_R(v):
    """Rotation matrix of an axis-angle vector, (3) -> (3,3) or (B,3) -> (B,3,3) (ray_utils.py:77-106): Rodrigues'
    formula R = I + sin(a)/a [v]x + (1 - cos a)/a^2 [v]x^2 with a = |v| + 1e-7, so that v = 0 gives I with finite
    gradients."""
    single = v.ndim == 1
    w = v.reshape(-1, 3)
    x, y, z = w[:, 0], w[:, 1], w[:, 2]
    o = torch.zeros_like(x)
    cross = torch.stack([o, -z, y, z, o, -x, -y, x, o], 1).reshape(-1, 3, 3)       # [v]x
    angle = (w.norm(dim=1) + 1e-7).reshape(-1, 1, 1)
    eye = torch.eye(3, dtype=w.dtype, device=w.device)
    R = eye + (torch.sin(angle) / angle) * cross + ((1.0 - torch.cos(angle)) / (angle * angle)) * (cross @ cross)
    return R[0] if single else R


def perturbed_poses(poses, dR, dT):
    """train.py:86-89: c2w (B,3,4) with the learnable extrinsic correction applied -- rotation axisangle_to_R(dR) (B,3)
    on the camera frame, translation dT (B,3) on the camera centre.  Returns a new tensor (the reference edits a
    gathered copy in place)."""
    R = axisangle_to_R(dR)
    return torch.cat([R @ poses[..., :3], (poses[..., 3] + dT).unsqueeze(-1)], -1)
