"""CPU: ngp_pl_amd.ray_utils against vectors produced by the reference's own datasets/ray_utils.py
(tests/golden/make_ray_golden.py), and the differentiability the pose-optimisation path needs."""
import os

import numpy as np
import torch

from ngp_pl_amd import ray_utils as ru

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ray_golden.npz"))
t = lambda k: torch.from_numpy(G[k])


def test_ray_directions_and_rays_match_the_reference():
    dirs = ru.get_ray_directions(int(G["H"]), int(G["W"]), t("K"))
    assert torch.equal(dirs, t("dirs"))                                    # same arithmetic: bit for bit
    o1, d1 = ru.get_rays(dirs, t("c2w"))
    assert torch.equal(o1, t("o1")) and torch.allclose(d1, t("d1"), rtol=0, atol=1e-6)
    o2, d2 = ru.get_rays(dirs, t("c2w_b"))                                 # per-ray poses: elementwise form instead of a batched GEMM
    assert torch.equal(o2, t("o2")) and torch.allclose(d2, t("d2"), rtol=0, atol=1e-6)


def test_axisangle_rotation_matches_the_reference_and_is_a_rotation():
    R = ru.axisangle_to_R(t("v"))
    assert torch.allclose(R, t("R"), rtol=0, atol=1e-6)
    assert torch.allclose(ru.axisangle_to_R(t("v")[3]), t("R1"), rtol=0, atol=1e-6)
    eye = torch.eye(3).expand(7, 3, 3)
    assert torch.allclose(R @ R.transpose(1, 2), eye, atol=1e-5) and torch.allclose(torch.linalg.det(R), torch.ones(7), atol=1e-5)
    assert torch.allclose(ru.axisangle_to_R(torch.zeros(3)), torch.eye(3), atol=1e-7)          # zero vector: identity, no NaN
    assert torch.allclose(ru.perturbed_poses(t("c2w_b")[:7], t("v"), t("dT")), t("poses_opt"), rtol=0, atol=1e-6)


def test_pose_correction_is_differentiable_at_zero():
    """train.py:117-122 initialises dR = dT = 0 and learns them: gradients must be finite there and reach both."""
    dR = torch.zeros(5, 3, requires_grad=True); dT = torch.zeros(5, 3, requires_grad=True)
    poses = t("c2w_b")[:5]
    dirs = t("dirs")[:5]
    o, d = ru.get_rays(dirs, ru.perturbed_poses(poses, dR, dT))
    (o.square().sum() + (d * torch.arange(1., 4.)).sum()).backward()
    assert torch.isfinite(dR.grad).all() and torch.isfinite(dT.grad).all()
    assert bool((dR.grad.abs().sum(1) > 0).all()) and bool((dT.grad.abs().sum(1) > 0).all())
    # first order: R(dR) ~ I + [dR]x, so d(rays_d)/d(dR) = -[R0 dir]x  (checked against the analytic cross product)
    w = torch.tensor([0.3, -0.2, 0.1])
    base = (poses[0, :, :3] @ dirs[0])
    eps = 1e-3
    d_eps = ru.get_rays(dirs[:1], ru.perturbed_poses(poses[:1], (w * eps)[None], torch.zeros(1, 3)))[1][0]
    assert torch.allclose((d_eps - base) / eps, torch.linalg.cross(w, base), atol=5e-3)


def test_segment_sum_is_independent_of_the_row_order_of_rays_a():
    """RayMarcher.backward's per-ray reduction (custom_functions.py:102-112): the reference's `segment_csr(.., rays_a[:, 1])`
    is only right when the rows of rays_a are ordered by start_idx (its atomics do not guarantee it); segment_sum derives
    a sample's owner from the start indices, so any row order and any packing order give the per-ray sums at ray_idx."""
    import torch
    from ngp_pl_amd.custom_functions import segment_sum
    g = torch.Generator().manual_seed(0)
    R = 60
    counts = torch.randint(0, 9, (R,), generator=g); counts[3] = 0; counts[0] = 4
    order = torch.randperm(R, generator=g)                       # segments laid out in an order unrelated to the ray index
    start = torch.zeros(R, dtype=torch.long); pos = 0
    for r in order.tolist():
        start[r] = pos; pos += int(counts[r])
    vals = torch.randn(pos, 3, generator=g)
    want = torch.stack([vals[start[r]:start[r] + counts[r]].sum(0) for r in range(R)])
    rays_a = torch.stack([torch.arange(R), start, counts], 1)
    for rows in (rays_a, rays_a[torch.randperm(R, generator=g)]):
        assert torch.allclose(segment_sum(vals, rows), want, atol=1e-6)
    assert segment_sum(vals[:0], rays_a).abs().sum() == 0
