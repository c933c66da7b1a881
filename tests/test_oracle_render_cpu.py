"""CPU: the oracle's restatement of render() (oracle/render_oracle.py, rendering.py:46-163) is
self-consistent -- the iterative test-time loop and the packed training path composite the same
samples when the marcher's jitter is zero -- and its occupancy-grid merge reproduces hand-computed
cases (networks.py:256-268)."""
import numpy as np
import pytest
import torch

from oracle import render_oracle as RO
from oracle import tcnn_oracle as T
from oracle.vren_oracle import Oracle
from tests.helpers import make_rays


def blob_bitfield(vr, G=128, radius=0.3):
    """cells within `radius` of the origin are occupied (Morton order, one cascade)."""
    c = np.stack(np.meshgrid(np.arange(G), np.arange(G), np.arange(G), indexing="ij"), -1).reshape(-1, 3).astype(np.int32)
    centre = (c.astype(np.float32) + 0.5) / G - 0.5
    occ = (np.linalg.norm(centre, axis=1) < radius).astype(np.float32)
    grid = np.zeros(G ** 3, np.float32)
    grid[vr.morton3D(c).astype(np.int64)] = occ
    bits = np.zeros(G ** 3 // 8, np.uint8)
    vr.packbits(grid, 0.5, bits)
    return bits


@pytest.fixture(scope="module")
def setup():
    vr = Oracle()
    f = T.Field(scale=0.5, seed=3)
    g = torch.Generator().manual_seed(4)
    f.table = ((torch.rand(f.meta.total, 2, generator=g) * 2 - 1) * 0.5).half().float()     # densities of order 1..100: rays saturate
    f.density_w = (f.density_w * 2.0).half().float()
    return vr, f, blob_bitfield(vr)


def test_test_loop_equals_train_path_without_jitter(setup):
    vr, f, bits = setup
    ro, rd = make_rays(300, seed=5, W=200)
    op, depth, rgb, total, iters = RO.render_rays_test(vr, f, ro, rd, bits)
    tr = RO.render_rays_train(vr, f, ro, rd, bits, noise=np.zeros(300, np.float32))
    assert iters > 3 and total > 0
    # same lattice of samples, same front-to-back arithmetic up to where T = 1 - opacity is re-read between chunks
    np.testing.assert_allclose(op, tr["opacity"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(rgb, tr["rgb"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(depth, tr["depth"], rtol=0, atol=2e-5)
    hit = tr["rays_a"][:, 2] > 0
    assert hit.any() and (~hit).any()
    miss_idx = tr["rays_a"][~hit, 0]
    assert np.all(op[miss_idx] == 0) and np.all(rgb[miss_idx] == 1.0)               # white background (rendering.py:111-112)
    # the loop never marches more samples than the training path emits, and stops early on saturated rays
    assert total <= tr["rm_samples"] + 64 * 300


def test_test_loop_on_rays_that_miss_everything(setup):
    vr, f, bits = setup
    ro = np.full((50, 3), 5.0, np.float32)
    rd = np.tile(np.array([[0.3, 0.5, 0.8]], np.float32), (50, 1))
    op, depth, rgb, total, iters = RO.render_rays_test(vr, f, ro, rd, bits)
    assert total == 0 and iters == 1 and np.all(op == 0) and np.all(rgb == 1.0)


def test_occupancy_merge_known_answers():
    vr = Oracle()
    grid = np.zeros(64, np.float32)
    grid[0] = -1.0; grid[1] = 10.0; grid[2] = 0.0; grid[3] = 4.0
    new, bits, thr = RO.update_density_grid(vr, grid, cells=[0, 1, 2, 2, 5], sigmas=[100.0, 1.0, 3.0, 7.0, 2.0], density_threshold=5.9)
    assert new[0] == -1.0                                   # invisible cells stay -1 (networks.py:259)
    assert new[1] == np.float32(10.0) * np.float32(0.95)    # decayed old value beats the new sigma
    assert new[2] == 7.0                                    # duplicate cell: the last write wins
    assert new[3] == np.float32(4.0) * np.float32(0.95) and new[5] == 2.0 and new[4] == 0.0
    mean = np.mean([9.5, 7.0, 3.8, 2.0])
    assert abs(thr - min(mean, 5.9)) < 1e-6
    occupied = [i for i in range(64) if bits[i // 8] >> (i % 8) & 1]
    assert occupied == [1, 2]                               # > 5.575: 9.5 and 7.0


def test_loss_seeds_statement_matches_autograd_of_the_reference_shaped_loss():
    """losses.mean_loss_and_seeds (the torch statement of the fused kernel's loss half) == autograd through
    NeRFLoss + background blend + mean, the way train.py:159-176 forms the scalar."""
    import torch
    from ngp_pl_amd.losses import NeRFLoss, mean_loss_and_seeds
    g = torch.Generator().manual_seed(3)
    n = 257
    rgb = torch.rand(n, 3, generator=g, dtype=torch.float64).requires_grad_(True)
    opacity = torch.rand(n, generator=g, dtype=torch.float64).requires_grad_(True)
    gt = torch.rand(n, 3, generator=g, dtype=torch.float64)
    for bg in (None, torch.ones(3, dtype=torch.float64), torch.tensor([0.2, 0.5, 0.9], dtype=torch.float64)):
        blended = rgb if bg is None else rgb + bg.view(1, 3) * (1 - opacity).unsqueeze(1)       # rendering.py:153-161
        terms = NeRFLoss(lambda_opacity=1e-3, lambda_distortion=0)({"rgb": blended, "opacity": opacity}, {"rgb": gt})
        loss = sum(t.mean() for t in terms.values())                                             # train.py:173
        want = torch.autograd.grad(loss * 128.0, (rgb, opacity))
        got = mean_loss_and_seeds(rgb.detach(), opacity.detach(), gt, bg, 1e-3, 128.0)
        assert torch.allclose(got[0], loss.detach(), rtol=1e-12)
        assert torch.allclose(got[1], ((blended - gt) ** 2).sum().detach(), rtol=1e-12)
        assert torch.allclose(got[2], want[0], rtol=1e-10, atol=1e-15) and torch.allclose(got[3], want[1], rtol=1e-10, atol=1e-15)
