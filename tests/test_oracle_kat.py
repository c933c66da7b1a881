"""CPU: known-answer and property tests that pin the ORACLE itself (beyond the bit-exact
comparison with the reference kernels in test_oracle_vs_ref.py and the golden fixtures)."""
import math

import numpy as np
import pytest
import torch

from oracle import tcnn_oracle as T
from oracle.vren_oracle import Oracle

o = Oracle(fma=True)


def test_morton_known_answers():
    # one bit per axis; all-ones corner; the documented interleave x -> bit 0, y -> bit 1, z -> bit 2
    c = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [127, 127, 127], [3, 0, 0], [0, 5, 0]], np.int32)
    assert o.morton3D(c).tolist() == [1, 2, 4, 2097151, 0b1001, 0b10000010]
    idx = np.arange(128 ** 3, dtype=np.int32)
    assert np.array_equal(o.morton3D(o.morton3D_invert(idx)), idx)          # bijection over the grid


def test_packbits_known_answer():
    grid = np.array([0.0, 1.0, 0.5, 0.51, -1.0, 2.0, 0.5, 9.0] + [1.0] * 8, np.float32)
    out = np.zeros(2, np.uint8)
    o.packbits(grid, 0.5, out)
    assert out.tolist() == [0b10101010, 0xFF]                               # strict '>' (raymarching.cu:137)


def test_aabb_analytic_cases():
    c = np.zeros((1, 3), np.float32); h = np.full((1, 3), 0.5, np.float32)
    ro = np.array([[0, 0, -2], [0, 0, 0], [2, 2, 2], [0, 0, -2]], np.float32)
    rd = np.array([[0, 0, 1], [0, 0, 1], [0, 0, 1], [0, 0.5, 1]], np.float32)
    cnt, ht, idx = o.ray_aabb_intersect(ro, rd, c, h, 1)
    assert cnt.tolist() == [1, 1, 0, 0]
    np.testing.assert_allclose(ht[0, 0], [1.5, 2.5]); np.testing.assert_allclose(ht[1, 0], [0.0, 0.5])   # inside: t1 clamps to 0
    assert ht[2, 0].tolist() == [-1, -1] and idx[2, 0] == -1 and idx[0, 0] == 0


def test_mip_tables_through_marching():
    """The comment tables of raymarching.cu:15-18,25-28: |x| in [0,.5)->mip 0, [.5,1)->1, [1,2)->2.
    Occupy exactly one cascade and check where a ray along +x emits samples."""
    G = 128
    for mip, (lo, hi) in enumerate([(0.0, 0.5), (0.5, 1.0), (1.0, 2.0)]):
        bf = np.zeros(3 * G ** 3 // 8, np.uint8)
        bf[mip * G ** 3 // 8:(mip + 1) * G ** 3 // 8] = 255
        ro = np.array([[0.001, 0.002, 0.003]], np.float32); rd = np.array([[1.0, 0.0, 0.0]], np.float32)
        ht = np.array([[0.01, 1.99]], np.float32)
        _, xyzs, _, deltas, ts, cnt = o.raymarching_train(ro, rd, ht, bf, 3, 2.0, 0.0, np.zeros(1, np.float32), G, 1024)
        assert cnt[0] > 0
        assert xyzs[:, 0].min() >= lo - 2e-3 and xyzs[:, 0].max() < hi + 2e-3
        np.testing.assert_allclose(deltas, math.sqrt(3) / 1024, rtol=1e-6)   # exp_step_factor = 0 -> constant dt


def test_marching_edge_cases():
    bf = np.full(128 ** 3 // 8, 255, np.uint8)
    ro = np.array([[0, 0, -2], [0, 0, -2]], np.float32); rd = np.array([[0, 0, 1], [0, 0, 1]], np.float32)
    ht = np.array([[1.5, 2.5], [-1, -1]], np.float32)                        # second ray misses the box
    rays_a, xyzs, dirs, deltas, ts, cnt = o.raymarching_train(ro, rd, ht, bf, 1, 0.5, 0.0, np.zeros(2, np.float32), 128, 1024)
    assert rays_a.tolist() == [[0, 0, 592], [1, 592, 0]]                      # ceil(1.0 / (sqrt3/1024)) samples; a miss still gets a row
    assert cnt.tolist() == [592, 2] and np.all(np.diff(ts) > 0)
    # max_samples sets the step: dt = sqrt3/100 -> ceil(1.0/dt) = 58 samples (raymarching.cu:11-13)
    ra2, *_ = o.raymarching_train(ro[:1], rd[:1], ht[:1], bf, 1, 0.5, 0.0, np.zeros(1, np.float32), 128, 100)
    assert ra2[0, 2] == 58
    # test-time marching resumes where it stopped
    h = ht[:1].copy()
    a = o.raymarching_test(ro[:1], rd[:1], h, np.array([0]), bf, 1, 0.5, 0.0, 128, 1024, 4)
    b = o.raymarching_test(ro[:1], rd[:1], h, np.array([0]), bf, 1, 0.5, 0.0, 128, 1024, 4)
    assert a[4][0] == 4 and b[4][0] == 4 and np.isclose(b[3][0, 0], a[3][0, 3] + a[2][0, 3])


def test_composite_closed_form_and_early_stop():
    n = 50
    deltas = np.full(n, 0.01, np.float32); ts = np.arange(n, dtype=np.float32) * 0.01
    rays_a = np.array([[0, 0, n]], np.int64)
    sig = np.full(n, 3.0, np.float32); rgbs = np.tile(np.array([[0.2, 0.5, 0.9]], np.float32), (n, 1))
    total, op, depth, rgb, ws = o.composite_train_fw(sig, rgbs, deltas, ts, rays_a, 1e-4)
    np.testing.assert_allclose(op[0], 1 - math.exp(-3.0 * 0.5), rtol=1e-5)     # constant sigma: 1 - exp(-sigma * sum(delta))
    np.testing.assert_allclose(rgb[0], op[0] * rgbs[0], rtol=1e-5)
    assert total[0] == n
    sig[:] = 2000.0                                                             # opaque after the first sample
    total, op, depth, rgb, ws = o.composite_train_fw(sig, rgbs, deltas, ts, rays_a, 1e-4)
    assert total[0] == 0 and ws[0] > 0.99 and np.all(ws[1:] == 0)               # the stopping sample is composited but not counted (:41-44)


def composite_torch(sig, rgbs, deltas, ts, T_thr):
    """Differentiable fp64 re-derivation (no early stop), for gradient checking."""
    a = 1 - torch.exp(-sig * deltas)
    T = torch.cumprod(torch.cat([torch.ones(1, dtype=a.dtype), 1 - a[:-1]]), 0)
    w = a * T
    return w.sum(), (w * ts).sum(), (w[:, None] * rgbs).sum(0), w


def test_composite_backward_matches_autograd():
    g = np.random.RandomState(0)
    n = 40
    sig = (g.rand(n) * 20).astype(np.float32); rgbs = g.rand(n, 3).astype(np.float32)
    deltas = np.full(n, 0.004, np.float32); ts = np.cumsum(deltas).astype(np.float32)
    rays_a = np.array([[0, 0, n]], np.int64)
    total, op, depth, rgb, ws = o.composite_train_fw(sig, rgbs, deltas, ts, rays_a, 0.0)   # threshold 0: never stops
    dO, dD, dC, dW = g.randn(1).astype(np.float32), g.randn(1).astype(np.float32), g.randn(1, 3).astype(np.float32), g.randn(n).astype(np.float32)
    dsig, drgbs = o.composite_train_bw(dO, dD, dC, dW, sig, rgbs, ws, deltas, ts, rays_a, op, depth, rgb, 0.0)
    s = torch.tensor(sig, dtype=torch.float64, requires_grad=True); c = torch.tensor(rgbs, dtype=torch.float64, requires_grad=True)
    O, D, C, W = composite_torch(s, c, torch.tensor(deltas, dtype=torch.float64), torch.tensor(ts, dtype=torch.float64), 0.0)
    (O * float(dO[0]) + D * float(dD[0]) + (C * torch.tensor(dC[0], dtype=torch.float64)).sum() + (W * torch.tensor(dW, dtype=torch.float64)).sum()).backward()
    np.testing.assert_allclose(dsig, s.grad.numpy(), rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(drgbs, c.grad.numpy(), rtol=2e-4, atol=2e-6)


def test_distortion_loss_matches_quadratic_definition():
    g = np.random.RandomState(1)
    n = 30
    ws = (g.rand(n) * 0.05).astype(np.float32); deltas = np.full(n, 0.01, np.float32); ts = (np.arange(n) * 0.01 + 0.005).astype(np.float32)
    rays_a = np.array([[0, 0, n]], np.int64)
    loss, wi, wti = o.distortion_loss_fw(ws, deltas, ts, rays_a)
    # Mip-NeRF 360 eq. 15: sum_ij w_i w_j |t_i - t_j| + 1/3 sum_i w_i^2 delta_i
    ref = (ws[:, None] * ws[None] * np.abs(ts[:, None] - ts[None])).sum() + (ws ** 2 * deltas).sum() / 3
    np.testing.assert_allclose(loss[0], ref, rtol=1e-4)
    w = torch.tensor(ws, dtype=torch.float64, requires_grad=True); t = torch.tensor(ts, dtype=torch.float64)
    ((w[:, None] * w[None] * (t[:, None] - t[None]).abs()).sum() + (w ** 2 * torch.tensor(deltas, dtype=torch.float64)).sum() / 3).backward()
    got = o.distortion_loss_bw(np.ones(1, np.float32), wi, wti, ws, deltas, ts, rays_a)
    np.testing.assert_allclose(got, w.grad.numpy(), rtol=1e-3, atol=1e-6)


# ---- tiny-cuda-nn restatement --------------------------------------------------------------------
def test_hash_grid_properties():
    meta = T.GridMeta(16, 2, 19, 16, math.exp(math.log(2048 * 0.5 / 16) / 15))
    assert meta.resolution[:5] == [16, 22, 28, 37, 49] and meta.offset[1] == 4096
    assert [meta.level_is_hashed(l) for l in range(16)] == [False] * 6 + [True] * 10
    g = torch.Generator().manual_seed(0)
    table = torch.rand(meta.total, 2, generator=g)
    # constant table -> constant features (trilinear weights sum to 1)
    f = T.hash_encode(torch.rand(100, 3, generator=g), torch.full((meta.total, 2), 0.37), meta)
    np.testing.assert_allclose(f.numpy(), 0.37, rtol=1e-5)
    # at a grid vertex of dense level 0 the feature IS the table entry x + y*res + z*res^2
    ix, iy, iz = 3, 7, 11
    x = (torch.tensor([[ix, iy, iz]], dtype=torch.float32) - 0.5 + 1e-4) / meta.scale[0]     # pos = x*scale + 0.5
    f = T.hash_encode(x, table, meta)
    np.testing.assert_allclose(f[0, :2].numpy(), table[ix + iy * 16 + iz * 256].numpy(), atol=1e-3)
    # hashed level: index = (x ^ y*2654435761 ^ z*805459861) mod 2^19 in uint32 arithmetic
    idx = T._corner_indices(meta, 10, torch.tensor([[5, 9, 200]]))[0]
    assert int(idx[0]) == ((5 * 1) ^ ((9 * 2654435761) & 0xFFFFFFFF) ^ ((200 * 805459861) & 0xFFFFFFFF)) % (1 << 19)
    # gradient: d feat / d table entries are the trilinear weights (sum 1 per level and feature)
    t2 = table.clone().requires_grad_(True)
    T.hash_encode(torch.rand(1, 3, generator=g), t2, meta).sum().backward()
    np.testing.assert_allclose(t2.grad.sum().item(), 32.0, rtol=1e-5)


def test_sh4_is_orthonormal():
    g = torch.Generator().manual_seed(0)
    d = torch.randn(200000, 3, generator=g, dtype=torch.float64); d /= d.norm(dim=1, keepdim=True)
    Y = T.sh4(d)
    gram = (Y.T @ Y) / d.shape[0] * 4 * math.pi                                  # Monte-Carlo integral over the sphere
    np.testing.assert_allclose(gram.numpy(), np.eye(16), atol=0.03)
    assert abs(T.sh4(torch.tensor([[0.0, 0.0, 1.0]]))[0, 6].item() - (0.94617469575756 - 0.31539156525252)) < 1e-6


def test_mlp_layout():
    """(out,in) row-major, layer after layer, output padded to 16 rows, no bias."""
    g = torch.Generator().manual_seed(0)
    p = torch.randn(64 * 32 + 64 * 64 + 16 * 64, generator=g)
    x = torch.randn(5, 32, generator=g)
    w0, w1, w2 = p[:2048].view(64, 32), p[2048:2048 + 4096].view(64, 64), p[6144:].view(16, 64)
    want = torch.sigmoid((torch.relu(torch.relu(x @ w0.T) @ w1.T) @ w2.T)[:, :3])
    np.testing.assert_allclose(T.mlp(x, p, 32, 2, 3, "Sigmoid").numpy(), want.numpy(), rtol=1e-6)
    f = T.Field()
    assert f.density_w.numel() == 3072 and f.rgb_w.numel() == 7168


# ---- known answers from sources independent of the restatement (tests/kat_independent.py) ---------------------
def test_sh4_against_scipy():
    """oracle/tcnn_oracle.py:sh4 (the polynomial forms recalled from spherical_harmonics.h) against scipy.special's spherical harmonics
    on 10 000 directions incl. the poles and axes: a8's constants and signs are pinned to an independent implementation."""
    import kat_independent as K
    d = K.unit_directions(10000, seed=3)
    want = K.sh4_scipy(d)
    got = T.sh4(torch.from_numpy(d)).numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
    assert np.abs(want).max() > 0.7 and (np.abs(want).max(0) > 0.28).all()      # every basis function is exercised


def test_level_tables_are_the_literal_known_answers():
    """Resolutions / offsets / parameter counts of the hash grid for scale 0.5 and 16 in both table modes, as literal numbers: the oracle's
    GridMeta and the product's make_grid_meta (library ngp_grid_meta_init for float32, Python for exact) against them."""
    import kat_independent as K
    from ngp_pl_amd import tcnn
    for (scale, mode), want in K.LEVELS.items():
        b = K.per_level_scale(scale)
        m = T.GridMeta(16, 2, 19, 16, b, exact=(mode == "exact"))
        assert m.resolution == want["resolution"] and m.offset == want["offset"] and 2 * m.total == want["n_params"], (scale, mode)
        pm = tcnn.make_grid_meta({"otype": "Grid", "type": "Hash", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19,
                                  "base_resolution": 16, "per_level_scale": b, "interpolation": "Linear"}, level_table=mode)
        assert [pm.resolution[i] for i in range(16)] == want["resolution"] and [pm.offset[i] for i in range(17)] == want["offset"], (scale, mode)
        assert all(abs(pm.scale[i] - m.scale[i]) <= 1e-6 * m.scale[i] for i in range(16))
    # the parameter vector of `xyz_encoder` = 3072 MLP weights + the table (11 423 136 is the exact table's)
    assert 3072 + K.LEVELS[(0.5, "exact")]["n_params"] == 11423136 and 3072 + K.LEVELS[(0.5, "float32")]["n_params"] == 11448112


def test_hash_indices_are_the_hand_computed_known_answers():
    import kat_independent as K
    meta = T.GridMeta(16, 2, 19, 16, K.per_level_scale(0.5))
    for (x, y, z), want in K.HASH_KAT:
        got = int(T._corner_indices(meta, 15, torch.tensor([[x, y, z]]))[0][0])
        assert got == want, ((x, y, z), got, want)
    # all 8 corners of one cell: corner c adds (c & 1, (c >> 1) & 1, c >> 2)
    idx = [int(t[0]) for t in T._corner_indices(meta, 15, torch.tensor([[0, 0, 0]]))]
    assert idx == [0, 1, 489905, 489905 ^ 1, 153493, 153493 ^ 1, 489905 ^ 153493, 489905 ^ 153493 ^ 1]


@pytest.mark.parametrize("n_hidden", [1, 2])
def test_mlp_with_hand_computable_weights(n_hidden):
    """The restatement of FullyFusedMLP on the weights of tests/test_kat_gpu.py::test_fully_fused_mlp_with_hand_computable_weights
    (h[i] = relu(x[i]), h[32 + i] = relu(-x[i]); optional reversal layer; y[k] = 2 x[k] + 0.5 max(x[k + 16], 0)): the answer is known
    without evaluating any matrix product, so the (out, in) row-major / layer-after-layer layout of oracle and product are pinned to
    the same independent statement."""
    W0 = torch.zeros(64, 32); W1 = torch.zeros(64, 64); Wo = torch.zeros(16, 64)
    for i in range(32):
        W0[i, i] = 1.0; W0[32 + i, i] = -1.0
    pos = (lambda j: 63 - j) if n_hidden == 2 else (lambda j: j)
    for j in range(64):
        W1[j, 63 - j] = 1.0
    for k in range(16):
        Wo[k, pos(k)] = 2.0; Wo[k, pos(32 + k)] = -2.0; Wo[k, pos(16 + k)] = 0.5
    blob = torch.cat([W0.reshape(-1)] + ([W1.reshape(-1)] if n_hidden == 2 else []) + [Wo.reshape(-1)])
    g = torch.Generator().manual_seed(1)
    x = torch.randint(-32, 33, (257, 32), generator=g).float() / 8.0
    want = 2.0 * x[:, :16] + 0.5 * x[:, 16:32].clamp(min=0)
    assert torch.equal(T.mlp(x, blob, 32, n_hidden, 16, quantize=True), want)
    assert torch.equal(T.mlp(x, blob, 32, n_hidden, 16, acc16=True), want)
