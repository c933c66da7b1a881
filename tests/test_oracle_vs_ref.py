"""Pins oracle/ngp_oracle.c against the reference's own kernels compiled for the CPU.

oracle/_ref/libvren_ref_{nofma,fma}.so are /root/reference/models/csrc/*.cu built by
oracle/build_ref.sh.  With matching contraction settings every output must agree BIT FOR BIT
(integer outputs and floats alike): same per-ray sample counts, same packed (t, dt, xyz), same
composite sums and gradients.  Skipped only if the reference build is absent.
"""
import numpy as np
import pytest

from ngp_pl_amd import synthetic as syn
from oracle.vren_oracle import Oracle, Reference
from tests.helpers import aabb_hits, make_rays

pytestmark = pytest.mark.skipif(not (Reference.available(True) and Reference.available(False)),
                                reason="oracle/_ref not built (needs /root/reference)")


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def assert_same(a, b, what):
    assert a.shape == b.shape, what
    assert np.array_equal(bits(a), bits(b)), "%s differs: %d of %d elements" % (what, (bits(a) != bits(b)).sum(), a.size)


def assert_close_or_same(o, a, b, what, rtol=2e-5):
    """Bit-exact without contraction.  With contraction the compositing / sphere arithmetic has
    many mul+add chains that g++ and nvcc are free to fuse differently; the oracle only models
    the two fusions that can change the MARCHING result (see oracle/ngp_oracle.c), so there the
    comparison is to rounding error instead."""
    if not o.fma or a.dtype != np.float32:
        return assert_same(a, b, what)
    assert a.shape == b.shape, what
    np.testing.assert_allclose(a, b, rtol=rtol, atol=1e-6, err_msg=what)


@pytest.fixture(scope="module", params=[False, True], ids=["nofma", "fma"])
def pair(request):
    return Oracle(fma=request.param), Reference(fma=request.param)


def test_morton_and_packbits(pair):
    o, r = pair
    g = np.random.RandomState(0)
    coords = g.randint(0, 128, (5000, 3)).astype(np.int32)
    assert_same(o.morton3D(coords), r.morton3D(coords), "morton3D")
    idx = g.randint(0, 128 ** 3, 5000).astype(np.int32)
    assert_same(o.morton3D_invert(idx), r.morton3D_invert(idx), "morton3D_invert")
    grid = g.rand(128 ** 3 // 64).astype(np.float32)
    grid[::7] = -1.0
    b0 = np.zeros(grid.size // 8, np.uint8); b1 = np.zeros_like(b0)
    o.packbits(grid, 0.37, b0); r.packbits(grid, 0.37, b1)
    assert_same(b0, b1, "packbits")


def test_intersections(pair):
    o, r = pair
    ro, rd = make_rays(3000, seed=1)
    g = np.random.RandomState(2)
    # 27 disjoint voxels, max_hits below and above the true hit count
    c = np.stack(np.meshgrid(*[np.array([-0.3, 0.0, 0.3])] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    h = np.full_like(c, 0.12)
    for mh in (1, 4, 8):
        for a, b, name in zip(o.ray_aabb_intersect(ro, rd, c, h, mh), r.ray_aabb_intersect(ro, rd, c, h, mh),
                              ("hit_cnt", "hits_t", "hits_idx")):
            if name != "hit_cnt" and mh < 7:
                # with more hits than slots the surviving subset depends on atomic order; compare
                # only rays whose hits all fit
                ok = o.ray_aabb_intersect(ro, rd, c, h, mh)[0] <= mh
                a, b = a[ok], b[ok]
            assert_same(a, b, "aabb %s max_hits=%d" % (name, mh))
    radii = g.uniform(0.05, 0.2, c.shape[0]).astype(np.float32)
    for a, b, name in zip(o.ray_sphere_intersect(ro, rd, c, radii, 8), r.ray_sphere_intersect(ro, rd, c, radii, 8),
                          ("hit_cnt", "hits_t", "hits_idx")):
        # the discriminant half_b^2 - a*c cancels catastrophically: a different fusion moves t by ~1e-4 rel
        assert_close_or_same(o, a, b, "sphere " + name, rtol=2e-3)


@pytest.mark.parametrize("cfg", [
    dict(cascades=1, scale=0.5, esf=0.0, fill=0.08),      # Synthetic-NeRF setting
    dict(cascades=1, scale=0.5, esf=0.0, fill=1.0),       # warm-up: every cell occupied
    dict(cascades=3, scale=2.0, esf=1 / 256, fill=0.15),  # real-scene setting: cascades + exponential steps
    dict(cascades=6, scale=16.0, esf=1 / 256, fill=0.12), # mip-NeRF360 recipe (benchmark_mipnerf360.sh:21-24): camera radii 1.5..12
], ids=["synthetic", "dense", "cascaded", "garden"])
def test_raymarching_train_and_test(pair, cfg):
    o, r = pair
    n = 1500
    ro, rd = make_rays(n, seed=3)
    if cfg["cascades"] == 6:
        ro = (ro * np.random.RandomState(3).choice([1.0, 2.0, 3.0, 5.0, 8.0], n).astype(np.float32)[:, None]).astype(np.float32)
    elif cfg["scale"] > 0.5:
        ro = ro * 1.5
    bf = (np.full(cfg["cascades"] * 128 ** 3 // 8, 255, np.uint8) if cfg["fill"] >= 1.0
          else syn.random_blob_bitfield(cfg["cascades"], 128, cfg["fill"], seed=4))
    ht = aabb_hits(o, ro, rd, cfg["scale"])
    noise = np.random.RandomState(5).rand(n).astype(np.float32)
    a = o.raymarching_train(ro, rd, ht, bf, cfg["cascades"], cfg["scale"], cfg["esf"], noise, 128, 1024)
    b = r.raymarching_train(ro, rd, ht, bf, cfg["cascades"], cfg["scale"], cfg["esf"], noise, 128, 1024)
    for x, y, name in zip(a, b, ("rays_a", "xyzs", "dirs", "deltas", "ts", "counter")):
        assert_same(x, y, "train " + name)
    assert a[5][0] > 0 and a[5][1] == n
    # test-time marching: several rounds with growing N_samples, in-place hits_t
    alive = np.arange(n, dtype=np.int64)
    h0, h1 = ht.copy(), ht.copy()
    for ns in (1, 2, 4):
        x = o.raymarching_test(ro, rd, h0, alive, bf, cfg["cascades"], cfg["scale"], cfg["esf"], 128, 1024, ns)
        y = r.raymarching_test(ro, rd, h1, alive, bf, cfg["cascades"], cfg["scale"], cfg["esf"], 128, 1024, ns)
        for p, q, name in zip(x, y, ("xyzs", "dirs", "deltas", "ts", "N_eff")):
            assert_same(p, q, "test %s N_samples=%d" % (name, ns))
        assert_same(h0, h1, "hits_t after N_samples=%d" % ns)


def _packed_inputs(o, n=1200, seed=6):
    ro, rd = make_rays(n, seed=seed)
    bf = syn.random_blob_bitfield(1, 128, 0.1, seed=seed)
    ht = aabb_hits(o, ro, rd)
    noise = np.random.RandomState(seed).rand(n).astype(np.float32)
    rays_a, xyzs, dirs, deltas, ts, _ = o.raymarching_train(ro, rd, ht, bf, 1, 0.5, 0.0, noise, 128, 1024)
    g = np.random.RandomState(seed + 1)
    S = ts.shape[0]
    sigmas = (g.rand(S).astype(np.float32) ** 3) * 400      # mixes near-transparent and opaque samples
    rgbs = g.rand(S, 3).astype(np.float32)
    return rays_a, sigmas, rgbs, deltas, ts


def test_composite_and_distortion(pair):
    o, r = pair
    rays_a, sigmas, rgbs, deltas, ts = _packed_inputs(o)
    fa = o.composite_train_fw(sigmas, rgbs, deltas, ts, rays_a, 1e-4)
    fb = r.composite_train_fw(sigmas, rgbs, deltas, ts, rays_a, 1e-4)
    for x, y, name in zip(fa, fb, ("total_samples", "opacity", "depth", "rgb", "ws")):
        assert_close_or_same(o, x, y, "composite_train_fw " + name)
    assert (fa[0] < rays_a[:, 2]).any(), "early stop must be exercised"
    g = np.random.RandomState(9)
    R, S = rays_a.shape[0], sigmas.shape[0]
    dO, dD, dRGB = g.randn(R).astype(np.float32), g.randn(R).astype(np.float32), g.randn(R, 3).astype(np.float32)
    dW = g.randn(S).astype(np.float32)
    total, opacity, depth, rgb, ws = fa
    ba = o.composite_train_bw(dO, dD, dRGB, dW, sigmas, rgbs, ws, deltas, ts, rays_a, opacity, depth, rgb, 1e-4)
    bb = r.composite_train_bw(dO, dD, dRGB, dW, sigmas, rgbs, ws, deltas, ts, rays_a, opacity, depth, rgb, 1e-4)
    assert_close_or_same(o, ba[0], bb[0], "dL_dsigmas"); assert_close_or_same(o, ba[1], bb[1], "dL_drgbs")
    la = o.distortion_loss_fw(ws, deltas, ts, rays_a); lb = r.distortion_loss_fw(ws, deltas, ts, rays_a)
    for x, y, name in zip(la, lb, ("loss", "ws_incl", "wts_incl")):
        assert_close_or_same(o, x, y, "distortion_fw " + name)
    dl = g.randn(R).astype(np.float32)
    assert_close_or_same(o, o.distortion_loss_bw(dl, la[1], la[2], ws, deltas, ts, rays_a),
                         r.distortion_loss_bw(dl, la[1], la[2], ws, deltas, ts, rays_a), "distortion_bw")


def test_composite_test_fw(pair):
    o, r = pair
    g = np.random.RandomState(11)
    n_rays, na, ns = 500, 300, 4
    alive0 = np.sort(g.choice(n_rays, na, replace=False)).astype(np.int64)
    sig = (g.rand(na, ns).astype(np.float32) ** 2) * 3000
    rgbs = g.rand(na, ns, 3).astype(np.float32)
    deltas = np.full((na, ns), 1.7e-3, np.float32); ts = g.rand(na, ns).astype(np.float32)
    n_eff = g.randint(0, ns + 1, na).astype(np.int32)
    hits_t = np.zeros((n_rays, 2), np.float32)
    st = [dict(alive=alive0.copy(), o=g.rand(n_rays).astype(np.float32) * 0.5, d=np.zeros(n_rays, np.float32),
               c=np.zeros((n_rays, 3), np.float32)) for _ in range(2)]
    st[1]["o"][:] = st[0]["o"]
    o.composite_test_fw(sig, rgbs, deltas, ts, hits_t, st[0]["alive"], 1e-4, n_eff, st[0]["o"], st[0]["d"], st[0]["c"])
    r.composite_test_fw(sig, rgbs, deltas, ts, hits_t, st[1]["alive"], 1e-4, n_eff, st[1]["o"], st[1]["d"], st[1]["c"])
    for k in ("alive", "o", "d", "c"):
        assert_close_or_same(o, st[0][k], st[1][k], "composite_test " + k)
    assert (st[0]["alive"] == -1).any() and (st[0]["alive"] >= 0).any()
