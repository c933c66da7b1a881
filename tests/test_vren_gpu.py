"""GPU parity: ngp_pl_amd.vren (HIP, through the C ABI) against the CPU oracle on the same inputs.

Bars (BASELINE.json north_star): packed sample indices / ray counts bit-exact; t, dt, xyz
bit-exact (the oracle's fma mode states the contraction nvcc applies, the kernels use the same
explicit fmaf); composited RGB / opacity / depth and their gradients within 1e-4 abs (we test
1e-5; the only modelled difference is __expf vs expf).
"""
import numpy as np
import pytest
import torch

from ngp_pl_amd import synthetic as syn
from oracle.vren_oracle import Oracle
from tests.helpers import aabb_hits, make_rays

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vren():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    import ngp_pl_amd.vren as v
    return v


@pytest.fixture(scope="module")
def oracle():
    return Oracle(fma=True)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def same_bits(t, a, what):
    b = t.cpu().numpy()
    assert b.shape == a.shape, "%s shape %s vs %s" % (what, b.shape, a.shape)
    if a.dtype == np.float32:
        a, b = a.view(np.uint32), b.view(np.uint32)
    assert np.array_equal(a, b), "%s: %d of %d differ" % (what, (a != b).sum(), a.size)


def test_morton_packbits(vren, oracle):
    g = np.random.RandomState(0)
    # known answers: one bit per axis, and the all-ones corner
    kat = torch.tensor([[1, 0, 0], [0, 1, 0], [0, 0, 1], [127, 127, 127], [0, 0, 0]], dtype=torch.int32).cuda()
    assert vren.morton3D(kat).cpu().tolist() == [1, 2, 4, 2097151, 0]
    coords = g.randint(0, 128, (100000, 3)).astype(np.int32)
    same_bits(vren.morton3D(dev(coords)), oracle.morton3D(coords), "morton3D")
    allidx = torch.arange(128 ** 3, dtype=torch.int32).cuda()        # full round trip over the grid
    assert torch.equal(vren.morton3D(vren.morton3D_invert(allidx)), allidx)
    idx = g.randint(0, 128 ** 3, 100000).astype(np.int32)
    same_bits(vren.morton3D_invert(dev(idx)), oracle.morton3D_invert(idx), "morton3D_invert")
    grid = g.rand(2 * 128 ** 3).astype(np.float32); grid[::5] = -1
    b0 = np.zeros(grid.size // 8, np.uint8); oracle.packbits(grid, 0.4, b0)
    b1 = torch.zeros(grid.size // 8, dtype=torch.uint8).cuda()
    vren.packbits(dev(grid), 0.4, b1)
    same_bits(b1, b0, "packbits")
    # empty input
    assert vren.morton3D(torch.zeros(0, 3, dtype=torch.int32).cuda()).shape == (0,)


def test_intersections(vren, oracle):
    ro, rd = make_rays(20000, seed=1)
    c = np.stack(np.meshgrid(*[np.array([-0.3, 0.0, 0.3])] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    h = np.full_like(c, 0.12)
    for mh in (1, 8):
        got = vren.ray_aabb_intersect(dev(ro), dev(rd), dev(c), dev(h), mh)
        want = oracle.ray_aabb_intersect(ro, rd, c, h, mh)
        for t, a, name in zip(got, want, ("hit_cnt", "hits_t", "hits_idx")):
            same_bits(t, a, "aabb %s mh=%d" % (name, mh))
    # the hot-path call: one box, one hit (rendering.py:27-28)
    got = vren.ray_aabb_intersect(dev(ro), dev(rd), torch.zeros(1, 3).cuda(), torch.full((1, 3), 0.5).cuda(), 1)
    want = oracle.ray_aabb_intersect(ro, rd, np.zeros((1, 3), np.float32), np.full((1, 3), 0.5, np.float32), 1)
    for t, a, name in zip(got, want, ("hit_cnt", "hits_t", "hits_idx")):
        same_bits(t, a, "scene box " + name)
    assert (want[1][:, 0, 0] == -1).any() and (want[1][:, 0, 0] == 0).any()   # misses and inside-origin covered
    radii = np.random.RandomState(2).uniform(0.05, 0.2, c.shape[0]).astype(np.float32)
    got = vren.ray_sphere_intersect(dev(ro), dev(rd), dev(c), dev(radii), 8)
    want = oracle.ray_sphere_intersect(ro, rd, c, radii, 8)
    assert np.array_equal(got[0].cpu().numpy(), want[0])
    np.testing.assert_allclose(got[1].cpu().numpy(), want[1], rtol=2e-3, atol=1e-6)   # cancellation in the discriminant


CFGS = [
    dict(cascades=1, scale=0.5, esf=0.0, fill=0.08, n=8192),
    dict(cascades=1, scale=0.5, esf=0.0, fill=1.0, n=2048),
    dict(cascades=3, scale=2.0, esf=1 / 256, fill=0.15, n=4096),
    dict(cascades=1, scale=0.5, esf=0.0, fill=0.0, n=512),     # nothing occupied: S = 0
    # the mip-NeRF360 recipe (benchmarking/benchmark_mipnerf360.sh:21-24: --scale 16 => networks.py:26 gives 6 cascades,
    # train.py:95-96 exp_step_factor 1/256): rays start at radius 1.5..12 so that samples land in cascades 1..5, dt runs
    # from sqrt(3)/1024 up to the sqrt(3)*2*16/128 clamp, mip_from_dt takes over from mip_from_pos far out
    dict(cascades=6, scale=16.0, esf=1 / 256, fill=0.12, n=4096, spread=True),
    dict(cascades=6, scale=16.0, esf=1 / 256, fill=1.0, n=1024, spread=True),
]
IDS = ["synthetic", "dense", "cascaded", "empty", "garden", "garden_dense"]


def scaled_origins(ro, cfg, seed=0):
    """Spread the camera radii over the cascades of a large scene (per-ray factor 1..8 on the 1.5-radius hemisphere)."""
    if cfg.get("spread"):
        f = np.random.RandomState(seed).choice([1.0, 2.0, 3.0, 5.0, 8.0], ro.shape[0]).astype(np.float32)
        return (ro * f[:, None]).astype(np.float32)
    return ro * 1.5 if cfg["scale"] > 0.5 else ro


@pytest.mark.parametrize("cfg", CFGS, ids=IDS)
def test_raymarching_train(vren, oracle, cfg):
    n = cfg["n"]
    ro, rd = make_rays(n, seed=3)
    ro = scaled_origins(ro, cfg, seed=3)
    if cfg["fill"] >= 1.0:
        bf = np.full(cfg["cascades"] * 128 ** 3 // 8, 255, np.uint8)
    elif cfg["fill"] == 0.0:
        bf = np.zeros(cfg["cascades"] * 128 ** 3 // 8, np.uint8)
    else:
        bf = syn.random_blob_bitfield(cfg["cascades"], 128, cfg["fill"], seed=4)
    ht = aabb_hits(oracle, ro, rd, cfg["scale"])
    noise = np.random.RandomState(5).rand(n).astype(np.float32)
    want = oracle.raymarching_train(ro, rd, ht, bf, cfg["cascades"], cfg["scale"], cfg["esf"], noise, 128, 1024)
    got = vren.raymarching_train(dev(ro), dev(rd), dev(ht), dev(bf), cfg["cascades"], cfg["scale"], cfg["esf"],
                                 dev(noise), 128, 1024)
    for t, a, name in zip(got, want, ("rays_a", "xyzs", "dirs", "deltas", "ts", "counter")):
        same_bits(t, a, "train " + name)
    if cfg["fill"] >= 1.0:
        assert want[0][:, 2].max() > 400      # long rays exercised (cube crossing ~591 steps)
    if cfg["cascades"] == 6:
        # every cascade is visited; the step grows from the sqrt(3)/1024 floor with t/256 (the sqrt(3)*2*16/128 = 0.43 ceiling
        # of raymarching.cu:11-13 is out of reach inside a scale-16 box: t <= 2*16*sqrt(3) gives dt <= 0.22)
        m = np.abs(want[1]).max(1)
        assert (m < 0.5).any() and (m > 8.0).any() and want[3].min() < 2e-3 and want[3].max() > 0.03


@pytest.mark.parametrize("cfg", CFGS[:3] + CFGS[4:], ids=IDS[:3] + IDS[4:])
def test_raymarching_test(vren, oracle, cfg):
    n = 4096
    ro, rd = make_rays(n, seed=7)
    ro = scaled_origins(ro, cfg, seed=7)
    bf = (np.full(cfg["cascades"] * 128 ** 3 // 8, 255, np.uint8) if cfg["fill"] >= 1.0
          else syn.random_blob_bitfield(cfg["cascades"], 128, cfg["fill"], seed=8))
    ht = aabb_hits(oracle, ro, rd, cfg["scale"])
    alive = np.arange(0, n, 2, dtype=np.int64)        # a non-trivial alive subset
    h_cpu = ht.copy(); h_gpu = dev(ht)
    for ns in (1, 3, 8):
        want = oracle.raymarching_test(ro, rd, h_cpu, alive, bf, cfg["cascades"], cfg["scale"], cfg["esf"], 128, 1024, ns)
        got = vren.raymarching_test(dev(ro), dev(rd), h_gpu, dev(alive), dev(bf), cfg["cascades"], cfg["scale"], cfg["esf"],
                                    128, 1024, ns)
        for t, a, name in zip(got, want, ("xyzs", "dirs", "deltas", "ts", "N_eff")):
            same_bits(t, a, "test %s ns=%d" % (name, ns))
        same_bits(h_gpu, h_cpu, "hits_t ns=%d" % ns)


@pytest.mark.parametrize("scale,cascades,esf,n", [(0.5, 1, 0.0, 8192), (2.0, 3, 1 / 256, 3001), (16.0, 6, 1 / 256, 2050)])
def test_the_steppers_two_launch_march_equals_the_api_shaped_sequence(vren, oracle, scale, cascades, esf, n):
    """ngp_march_train_fused (prologue + count in one launch, prefix + expansion in the next: what the native stepper enqueues) against
    ngp_ray_aabb_near_noise -> ngp_raymarching_train_count -> ngp_raymarching_train_write, the sequence that mirrors the reference's
    API (and which the tests above hold to the oracle bit for bit): hit intervals, jitter, (ray, start, count) triples, the pinned
    {S, R} record and every packed sample, bit for bit; ray counts that are not a multiple of the 4 rays a workgroup takes."""
    import ctypes as C
    from ngp_pl_amd import _lib
    from ngp_pl_amd._lib import call, ptr, stream
    g = np.random.RandomState(5)
    ro = ((g.rand(n, 3) - 0.5) * 3.0 * scale).astype(np.float32)
    tgt = ((g.rand(n, 3) - 0.5) * scale).astype(np.float32)
    rd = tgt - ro; rd /= np.linalg.norm(rd, axis=1, keepdims=True); rd[: n // 9] *= -1          # some rays miss the box
    bf = syn.random_blob_bitfield(cascades, 128, 0.15, seed=11)
    ro_d, rd_d, bf_d = dev(ro), dev(rd.astype(np.float32)), dev(bf)
    centre, half = torch.zeros(1, 3).cuda(), torch.full((1, 3), scale).cuda()
    seed, near, M = 0x1234567890ABCDEF, 0.01, 1024
    f32, i32 = dict(dtype=torch.float32, device="cuda"), dict(dtype=torch.int32, device="cuda")
    # the API-shaped sequence
    hits_a, noise_a = torch.empty(n, 2, **f32), torch.empty(n, **f32)
    rays_a_a = torch.empty(n, 3, dtype=torch.int64, device="cuda"); scratch_a = torch.empty(n * M, **f32)
    counter_a = torch.full((4,), -1, dtype=torch.int32).pin_memory()
    call("ngp_ray_aabb_near_noise", ptr(ro_d), ptr(rd_d), ptr(centre), ptr(half), near, n, seed, ptr(hits_a), ptr(noise_a), stream())
    call("ngp_raymarching_train_count", ptr(ro_d), ptr(rd_d), ptr(hits_a), ptr(bf_d), cascades, scale, esf, ptr(noise_a), 128, M, n,
         ptr(rays_a_a), counter_a.data_ptr(), ptr(scratch_a), stream())
    torch.cuda.synchronize()
    S = int(counter_a[0])
    assert S > 1000 and int(counter_a[1]) == n
    out_a = [torch.empty(S, 3, **f32), torch.empty(S, 3, **f32), torch.empty(S, **f32), torch.empty(S, **f32)]
    call("ngp_raymarching_train_write", ptr(ro_d), ptr(rd_d), ptr(rays_a_a), ptr(scratch_a), scale, esf, 128, M, n, *[ptr(t) for t in out_a], stream())
    # the stepper's two launches
    hits_b, noise_b = torch.empty(n, 2, **f32), torch.empty(n, **f32)
    rays_a_b = torch.empty(n, 3, dtype=torch.int64, device="cuda"); scratch_b = torch.empty(n * M, **f32)
    counts = torch.full((n + 8,), -7, **i32)
    counter_b = torch.full((4,), -1, dtype=torch.int32).pin_memory()
    cap = S + 4096
    out_b = [torch.zeros(cap, 3, **f32), torch.zeros(cap, 3, **f32), torch.zeros(cap, **f32), torch.zeros(cap, **f32)]
    call("ngp_march_train_fused", ptr(ro_d), ptr(rd_d), ptr(centre), ptr(half), near, seed, ptr(bf_d), cascades, scale, esf, 128, M, n,
         ptr(hits_b), ptr(noise_b), ptr(rays_a_b), ptr(counts), counter_b.data_ptr(), ptr(scratch_b), *[ptr(t) for t in out_b], stream())
    torch.cuda.synchronize()
    assert counter_b[:2].tolist() == [S, n]
    assert torch.equal(hits_a.view(torch.int32), hits_b.view(torch.int32)) and torch.equal(noise_a, noise_b)
    assert torch.equal(rays_a_a, rays_a_b) and torch.equal(counts[:n].long(), rays_a_a[:, 2]) and bool((counts[n:] == -7).all())
    for a, b, name in zip(out_a, out_b, ("xyzs", "dirs", "deltas", "ts")):
        assert torch.equal(a.view(torch.int32), b[:S].view(torch.int32)), name
        assert not bool(b[S:].any()), name + ": written past S"


def _packed(oracle, n=8192, seed=6):
    ro, rd = make_rays(n, seed=seed)
    bf = syn.random_blob_bitfield(1, 128, 0.1, seed=seed)
    ht = aabb_hits(oracle, ro, rd)
    noise = np.random.RandomState(seed).rand(n).astype(np.float32)
    rays_a, xyzs, dirs, deltas, ts, _ = oracle.raymarching_train(ro, rd, ht, bf, 1, 0.5, 0.0, noise, 128, 1024)
    g = np.random.RandomState(seed + 1)
    S = ts.shape[0]
    sigmas = (g.rand(S).astype(np.float32) ** 3) * 400
    rgbs = g.rand(S, 3).astype(np.float32)
    return rays_a, sigmas, rgbs, deltas, ts


def _T_sequence(sig, dl, T0=1.0):
    """Transmittance after each sample, float32, sequential, expf -- the oracle's arithmetic (volumerendering.cu:30-43,229-247)."""
    T, out = np.float32(T0), []
    for s_, d_ in zip(sig, dl):
        a = np.float32(1) - np.float32(np.exp(np.float32(-(np.float32(s_) * np.float32(d_)))))
        T = np.float32(T * (np.float32(1) - a))
        out.append(T)
    return np.array(out, np.float32)


def _stop_is_at_the_threshold(T_seq, k, thr=1e-4):
    """The two sides may only disagree about a stop where T sits on the threshold: |T - thr| <= (4 + 2 (k + 1)) ulp(thr) -- each of the
    k + 1 factors exp(-sigma delta) differs by <= 2 ulp between `__expf` (v_exp_f32 on x log2 e) and expf, 4 ulp for the product's
    own rounding / association (the wave multiplies as a scan tree)."""
    ulp = np.spacing(np.float32(thr))
    return abs(float(T_seq[k]) - thr) <= (4 + 2 * (k + 1)) * float(ulp)


def test_composite_train(vren, oracle):
    rays_a, sigmas, rgbs, deltas, ts = _packed(oracle)
    want = oracle.composite_train_fw(sigmas, rgbs, deltas, ts, rays_a, 1e-4)
    got = vren.composite_train_fw(dev(sigmas), dev(rgbs), dev(deltas), dev(ts), dev(rays_a), 1e-4)
    # closed form: constant sigma along a ray -> opacity = 1 - exp(-sigma * sum(delta)) is covered in test_properties
    tol = dict(rtol=0, atol=1e-5)
    # ray counts: bit-exact, except where the deciding sample leaves T ON the threshold (the only modelled difference is __expf)
    n_g, n_o = got[0].cpu().numpy(), want[0]
    differ = np.nonzero(n_g != n_o)[0]
    assert len(differ) < 1e-3 * len(n_o), "total_samples differs on %d rays" % len(differ)
    by_ray = {int(r): (int(s0), int(n)) for r, s0, n in rays_a}
    for r in differ:
        s0, n = by_ray[int(r)]
        k = int(min(n_g[r], n_o[r]))                    # the sample at which one side stopped (counted samples precede the stop)
        assert k < n
        Ts = _T_sequence(sigmas[s0:s0 + k + 1], deltas[s0:s0 + k + 1])
        assert _stop_is_at_the_threshold(Ts, k), "ray %d: counts %d vs %d but T = %.9g at the deciding sample" % (r, n_g[r], n_o[r], Ts[k])
    for t, a, name in zip(got[1:], want[1:], ("opacity", "depth", "rgb", "ws")):
        np.testing.assert_allclose(t.cpu().numpy(), a, err_msg=name, **tol)
    g = np.random.RandomState(9)
    R, S = rays_a.shape[0], sigmas.shape[0]
    dO, dD, dRGB, dW = (g.randn(R).astype(np.float32), g.randn(R).astype(np.float32),
                        g.randn(R, 3).astype(np.float32), g.randn(S).astype(np.float32))
    total, opacity, depth, rgb, ws = want
    wb = oracle.composite_train_bw(dO, dD, dRGB, dW, sigmas, rgbs, ws, deltas, ts, rays_a, opacity, depth, rgb, 1e-4)
    gb = vren.composite_train_bw(dev(dO), dev(dD), dev(dRGB), dev(dW), dev(sigmas), dev(rgbs), dev(ws), dev(deltas), dev(ts),
                                 dev(rays_a), dev(opacity), dev(depth), dev(rgb), 1e-4)
    # gradients w.r.t. sigma are O(delta * |seeds|): compare relative to their scale
    scale = np.abs(wb[0]).max()
    np.testing.assert_allclose(gb[0].cpu().numpy() / scale, wb[0] / scale, rtol=0, atol=1e-5, err_msg="dL_dsigmas")
    np.testing.assert_allclose(gb[1].cpu().numpy(), wb[1], rtol=0, atol=1e-5, err_msg="dL_drgbs")
    # distortion loss on the same packing
    wl = oracle.distortion_loss_fw(ws, deltas, ts, rays_a)
    gl = vren.distortion_loss_fw(dev(ws), dev(deltas), dev(ts), dev(rays_a))
    for t, a, name in zip(gl, wl, ("loss", "ws_incl", "wts_incl")):
        np.testing.assert_allclose(t.cpu().numpy(), a, rtol=1e-5, atol=1e-6, err_msg="distortion " + name)
    dl = g.randn(R).astype(np.float32)
    wdb = oracle.distortion_loss_bw(dl, wl[1], wl[2], ws, deltas, ts, rays_a)
    gdb = vren.distortion_loss_bw(dev(dl), dev(wl[1]), dev(wl[2]), dev(ws), dev(deltas), dev(ts), dev(rays_a))
    np.testing.assert_allclose(gdb.cpu().numpy(), wdb, rtol=1e-5, atol=1e-5, err_msg="distortion bw")


def test_composite_test_fw(vren, oracle):
    g = np.random.RandomState(11)
    n_rays, na, ns = 5000, 3000, 4
    alive0 = np.sort(g.choice(n_rays, na, replace=False)).astype(np.int64)
    sig = (g.rand(na, ns).astype(np.float32) ** 2) * 3000
    rgbs = g.rand(na, ns, 3).astype(np.float32)
    deltas = np.full((na, ns), 1.7e-3, np.float32); ts = g.rand(na, ns).astype(np.float32)
    n_eff = g.randint(0, ns + 1, na).astype(np.int32)
    hits_t = np.zeros((n_rays, 2), np.float32)
    o0 = g.rand(n_rays).astype(np.float32) * 0.5
    a_c, o_c, d_c, c_c = alive0.copy(), o0.copy(), np.zeros(n_rays, np.float32), np.zeros((n_rays, 3), np.float32)
    oracle.composite_test_fw(sig, rgbs, deltas, ts, hits_t, a_c, 1e-4, n_eff, o_c, d_c, c_c)
    a_g, o_g, d_g, c_g = dev(alive0), dev(o0), torch.zeros(n_rays).cuda(), torch.zeros(n_rays, 3).cuda()
    vren.composite_test_fw(dev(sig), dev(rgbs), dev(deltas), dev(ts), dev(hits_t), a_g, 1e-4, dev(n_eff), o_g, d_g, c_g)
    # the alive set: identical, except for rays one of whose samples leaves T ON the threshold (__expf vs expf)
    differ = np.nonzero(a_g.cpu().numpy() != a_c)[0]
    assert len(differ) < 1e-3 * na
    for i in differ:
        r = int(alive0[i])
        Ts = _T_sequence(sig[i, :n_eff[i]], deltas[i, :n_eff[i]], T0=np.float32(1) - o0[r])
        assert any(_stop_is_at_the_threshold(Ts, k) for k in range(len(Ts))), "alive[%d] (ray %d): %d vs %d, T = %s" % (i, r, a_g[i], a_c[i], Ts)
    for t, a, name in ((o_g, o_c, "opacity"), (d_g, d_c, "depth"), (c_g, c_c, "rgb")):
        np.testing.assert_allclose(t.cpu().numpy(), a, rtol=0, atol=1e-5, err_msg=name)


def test_rejects_cpu_and_noncontiguous(vren):
    x = torch.zeros(4, 3)
    with pytest.raises(RuntimeError):      # CHECK_CUDA of the reference (include/utils.h:4)
        vren.morton3D(x.int())
    y = torch.zeros(3, 4, dtype=torch.int32).cuda().t()
    with pytest.raises(RuntimeError):      # CHECK_CONTIGUOUS (include/utils.h:5)
        vren.morton3D(y)


def test_marching_guards_end_rays_that_would_never_end(vren):
    """Rays whose far hit is infinite (or whose t is so large that a step is absorbed by rounding) make the reference's loops
    (raymarching.cu:225-232) spin for ever.  Here every marching kernel ends such a ray, leaves the other rays of the
    launch untouched (bit-identical to a launch without the bad rays), and counts the event (ngp_march_guard_read).  Rays from
    an AABB intersection never trip a guard."""
    from ngp_pl_amd import _lib
    n = 512
    ro, rd = make_rays(n, seed=11)
    o = Oracle(fma=True)
    ht = aabb_hits(o, ro, rd, 0.5)
    noise = np.random.RandomState(2).rand(n).astype(np.float32)
    bf = syn.random_blob_bitfield(1, 128, 0.08, seed=4)
    _lib.march_guard_counts(reset=True)
    good = vren.raymarching_train(dev(ro), dev(rd), dev(ht), dev(bf), 1, 0.5, 0.0, dev(noise), 128, 1024)
    torch.cuda.synchronize()
    assert _lib.march_guard_counts() == [0, 0, 0, 0]
    bad = ht.copy()
    hit = np.nonzero(ht[:, 0] >= 0)[0]
    worst = hit[:7]
    bad[worst[:4], 1] = np.inf                                   # no far hit: t runs until a step is absorbed
    bad[worst[4:], 0] = 3.0e7; bad[worst[4:], 1] = 4.0e7         # t so large that t + dt == t from the start
    empty = np.zeros_like(bf)
    got = vren.raymarching_train(dev(ro), dev(rd), dev(bad), dev(empty), 1, 0.5, 0.0, dev(noise), 128, 1024)
    torch.cuda.synchronize()
    assert int(got[5][0]) == 0                                   # empty grid: no samples, and the launch came back
    c = _lib.march_guard_counts(reset=True)
    assert c[0] + c[1] >= 7 and c[2] == 0, c
    got = vren.raymarching_train(dev(ro), dev(rd), dev(bad), dev(bf), 1, 0.5, 0.0, dev(noise), 128, 1024)
    torch.cuda.synchronize()
    keep = np.ones(n, bool); keep[worst] = False
    ga, wa = got[0].cpu().numpy(), good[0].cpu().numpy()
    assert np.array_equal(ga[keep][:, 2], wa[keep][:, 2])        # the other rays' sample counts are untouched
    # their samples too: compare ray by ray through the start offsets
    gt, wt = got[4].cpu().numpy(), good[4].cpu().numpy()
    for r in np.nonzero(keep)[0][::17]:
        a, b = ga[r], wa[r]
        assert np.array_equal(gt[a[1]:a[1] + a[2]].view(np.uint32), wt[b[1]:b[1] + b[2]].view(np.uint32))
    # test-time kernel: same bad rays, bounded
    alive = torch.arange(n, device="cuda")
    hits = dev(bad.copy())
    out = vren.raymarching_test(dev(ro), dev(rd), hits, alive, dev(np.zeros_like(bf)), 1, 0.5, 0.0, 128, 1024, 4)
    torch.cuda.synchronize()
    assert int(out[4].sum()) == 0
    _lib.march_guard_counts(reset=True)


@pytest.mark.parametrize("first_k", [1, 4, 8, 32, 64])
def test_composite_probe_lists_what_the_reference_composite_goes_on_to_read(vren, oracle, first_k):
    """ngp_composite_probe (two-round forward) against the oracle's composite_train_fw (bit-pinned to volumerendering.cu:20-44):
    a ray's remaining samples are listed iff it has more than first_k samples and the reference loop is still running behind
    the first first_k (its count of composited samples `total_samples` has reached first_k); rays whose transmittance sits within
    rounding of the threshold at that point may fall either way (counted, must be rare).  Also the padded first-K list of
    ngp_raymarching_train_write_k and the list variants of the hash / field forward against the full-range kernels."""
    from ngp_pl_amd import _lib
    from ngp_pl_amd._lib import call, ptr, stream
    rays_a, sigmas, rgbs, deltas, ts = _packed(oracle)
    R, S = rays_a.shape[0], sigmas.shape[0]
    total = oracle.composite_train_fw(sigmas, rgbs, deltas, ts, rays_a, 1e-4)[0]
    N = rays_a[:, 2]
    want_cont = (N > first_k) & (total >= first_k)
    lst = torch.full((S,), -7, dtype=torch.int32, device="cuda"); cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
    d_sig, d_del, d_rays = dev(sigmas), dev(deltas), dev(rays_a)              # (held: a temporary would be freed before the launch)
    call("ngp_composite_probe", ptr(d_sig), ptr(d_del), ptr(d_rays), first_k, 1e-4, R, ptr(lst), ptr(cnt), stream())
    torch.cuda.synchronize()
    n = int(cnt[0])
    got = np.sort(lst[:n].cpu().numpy())
    assert (lst[n:] == -7).all() and len(np.unique(got)) == n                  # nothing written past the count, no sample twice
    want_ids = np.concatenate([np.arange(rays_a[r, 1] + first_k, rays_a[r, 1] + N[r]) for r in np.nonzero(want_cont)[0]] + [np.zeros(0, np.int64)])
    # threshold ties: rays whose membership differs
    ray_of = np.repeat(np.arange(R), N)
    diff_rays = np.unique(ray_of[np.setxor1d(got, want_ids).astype(np.int64)]) if len(np.setxor1d(got, want_ids)) else np.zeros(0)
    assert len(diff_rays) <= max(1, R // 500), (first_k, len(diff_rays))
    assert want_cont.sum() > 0 or first_k >= 32


def test_list_forward_kernels_match_the_full_range_kernels():
    """ngp_hashgrid_fwd_list / ngp_field_fwd_list write, for the listed samples, exactly what ngp_hashgrid_fwd / ngp_field_fwd write
    for them (per-sample kernels), leave every other sample untouched, skip padding entries (-1), and honour a device-side count."""
    import ctypes as C
    import math
    from ngp_pl_amd import _lib
    from ngp_pl_amd._lib import GridMeta, call, ptr, stream
    g = torch.Generator(device="cuda").manual_seed(3)
    S = 50000
    meta = GridMeta()
    call("ngp_grid_meta_init", C.byref(meta), 16, 2, 19, 16, float(math.exp(math.log(2048 * 0.5 / 16) / 15)))
    x = (torch.rand(S, 3, device="cuda", generator=g) - 0.5) * 0.98
    d = torch.randn(S, 3, device="cuda", generator=g)
    table = ((torch.rand(meta.offset[16], 2, device="cuda", generator=g) - 0.5) * 0.4).half()
    dw = (torch.randn(3072, device="cuda", generator=g) * 0.2).half(); rw = (torch.randn(7168, device="cuda", generator=g) * 0.2).half()
    mn = torch.full((3,), -0.5, device="cuda"); mx = torch.full((3,), 0.5, device="cuda")
    feats = torch.empty(16, S, 2, dtype=torch.float16, device="cuda")
    sig = torch.empty(S, device="cuda"); rgb = torch.empty(S, 3, device="cuda"); h = torch.empty(S, 16, dtype=torch.float16, device="cuda")
    call("ngp_hashgrid_fwd", ptr(x), ptr(mn), ptr(mx), ptr(table), C.byref(meta), S, ptr(feats), stream())
    call("ngp_field_fwd", ptr(feats), ptr(d), ptr(dw), ptr(rw), S, ptr(sig), ptr(rgb), ptr(h), stream())
    # a list of runs with padding in between, in shuffled run order
    ids = torch.cat([torch.arange(a * 8, a * 8 + 8) for a in torch.randperm(S // 8 - 1)[:2000].tolist()])        # 2000 runs of 8 consecutive samples
    lst = torch.stack([ids.view(-1, 8), torch.full((2000, 8), -1, dtype=torch.long)], 1).reshape(-1)                 # run, padding, run, ...
    lst = lst.int().cuda().contiguous()
    n_dev = torch.tensor([lst.numel() - 16 * 5], dtype=torch.int32, device="cuda")                                  # the last 5 runs are beyond the count
    f2 = torch.full_like(feats, 9.0); s2 = torch.full_like(sig, -3.0); c2 = torch.full_like(rgb, -3.0); h2 = torch.full_like(h, 9.0)
    call("ngp_hashgrid_fwd_list", ptr(x), ptr(mn), ptr(mx), ptr(table), C.byref(meta), S, ptr(lst), lst.numel(), ptr(n_dev), ptr(f2), stream())
    call("ngp_field_fwd_list", ptr(f2), ptr(d), ptr(dw), ptr(rw), S, ptr(lst), lst.numel(), ptr(n_dev), ptr(s2), ptr(c2), ptr(h2), stream())
    torch.cuda.synchronize()
    listed = lst[:int(n_dev[0])]
    listed = listed[listed >= 0].long()
    mask = torch.zeros(S, dtype=torch.bool, device="cuda"); mask[listed] = True
    assert int(mask.sum()) == 1995 * 8
    assert torch.equal(f2[:, mask], feats[:, mask]) and bool((f2[:, ~mask] == 9.0).all())
    assert torch.equal(s2[mask], sig[mask]) and torch.equal(c2[mask], rgb[mask]) and torch.equal(h2[mask], h[mask])
    assert bool((s2[~mask] == -3.0).all()) and bool((c2[~mask] == -3.0).all()) and bool((h2[~mask] == 9.0).all())


@pytest.mark.parametrize("first_k", [1, 8, 32, 64])
def test_first_k_lists_of_the_train_write(first_k):
    """ngp_raymarching_train_write_k / _kc write the same samples as ngp_raymarching_train_write and list every ray's first
    min(N, first_k) sample ids: padded (ray-major, -1 where a ray has fewer) or compact in ray order at the offsets that
    ngp_raymarching_train_count_k's scan left in offs_k (total in counter[3]); the side counter is cleared."""
    from ngp_pl_amd._lib import call, ptr, stream
    g = np.random.RandomState(5)
    R, M = 1003, 128                                                           # not a multiple of the 4-ray workgroup
    N = g.randint(0, M + 1, R); N[g.rand(R) < 0.6] = 0                        # most rays have no samples (late-training shape)
    start = np.concatenate([[0], np.cumsum(N)[:-1]])
    S = int(N.sum())
    perm = g.permutation(R)                                                    # rays_a rows are in march order, r is the ray id
    rays_a = np.stack([perm, start, N], 1).astype(np.int64)
    scratch = np.sort(g.rand(R, M).astype(np.float32) * 2 + 0.1, axis=1)
    ro = dev(g.randn(R, 3).astype(np.float32) * 0.1); rd = dev(g.randn(R, 3).astype(np.float32))
    d_rays, d_scr = dev(rays_a), dev(scratch)
    want_rows = np.full((R, first_k), -1, np.int64)
    for r in range(R):
        m = min(int(N[r]), first_k)
        want_rows[r, :m] = start[r] + np.arange(m)
    want_compact = want_rows[want_rows >= 0]                                   # ray order
    offs = dev(np.concatenate([[0], np.cumsum(np.minimum(N, first_k))[:-1]]).astype(np.int32))

    def outs():
        return [torch.full((S, 3), 7.0, device="cuda"), torch.full((S, 3), 7.0, device="cuda"), torch.full((S,), 7.0, device="cuda"), torch.full((S,), 7.0, device="cuda")]
    ref = outs()
    call("ngp_raymarching_train_write", ptr(ro), ptr(rd), ptr(d_rays), ptr(d_scr), 0.5, 0.0, 128, M, R, *[ptr(o) for o in ref], stream())
    pad = outs(); lst = torch.full((R * first_k,), -7, dtype=torch.int32, device="cuda"); side = torch.full((4,), 5, dtype=torch.int32, device="cuda")
    call("ngp_raymarching_train_write_k", ptr(ro), ptr(rd), ptr(d_rays), ptr(d_scr), 0.5, 0.0, 128, M, R, *[ptr(o) for o in pad], first_k, ptr(lst), ptr(side), stream())
    com = outs(); lst_c = torch.full((R * first_k,), -7, dtype=torch.int32, device="cuda"); side_c = torch.full((4,), 5, dtype=torch.int32, device="cuda")
    call("ngp_raymarching_train_write_kc", ptr(ro), ptr(rd), ptr(d_rays), ptr(d_scr), 0.5, 0.0, 128, M, R, *[ptr(o) for o in com], first_k, ptr(offs), ptr(lst_c),
         ptr(side_c), stream())
    torch.cuda.synchronize()
    for a, b, c in zip(ref, pad, com):
        assert torch.equal(a, b) and torch.equal(a, c)
    assert np.array_equal(lst.cpu().numpy().reshape(R, first_k), want_rows) and side.tolist() == [0, 5, 5, 5]
    n = len(want_compact)
    assert np.array_equal(lst_c[:n].cpu().numpy(), want_compact) and bool((lst_c[n:] == -7).all()) and side_c.tolist() == [0, 5, 5, 5]


@pytest.mark.parametrize("first_k", [4, 32])
def test_train_count_k_scans_the_first_k_offsets(vren, first_k):
    """ngp_raymarching_train_count_k = ngp_raymarching_train_count (same rays_a, scratch, {S, R}) + offs_k = exclusive scan of
    min(N, first_k) in ray order and counter[3] = its total; more than one 8192-ray tile of the scan kernel."""
    from ngp_pl_amd._lib import call, ptr, stream
    g = np.random.RandomState(11)
    R, G, M = 20000, 128, 256
    ro = (g.rand(R, 3).astype(np.float32) - 0.5) * 3.0
    rd = -ro / np.linalg.norm(ro, axis=1, keepdims=True) + g.randn(R, 3).astype(np.float32) * 0.2
    rd = (rd / np.linalg.norm(rd, axis=1, keepdims=True)).astype(np.float32)
    bits = np.packbits((g.rand(G ** 3) < 0.4).astype(np.uint8), bitorder="little")       # dense enough for rays with more than first_k samples
    d_o, d_d, d_bits = dev(ro), dev(rd), dev(bits)
    center = torch.zeros(1, 3, device="cuda"); half = torch.full((1, 3), 0.5, device="cuda")
    hits = torch.empty(R, 2, device="cuda"); noise = dev(g.rand(R).astype(np.float32))
    call("ngp_ray_aabb_near", ptr(d_o), ptr(d_d), ptr(center), ptr(half), 0.05, R, ptr(hits), stream())
    res = []
    for with_k in (False, True):
        rays_a = torch.full((R, 3), -1, dtype=torch.int64, device="cuda"); cnt = torch.full((4,), -9, dtype=torch.int32, device="cuda")
        scr = torch.zeros(R, M, device="cuda"); offs = torch.full((R,), -9, dtype=torch.int32, device="cuda")
        if with_k:
            call("ngp_raymarching_train_count_k", ptr(d_o), ptr(d_d), ptr(hits), ptr(d_bits), 1, 0.5, 0.0, ptr(noise), G, M, R, ptr(rays_a), ptr(cnt), ptr(scr),
                 first_k, ptr(offs), stream())
        else:
            call("ngp_raymarching_train_count", ptr(d_o), ptr(d_d), ptr(hits), ptr(d_bits), 1, 0.5, 0.0, ptr(noise), G, M, R, ptr(rays_a), ptr(cnt), ptr(scr), stream())
        torch.cuda.synchronize()
        res.append((rays_a.cpu(), cnt.cpu(), scr.cpu(), offs.cpu()))
    (ra0, c0, s0, _), (ra1, c1, s1, offs) = res
    assert torch.equal(ra0, ra1) and torch.equal(s0, s1) and c0[:2].tolist() == c1[:2].tolist() and c0[3] == -9
    N = ra1[:, 2].numpy()
    assert N.sum() == int(c1[0]) > 0 and (N > first_k).any() and (N == 0).any()
    want = np.concatenate([[0], np.cumsum(np.minimum(N, first_k))[:-1]])
    assert np.array_equal(offs.numpy(), want) and int(c1[3]) == int(np.minimum(N, first_k).sum())
