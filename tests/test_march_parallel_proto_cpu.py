"""CPU: the wave-parallel march algorithm behind march_train_count_wave_kernel (tools/march_parallel_proto.py, numpy:
fixed t sequence per ray, 64 candidates per tile, successor chase) emits exactly the serial loop's samples."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import march_parallel_proto as P                      # noqa: E402
from ngp_pl_amd import synthetic as syn               # noqa: E402
from oracle.vren_oracle import Oracle                 # noqa: E402


@pytest.mark.parametrize("cascades,scale,esf,fill,max_samples", [(1, 0.5, 0.0, 0.1, 1024), (3, 2.0, 1 / 256, 0.2, 1024), (1, 0.5, 0.0, 1.0, 48)],
                         ids=["synthetic", "cascaded_exponential", "dense_coarse_steps"])
def test_tile_parallel_march_equals_the_serial_loop(cascades, scale, esf, fill, max_samples):
    F = np.float32
    n = 90
    o = Oracle(fma=False)                             # the prototype uses separate multiply and add
    rng = np.random.default_rng(5)
    bf = np.full(cascades * 128 ** 3 // 8, 255, np.uint8) if fill >= 1 else syn.random_blob_bitfield(cascades, 128, fill, seed=22)
    ro = (rng.random((n, 3), dtype=F) - F(0.5)) * F(3 * scale)
    rd = rng.standard_normal((n, 3)).astype(F); rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    _, hits, _ = o.ray_aabb_intersect(ro, rd, np.zeros((1, 3), F), np.full((1, 3), scale, F), 1)
    hits_t = hits[:, 0].copy(); near = (hits_t[:, 0] >= 0) & (hits_t[:, 0] < 0.01); hits_t[near, 0] = 0.01
    noise = rng.random(n, dtype=F)
    rays_a, _, _, _, ts, _ = o.raymarching_train(ro, rd, hits_t, bf, cascades, scale, esf, noise, 128, max_samples)
    assert len(ts) > 200        # (the N_samples < max_samples cap itself can hardly bind: dt = sqrt3/max_samples and the box diagonal is sqrt3)
    for r in range(n):
        got = np.array(P.march_ray(ro[r], rd[r], hits_t[r, 0], hits_t[r, 1], noise[r], bf, cascades, 128, scale, esf, max_samples), F)
        row = rays_a[rays_a[:, 0] == r][0]
        want = ts[row[1]:row[1] + row[2]]
        assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)), r


def test_closed_form_lattice_tile_equals_the_chain_of_adds():
    """lattice_tile_const_dt (march.hip) replaces the 64 dependent float adds per tile of the constant-step lattice by integer
    arithmetic on the mantissa; its numpy restatement must reproduce the chain bit for bit, across binade crossings, for the
    reference's step sqrt(3)/1024 and for arbitrary steps, and must decline (fallback to the chain) on ties."""
    rng = np.random.RandomState(0)
    f32 = np.float32
    dts = [f32(3 ** 0.5 / 1024), f32(3 ** 0.5 / 512), f32(3 ** 0.5 / 100), f32(0.001953125), f32(1 / 3), f32(1e-3)]
    n_ok = n_fb = n_cross = 0
    for trial in range(6000):
        dt = dts[trial % len(dts)] if trial % 3 else f32(10 ** rng.uniform(-4, -1))
        t0 = f32(10 ** rng.uniform(-2.2, 1.7)) if trial % 5 else f32(2.0 ** rng.randint(-6, 5) - rng.randint(0, 40) * float(dt))
        if not t0 > 0:
            continue
        got = P.tile_closed_form(t0, dt)
        if got is None:
            n_fb += 1
            continue
        want = P.tile_chain(t0, dt)
        assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32)), (t0, dt)
        assert f32(got[1]).view(np.uint32) == f32(want[1]).view(np.uint32), (t0, dt)
        n_ok += 1
        n_cross += int(np.frexp(want[0][0])[1] != np.frexp(want[0][-1])[1])
    assert n_ok > 4000 and n_cross > 300 and n_fb > 0
    assert P.tile_closed_form(f32(1.0), f32(0.001953125)) is not None        # power-of-two step: remainder 0, no tie
    assert P.tile_closed_form(f32(0.001), f32(0.01)) is None                  # t below dt
