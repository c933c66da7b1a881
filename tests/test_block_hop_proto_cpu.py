"""CPU: the frame marcher's block hop (csrc/march.hip: march_probe crosses an 8^3-cell block without an occupied cell in one hop when
no lattice point lies within the rounding slack of the block's exit time) emits exactly the samples of the reference's cell-by-cell
walk (raymarching.cu:357-401).  tools/block_hop_proto.c restates both walks float for float; here its cell-by-cell walk is pinned to
the oracle's raymarching_test, the two walks are compared over random and adversarial rays (aimed at block / cell corners, edges and
faces a hair off, nearly axis-parallel, cameras inside the box), and the same comparison WITHOUT the slack rule is shown to differ --
the rule is what makes the hop exact, and this test can tell."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from ngp_pl_amd import synthetic as syn
from oracle.vren_oracle import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = np.float32


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("block_hop") / "libblock_hop.so")
    subprocess.run(["gcc", "-O2", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-std=c11", "-Wall", "-Werror", "-shared",
                    os.path.join(ROOT, "tools", "block_hop_proto.c"), "-o", so, "-lm"], check=True)
    L = C.CDLL(so)
    L.block_hop_compare.restype = C.c_longlong
    L.block_hop_compare.argtypes = [C.c_void_p] * 4 + [C.c_longlong, C.c_float, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    L.block_hop_walk.restype = None
    L.block_hop_walk.argtypes = [C.c_void_p] * 4 + [C.c_longlong, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return L


def _bitfield(kind, seed, orc):
    rng = np.random.default_rng(seed)
    if kind == "blobs":
        return syn.random_blob_bitfield(1, 128, 0.03, seed=seed)
    if kind == "dense_blobs":
        return syn.random_blob_bitfield(1, 128, 0.15, seed=seed)
    bits = np.zeros(128 ** 3, np.uint8)
    if kind == "scattered":
        bits[rng.integers(0, 128 ** 3, 300)] = 1
    else:                                              # a hollow box whose faces lie ON block boundaries (cells 40..87 = blocks 5..10)
        g = np.zeros((128, 128, 128), np.uint8); g[40:88, 40:88, 40:88] = 1; g[44:84, 44:84, 44:84] = 0
        z, y, x = np.nonzero(g)
        bits[orc.morton3D(np.stack([x, y, z], 1).astype(np.int32))] = 1
    return np.packbits(bits, bitorder="little")


def _rays(n, rng):
    ro = rng.standard_normal((n, 3)); ro = ro / np.linalg.norm(ro, axis=1, keepdims=True) * rng.uniform(0.9, 4.0, (n, 1))
    ro[: n // 8] = rng.uniform(-0.45, 0.45, (n // 8, 3))
    target = rng.uniform(-0.5, 0.5, (n, 3))
    m = n // 2
    pt = np.where(rng.random((m, 1)) < 0.5, rng.integers(0, 17, (m, 3)) / 16.0 - 0.5, rng.integers(0, 129, (m, 3)) / 128.0 - 0.5)
    pt = np.where(rng.random((m, 3)) < 0.4, rng.uniform(-0.5, 0.5, (m, 3)), pt)          # free coordinates: edges and faces, not only corners
    target[:m] = pt + rng.choice([-1e-3, -1e-5, -1e-7, 0, 0, 1e-7, 1e-5, 1e-3], (m, 3))
    rd = target - ro; rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    ax = rng.integers(0, n, n // 40)
    rd[ax] = np.eye(3)[rng.integers(0, 3, len(ax))] * rng.choice([-1, 1], (len(ax), 1)) \
        + rng.choice([0, 0, 1e-7, 1e-5, 1e-3], (len(ax), 3)) * rng.standard_normal((len(ax), 3))
    rd[ax] /= np.linalg.norm(rd[ax], axis=1, keepdims=True)
    return ro.astype(F), rd.astype(F)


def _hits(orc, ro, rd):
    _, hits, _ = orc.ray_aabb_intersect(ro, rd, np.zeros((1, 3), F), np.full((1, 3), 0.5, F), 1)
    h = hits[:, 0].copy()
    ok = h[:, 0] >= 0
    ro, rd, h = np.ascontiguousarray(ro[ok]), np.ascontiguousarray(rd[ok]), np.ascontiguousarray(h[ok])
    h[h[:, 0] < 0.01, 0] = 0.01
    return ro, rd, h


def test_the_restated_cell_walk_is_the_oracles(lib):
    orc = Oracle(fma=True)
    rng = np.random.default_rng(1)
    bf = _bitfield("blobs", 3, orc)
    ro, rd, h = _hits(orc, *_rays(4000, rng))
    n, cap = len(ro), 1024
    ts = np.zeros((n, cap), F); counts = np.zeros(n, np.int32)
    lib.block_hop_walk(bf.ctypes.data, ro.ctypes.data, rd.ctypes.data, h.ctypes.data, n, 0.5, 1024, cap, ts.ctypes.data, counts.ctypes.data)
    want = orc.raymarching_test(ro, rd, h.copy(), np.arange(n, dtype=np.int64), bf, 1, 0.5, 0.0, 128, 1024, cap)
    assert np.array_equal(counts, want[4]) and counts.sum() > 20000
    assert np.array_equal(ts.view(np.uint32), want[3].view(np.uint32))


@pytest.mark.parametrize("kind", ["blobs", "scattered", "hollow_box_on_block_faces", "dense_blobs"])
def test_block_hops_emit_the_cell_walks_samples(lib, kind):
    orc = Oracle(fma=True)
    bad_without_rule = 0
    for seed in (0, 1):
        rng = np.random.default_rng(40 + seed)
        bf = _bitfield(kind, seed, orc)
        ro, rd, h = _hits(orc, *_rays(150000, rng))
        for slack_scale in (1.0, 0.0):
            stats = np.zeros(5, np.int64); first = np.zeros(1, np.int64)
            bad = lib.block_hop_compare(bf.ctypes.data, ro.ctypes.data, rd.ctypes.data, h.ctypes.data, len(ro), 0.5, 1024, slack_scale,
                                        stats.ctypes.data, first.ctypes.data)
            if slack_scale == 1.0:
                assert bad == 0, (kind, seed, int(first[0]))
                assert stats[0] > 2 * stats[1] and stats[0] > 5e5           # the hop is taken far more often than declined
                assert stats[3] > 1000
            else:
                bad_without_rule += bad
            assert stats[4] == 0                   # the closed-form landing (one binade, constant ulps per step) is the chain of adds'
    if kind == "dense_blobs":
        assert bad_without_rule > 0           # without the slack rule the hop is NOT the walk (a handful of rays per million)


def test_the_restatement_and_the_kernel_use_the_same_rule():
    """The slack of a ray and the hop's acceptance test are written once in csrc/march.hip and once in tools/block_hop_proto.c: the two
    texts must stay the same expressions (a changed constant in one of them would leave this file testing something else)."""
    def squeeze(x):
        return "".join(x.split())
    kernel = squeeze(open(os.path.join(ROOT, "ngp_pl_amd", "csrc", "march.hip")).read())
    proto = squeeze(open(os.path.join(ROOT, "tools", "block_hop_proto.c")).read())
    slack = "8.0f*(fabsf(t2)*1.2e-7f+1.2e-7f+1.2e-7f*fmaxf(fabsf("
    assert slack in kernel and slack in proto
    for text in ("if(tt-tau>hop_slack&&tau-prev>hop_slack)", "if(k*delta<diff)++k;if(k*delta<diff)++k;if(k>1u&&(k-1u)*delta>=diff)--k;if(k==0u)k=1u;",
                 "tau-t<0.25f&&(ub>>23)==(tb>>23)&&sh>=1&&sh<=23"):
        assert text in kernel.replace("//k=thesmallestcountwithkdelta>=diff,atleast1:", "").replace("//thequotientaboveiswithinoneofit", ""), text
        assert text in proto, text
