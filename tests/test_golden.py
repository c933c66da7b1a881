"""CPU: the oracle against committed golden vectors produced by the reference's own kernels
(tests/golden/make_golden.py; they pin the oracle on machines without /root/reference), and
-m gpu: the HIP kernels against the same vectors."""
import os

import numpy as np
import pytest

from ngp_pl_amd import synthetic as syn
from oracle.vren_oracle import Oracle

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "vren_golden.npz"))


def same(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, what
    if a.dtype == np.float32:
        a, b = a.view(np.uint32), b.view(np.uint32)
    assert np.array_equal(a, b), what


CASES = {"syn": (1, 0.5, 0.0, 0.1), "real": (3, 2.0, 1 / 256, 0.2), "garden": (6, 16.0, 1 / 256, 0.12)}
TAGS = ["syn", "real", "garden"]


def origins(tag, scale):
    from tests.golden.make_golden import origins as f
    return f(G["rays_o"], tag, scale)


def bitfield(tag, cascades, fill):
    bf = syn.random_blob_bitfield(cascades, 128, fill, seed=22)
    if tag == "garden":
        import zlib
        assert zlib.crc32(bf.tobytes()) == int(G[tag + "_bitfield_crc"][0]), "bitfield generator drifted"
    else:
        same(bf, G[tag + "_bitfield_packed"], "bitfield generator drifted")
    return bf


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_marching_against_golden(tag):
    cascades, scale, esf, fill = CASES[tag]
    o = Oracle(fma=True)
    ro = origins(tag, scale); rd = G["rays_d"]
    bf = bitfield(tag, cascades, fill)
    rays_a, xyzs, dirs, deltas, ts, counter = o.raymarching_train(ro, rd, G[tag + "_hits_t"], bf, cascades, scale, esf, G[tag + "_noise"], 128, 1024)
    same(rays_a, G[tag + "_rays_a"], "rays_a"); same(xyzs, G[tag + "_xyzs"], "xyzs"); same(deltas, G[tag + "_deltas"], "deltas"); same(ts, G[tag + "_ts"], "ts")
    h = G[tag + "_hits_t"].copy()
    x2, d2, de2, t2, ne = o.raymarching_test(ro, rd, h, np.arange(ro.shape[0]), bf, cascades, scale, esf, 128, 1024, 4)
    same(t2, G[tag + "_test_ts"], "test ts"); same(ne, G[tag + "_test_neff"], "N_eff"); same(h, G[tag + "_test_hits_after"], "hits_t")


def test_oracle_composite_against_golden():
    o = Oracle(fma=True)
    total, op, depth, rgb, ws = o.composite_train_fw(G["sigmas"], G["rgbs"], G["syn_deltas"], G["syn_ts"], G["syn_rays_a"], 1e-4)
    same(total, G["total"], "total_samples")
    for a, k in ((op, "opacity"), (depth, "depth"), (rgb, "rgb"), (ws, "ws")):
        np.testing.assert_allclose(a, G[k], rtol=2e-5, atol=1e-6, err_msg=k)     # the golden build contracts mul+add (see test_oracle_vs_ref)
    dsig, drgbs = o.composite_train_bw(G["dO"], G["dD"], G["dC"], G["dW"], G["sigmas"], G["rgbs"], G["ws"], G["syn_deltas"], G["syn_ts"],
                                       G["syn_rays_a"], G["opacity"], G["depth"], G["rgb"], 1e-4)
    np.testing.assert_allclose(dsig, G["dsig"], rtol=1e-4, atol=1e-5 * np.abs(G["dsig"]).max())
    np.testing.assert_allclose(drgbs, G["drgbs"], rtol=2e-5, atol=1e-6)
    loss, wi, wti = o.distortion_loss_fw(G["ws"], G["syn_deltas"], G["syn_ts"], G["syn_rays_a"])
    np.testing.assert_allclose(loss, G["dist_loss"], rtol=1e-4, atol=1e-7)
    same(o.morton3D(G["morton_coords"]), G["morton_idx"], "morton")


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_hip_marching_against_golden(tag):
    import torch
    import ngp_pl_amd.vren as vren
    cascades, scale, esf, fill = CASES[tag]
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    ro = origins(tag, scale); rd = G["rays_d"]
    bf = bitfield(tag, cascades, fill)
    got = vren.raymarching_train(d(ro), d(rd), d(G[tag + "_hits_t"]), d(bf), cascades, scale, esf, d(G[tag + "_noise"]), 128, 1024)
    same(got[0].cpu().numpy(), G[tag + "_rays_a"], "rays_a"); same(got[1].cpu().numpy(), G[tag + "_xyzs"], "xyzs")
    same(got[3].cpu().numpy(), G[tag + "_deltas"], "deltas"); same(got[4].cpu().numpy(), G[tag + "_ts"], "ts")
    h = d(G[tag + "_hits_t"].copy())
    out = vren.raymarching_test(d(ro), d(rd), h, d(np.arange(ro.shape[0], dtype=np.int64)), d(bf), cascades, scale, esf, 128, 1024, 4)
    same(out[3].cpu().numpy(), G[tag + "_test_ts"], "test ts"); same(out[2].cpu().numpy(), G[tag + "_test_deltas"], "test deltas")
    same(out[4].cpu().numpy(), G[tag + "_test_neff"], "N_eff"); same(h.cpu().numpy(), G[tag + "_test_hits_after"], "hits_t after")
    if tag == "syn":
        out = vren.composite_train_fw(d(G["sigmas"]), d(G["rgbs"]), d(G["syn_deltas"]), d(G["syn_ts"]), d(G["syn_rays_a"]), 1e-4)
        np.testing.assert_allclose(out[3].cpu().numpy(), G["rgb"], rtol=0, atol=1e-5)      # north star: RGB within 1e-4 abs
        np.testing.assert_allclose(out[1].cpu().numpy(), G["opacity"], rtol=0, atol=1e-5)
