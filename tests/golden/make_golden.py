"""Generates tests/golden/vren_golden.npz from the REFERENCE's own kernels compiled for the CPU
(oracle/_ref/libvren_ref_fma.so, built by oracle/build_ref.sh from /root/reference/models/csrc).
Run in the development container (where /root/reference exists):  python tests/golden/make_golden.py
The fixture travels with the repo so the oracle stays pinned where the reference is absent."""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ngp_pl_amd import synthetic as syn          # noqa: E402
from oracle.vren_oracle import Reference          # noqa: E402
from tests.helpers import aabb_hits, make_rays    # noqa: E402


def origins(ro, tag, scale):
    if tag == "garden":
        f = np.random.RandomState(26).choice([1.0, 2.0, 3.0, 5.0, 8.0], ro.shape[0]).astype(np.float32)
        return (ro * f[:, None]).astype(np.float32)
    return ro * (1.5 if scale > 0.5 else 1.0)


def main():
    r = Reference(fma=True)
    out = {}
    n = 192
    ro, rd = make_rays(n, seed=21, W=64, n_cams=4)
    out["rays_o"], out["rays_d"] = ro, rd
    # "garden": the mip-NeRF360 recipe (benchmark_mipnerf360.sh:21-24: scale 16 -> 6 cascades, exp_step_factor 1/256), camera
    # radii 1.5..12 so that every cascade is sampled
    for tag, cascades, scale, esf, fill in (("syn", 1, 0.5, 0.0, 0.1), ("real", 3, 2.0, 1 / 256, 0.2), ("garden", 6, 16.0, 1 / 256, 0.12)):
        rr = origins(ro, tag, scale)
        bf = syn.random_blob_bitfield(cascades, 128, fill, seed=22)
        ht = aabb_hits(r, rr, rd, scale)
        noise = np.random.RandomState(23).rand(n).astype(np.float32)
        rays_a, xyzs, dirs, deltas, ts, counter = r.raymarching_train(rr, rd, ht, bf, cascades, scale, esf, noise, 128, 1024)
        if tag == "garden":
            out[tag + "_bitfield_crc"] = np.array([zlib.crc32(bf.tobytes())], np.int64)     # 1.5 MB: the generator is deterministic, pin its checksum
        else:
            out[tag + "_bitfield_packed"] = np.packbits(np.unpackbits(bf))      # as is (uint8)
        out[tag + "_hits_t"], out[tag + "_noise"] = ht, noise
        out[tag + "_rays_a"], out[tag + "_xyzs"], out[tag + "_deltas"], out[tag + "_ts"] = rays_a, xyzs, deltas, ts
        h2 = ht.copy()
        alive = np.arange(n, dtype=np.int64)
        x2, d2, de2, t2, ne = r.raymarching_test(rr, rd, h2, alive, bf, cascades, scale, esf, 128, 1024, 4)
        out[tag + "_test_ts"], out[tag + "_test_deltas"], out[tag + "_test_neff"], out[tag + "_test_hits_after"] = t2, de2, ne, h2
        if tag == "syn":
            g = np.random.RandomState(24)
            S = ts.shape[0]
            sig = (g.rand(S).astype(np.float32) ** 3) * 400; rgbs = g.rand(S, 3).astype(np.float32)
            total, op, depth, rgb, ws = r.composite_train_fw(sig, rgbs, deltas, ts, rays_a, 1e-4)
            dO, dD, dC, dW = g.randn(n).astype(np.float32), g.randn(n).astype(np.float32), g.randn(n, 3).astype(np.float32), g.randn(S).astype(np.float32)
            dsig, drgbs = r.composite_train_bw(dO, dD, dC, dW, sig, rgbs, ws, deltas, ts, rays_a, op, depth, rgb, 1e-4)
            loss, wi, wti = r.distortion_loss_fw(ws, deltas, ts, rays_a)
            out.update(sigmas=sig, rgbs=rgbs, total=total, opacity=op, depth=depth, rgb=rgb, ws=ws, dO=dO, dD=dD, dC=dC, dW=dW,
                       dsig=dsig, drgbs=drgbs, dist_loss=loss, dist_bw=r.distortion_loss_bw(dO, wi, wti, ws, deltas, ts, rays_a))
    coords = np.random.RandomState(25).randint(0, 128, (512, 3)).astype(np.int32)
    out["morton_coords"], out["morton_idx"] = coords, r.morton3D(coords)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "vren_golden.npz"), **out)
    print("wrote", {k: v.shape for k, v in out.items() if k.endswith(("ts", "rays_a"))})


if __name__ == "__main__":
    main()
