"""Shared deterministic inputs for the parity tests."""
import numpy as np
import torch

from ngp_pl_amd import synthetic as syn


def make_rays(n_rays, seed=0, W=400, n_cams=8, include_misses=True):
    """Random pixels of random hemisphere cameras -> rays_o, rays_d (n,3) float32 numpy."""
    g = np.random.RandomState(seed)
    K = syn.intrinsics(W)
    dirs = syn.get_ray_directions(W, W, K)
    poses = syn.hemisphere_poses(n_cams, seed=seed)
    img = torch.from_numpy(g.randint(0, n_cams, n_rays))
    pix = torch.from_numpy(g.randint(0, W * W, n_rays))
    ro, rd = syn.get_rays(dirs[pix], poses[img])
    ro, rd = ro.numpy().copy(), rd.numpy().copy()
    if include_misses and n_rays >= 8:   # a few rays that miss the box, one axis-parallel ray
        rd[0] = [0.0, 0.0, 1.0]; ro[0] = [0.1, 0.1, -2.0]
        rd[1] = -rd[1]
        ro[2] = [0.0, 0.0, 0.0]          # origin inside the box
    return ro.astype(np.float32), rd.astype(np.float32)


def aabb_hits(oracle, ro, rd, scale=0.5, near=0.01):
    """render() prologue (rendering.py:27-29) with the oracle."""
    c = np.zeros((1, 3), np.float32); h = np.full((1, 3), scale, np.float32)
    _, hits_t, _ = oracle.ray_aabb_intersect(ro, rd, c, h, 1)
    ht = hits_t[:, 0].copy()
    m = (ht[:, 0] >= 0) & (ht[:, 0] < near)
    ht[m, 0] = near
    return ht


def error_distribution(name, got, want, log=None):
    """Per-row max-abs error between two arrays (rows = rays) as a distribution: mean / median / q90 / q99 / q999 / max.  Printed (pytest -s
    shows it) and, with NGP_PARITY_LOG set, appended to that file -- profiles/r06_parity_distribution.txt is such a log from an MI355X,
    and the tolerances of the end-to-end parity tests are 3 x its figures."""
    import os
    want = np.asarray(want, np.float64)
    err = np.abs(np.asarray(got, np.float64) - want).reshape(len(want), -1).max(1)
    d = {"n": int(len(err)), "mean": float(err.mean()), "median": float(np.median(err)), "q90": float(np.quantile(err, 0.90)),
         "q99": float(np.quantile(err, 0.99)), "q999": float(np.quantile(err, 0.999)), "max": float(err.max())}
    line = "%-52s n %6d  mean %.3e  median %.3e  q90 %.3e  q99 %.3e  q999 %.3e  max %.3e" % (
        name, d["n"], d["mean"], d["median"], d["q90"], d["q99"], d["q999"], d["max"])
    print(line)
    path = log or os.environ.get("NGP_PARITY_LOG")
    if path:
        with open(path, "a") as f:
            f.write(line + "\n")
    return err, d
