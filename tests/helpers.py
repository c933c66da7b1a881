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
