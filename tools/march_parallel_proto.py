"""Scratch (CPU, numpy): prototype of a WAVE-PARALLEL train march that is bit-identical to the reference's serial loop
(raymarching.cu:195-232), checked here against the oracle.

The serial loop visits a subsequence of ONE fixed sequence per ray, T[0] = t1 (jittered), T[j+1] = fl(T[j] +
calc_dt(T[j])): an occupied cell advances by one element, an empty cell by `do t += dt while t < t_target`, i.e.
to the first later element >= t_target.  So a wave can (1) generate 64 elements of the sequence (a chain of adds, no
memory), (2) evaluate all 64 candidates in parallel -- cell, occupancy bit, and for empty cells the skip target,
(3) find each candidate's successor index (j+1, or first j' > j with T[j'] >= t_target) and (4) mark the orbit of the
tile's entry index under `successor` (pointer jumping on the GPU, a plain loop here).  Samples = visited & occupied,
cut at max_samples.  A skip that leaves the tile carries its target into the next tile.

Candidates that the serial loop skips are NOT emitted even if their own cell test would say "occupied" (that can
happen next to a voxel face because t_target is computed in floating point) -- that is why step (4) is needed and a
plain "emit every occupied candidate" is not exact.
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pl_amd import synthetic as syn
from oracle.vren_oracle import Oracle

F = np.float32
SQRT3 = F(1.73205080757)
TILE = 64


def calc_dt(t, esf, max_samples, G, scale):
    lo = SQRT3 / F(max_samples); hi = SQRT3 * F(2) * F(scale) / F(G)
    return np.maximum(lo, np.minimum(t * F(esf), hi)).astype(F)       # helper_math.h clamp(f, a, b) = max(a, min(f, b)): lo wins if lo > hi


def expand_bits(v):
    v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
    v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
    v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
    v = (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
    return v


def candidates(T, o, d, d_inv, bf, cascades, G, scale, esf, max_samples):
    """Vectorised over the tile: occupancy bit and skip target of every candidate (march_step of the oracle)."""
    x = (T * d[0]).astype(F) + o[0]; y = (T * d[1]).astype(F) + o[1]; z = (T * d[2]).astype(F) + o[2]     # no FMA (nofma oracle)
    dt = calc_dt(T, esf, max_samples, G, scale)
    mx = np.maximum(np.abs(x), np.maximum(np.abs(y), np.abs(z)))
    mip_pos = np.clip(np.frexp(mx)[1] + 1, 0, cascades - 1)
    mip_dt = np.clip(np.frexp((dt * F(G)).astype(F))[1], 0, cascades - 1)
    mip = np.maximum(mip_pos, mip_dt)
    bound = np.minimum(np.ldexp(F(1), mip - 1).astype(F), F(scale)); binv = (F(1) / bound).astype(F)

    def cell(c):
        v = (F(0.5) * ((c * binv).astype(F) + F(1))).astype(F) * F(G)
        return np.clip(v.astype(F), F(0), F(G - 1)).astype(np.int32)
    nx, ny, nz = cell(x), cell(y), cell(z)
    idx = mip.astype(np.uint32) * np.uint32(G ** 3) + (expand_bits(nx.astype(np.uint32)) | (expand_bits(ny.astype(np.uint32)) << 1) |
                                                       (expand_bits(nz.astype(np.uint32)) << 2))
    occ = (bf[idx // 8] & (1 << (idx % 8)).astype(np.uint8)) != 0
    ginv = F(1) / F(G)

    def exit_t(n, c, dc, dci):
        sgn = F(np.copysign(1.0, dc))
        face = ((((n.astype(F) + F(0.5) + F(0.5) * sgn).astype(F) * ginv).astype(F) * F(2) - F(1)).astype(F) * bound).astype(F)
        return ((face - c).astype(F) * dci).astype(F)
    tt = np.minimum(exit_t(nx, x, d[0], d_inv[0]), np.minimum(exit_t(ny, y, d[1], d_inv[1]), exit_t(nz, z, d[2], d_inv[2])))
    target = (T + np.maximum(F(0), tt)).astype(F)
    return occ, target, dt


def march_ray(o, d, t1, t2, noise, bf, cascades, G, scale, esf, max_samples, naive=False):
    """naive=True: emit EVERY occupied candidate (no successor chase) -- to count how often that differs."""
    d_inv = (F(1) / d).astype(F)
    if not t1 >= 0:
        return []
    t1 = F(t1 + F(calc_dt(F(t1), esf, max_samples, G, scale) * noise))
    out, start, pending = [], F(t1), None
    while True:
        T = np.empty(TILE + 1, F); T[0] = start                     # (1) the chain
        for j in range(TILE):
            T[j + 1] = T[j] + calc_dt(T[j], esf, max_samples, G, scale)
        occ, target, _ = candidates(T[:TILE], o, d, d_inv, bf, cascades, G, scale, esf, max_samples)     # (2)
        succ = np.arange(1, TILE + 1)                                # (3) successor indices
        for j in np.nonzero(~occ)[0]:
            later = np.nonzero(T[j + 1:TILE + 1] >= target[j])[0]
            succ[j] = j + 1 + later[0] if len(later) else TILE + 1    # TILE+1: beyond T[TILE], carry the target
        v = 0 if pending is None else int(np.searchsorted(T, pending, side="left"))     # tile entry: first T >= carried target
        carry = pending if v > TILE else None
        while v < TILE:                                              # (4) orbit of the entry index
            if not (0 <= T[v] < t2) or len(out) >= max_samples:
                return out
            if occ[v]:
                out.append(T[v])
            elif succ[v] > TILE:
                carry = target[v]
            v = v + 1 if naive else succ[v]
        if v == TILE and not (T[TILE] < t2):
            return out
        start, pending = T[TILE], carry



def tile_closed_form(t_start, dt):
    """numpy restatement of lattice_tile_const_dt (csrc/march.hip): the 64 elements T[j] = fl(T[j-1] + dt) of a tile of the
    constant-step lattice and T[64], WITHOUT the chain of adds -- inside a binade every step adds the same integer number of
    ulps.  Returns None where the kernel falls back to the chain (ties, t below dt, zero/subnormal t)."""
    f32 = np.float32
    db = int(np.array(dt, f32).view(np.uint32)); ed = ((db >> 23) & 0xff) - 127; md = (db & 0x7fffff) | 0x800000
    if ((db >> 23) & 0xff) == 0:
        return None
    seg_bits = int(np.array(t_start, f32).view(np.uint32)); seg_j = 0
    mine = np.full(64, t_start, f32)
    lane = np.arange(64)
    while seg_j < 64:
        be = (seg_bits >> 23) & 0xff; s = (be - 127) - ed
        if be == 0 or be == 255 or (seg_bits >> 31) or s < 1 or s > 23:
            return None
        rem = md & ((1 << s) - 1); half = 1 << (s - 1)
        if rem == half:
            return None
        delta = (md >> s) + (1 if rem > half else 0)
        m0 = (seg_bits & 0x7fffff) | 0x800000
        mj = m0 + (lane - seg_j) * delta
        inn = (lane >= seg_j) & (mj < 0x1000000)
        vals = (((be << 23) | (mj & 0x7fffff)).astype(np.uint32)).view(f32)
        mine = np.where(inn, vals, mine)
        jc = seg_j + int(inn.sum())
        nxt = f32(mine[jc - 1] + f32(dt))
        if jc >= 64:
            return mine, nxt
        seg_bits = int(np.array(nxt, f32).view(np.uint32)); seg_j = jc


def tile_chain(t_start, dt):
    f32 = np.float32
    out = np.empty(65, f32); t = f32(t_start)
    for j in range(65):
        out[j] = t; t = f32(t + f32(dt))
    return out[:64], out[64]


def main(n_rays=1500):
    o = Oracle(fma=False)
    for cascades, scale, esf, fill in ((1, 0.5, 0.0, 0.1), (3, 2.0, 1 / 256, 0.2)):
        rng = np.random.default_rng(3)
        bf = syn.random_blob_bitfield(cascades, 128, fill, seed=22)
        ro = (rng.random((n_rays, 3), dtype=F) - F(0.5)) * F(3 * scale)
        rd = rng.standard_normal((n_rays, 3)).astype(F); rd /= np.linalg.norm(rd, axis=1, keepdims=True)
        _, hits, _ = o.ray_aabb_intersect(ro, rd, np.zeros((1, 3), F), np.full((1, 3), scale, F), 1)
        hits_t = hits[:, 0].copy(); near = (hits_t[:, 0] >= 0) & (hits_t[:, 0] < 0.01); hits_t[near, 0] = 0.01
        noise = rng.random(n_rays, dtype=F)
        rays_a, _, _, _, ts, _ = o.raymarching_train(ro, rd, hits_t, bf, cascades, scale, esf, noise, 128, 1024)
        bad = bad_naive = 0
        for r in range(n_rays):
            row = rays_a[rays_a[:, 0] == r][0]
            want = ts[row[1]:row[1] + row[2]]
            for naive in (False, True):
                got = np.array(march_ray(ro[r], rd[r], hits_t[r, 0], hits_t[r, 1], noise[r], bf, cascades, 128, scale, esf, 1024, naive), F)
                if got.shape != want.shape or not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                    bad += not naive; bad_naive += naive
        print("cascades %d scale %.1f esf %.4f: %d rays, %d samples; rays that differ from the serial loop: %d (successor chase), %d (every occupied candidate)"
              % (cascades, scale, esf, n_rays, len(ts), bad, bad_naive))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1500)
