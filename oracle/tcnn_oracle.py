"""TEST INFRASTRUCTURE -- torch-CPU restatement of the tiny-cuda-nn pieces of the hot path.

PARITY UNPINNED: tiny-cuda-nn is an un-vendored pip dependency of the reference
(/root/reference/README.md:39, no version pinned; call sites models/networks.py:36-92) and is not
installable here (CUDA only, no network).  The reference has no tests or golden vectors for it.
This file restates the PUBLISHED algorithm of NVlabs/tiny-cuda-nn:
  * include/tiny-cuda-nn/encodings/grid.h: grid_scale, grid_resolution, pos_fract (linear),
    grid_index, coherent prime hash (primes 1, 2654435761, 805459861), kernel_grid,
    per-level offset table of GridEncodingTemplated;
  * include/tiny-cuda-nn/encodings/spherical_harmonics.h (degree 4);
  * src/fully_fused_mlp.cu semantics: no bias, ReLU, weights (out,in) row-major, output padded to
    a multiple of 16, f16 activations between layers;
  * bindings/torch/tinycudann/modules.py: f32 master params, f16 outputs, loss scale 128.
and anchors it on the reference's own configuration (L=16, F=2, T=2^19, N_min=16,
b = exp(log(2048*scale/16)/15); density net 32->64->16, rgb net 32->64->64->3(16)).

It is plain fp32 torch with autograd (gradients come for free); `quantize=True` inserts the f16
rounding points the native kernels have, so the comparison tolerance only has to absorb
summation order.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.
"""
import math

import numpy as np
import torch

PRIMES = (1, 2654435761, 805459861)


class GridMeta:
    def __init__(self, n_levels=16, n_features=2, log2_hashmap_size=19, base_resolution=16, per_level_scale=2.0, exact=False):
        """Offset table as GridEncodingTemplated's constructor builds it, in float32 arithmetic.

        log2f / exp2f are taken correctly rounded (double evaluation rounded to float), which is
        what glibc's versions deliver on the host where tiny-cuda-nn builds this table.
        exact=True: the same formula in exact arithmetic (the product's level_table="exact" opt-in)."""
        f32 = np.float32
        self.n_levels, self.n_features = n_levels, n_features
        log2_pls = f32(math.log2(float(f32(per_level_scale))))
        self.scale, self.resolution, self.offset = [], [], [0]
        for l in range(n_levels):
            e = f32(f32(l) * log2_pls)
            scale = f32(f32(f32(2.0 ** float(e)) * f32(base_resolution)) - f32(1.0))
            if exact:
                scale = f32(round(2.0 ** (l * math.log2(per_level_scale)) * base_resolution - 1.0, 9))
            res = int(math.ceil(float(scale))) + 1
            n = min(res ** 3, (2 ** 32 - 1) // 2)
            n = (n + 7) // 8 * 8
            n = min(n, 1 << log2_hashmap_size)
            self.scale.append(float(scale)); self.resolution.append(res); self.offset.append(self.offset[-1] + n)
        self.total = self.offset[-1]

    def level_is_hashed(self, l):
        size = self.offset[l + 1] - self.offset[l]
        stride, res = 1, self.resolution[l]
        for _ in range(3):
            if stride > size:
                break
            stride *= res
        return size < stride


def _corner_indices(meta, l, pg):
    """pg (S,3) int64 cell coords -> list of 8 (S,) int64 table indices (within the level)."""
    res = meta.resolution[l]
    size = meta.offset[l + 1] - meta.offset[l]
    hashed = meta.level_is_hashed(l)
    M = 0xFFFFFFFF
    out = []
    for c in range(8):
        cx = (pg[:, 0] + (c & 1)) & M
        cy = (pg[:, 1] + ((c >> 1) & 1)) & M
        cz = (pg[:, 2] + (c >> 2)) & M
        if hashed:
            idx = ((cx * PRIMES[0]) & M) ^ ((cy * PRIMES[1]) & M) ^ ((cz * PRIMES[2]) & M)
        else:
            idx = (cx + ((cy * res) & M) + ((cz * ((res * res) & M)) & M)) & M
        out.append(idx % size)
    return out


def hash_encode(x01, table, meta, quantize=False):
    """kernel_grid of grid.h.  x01 (S,3) f32 in [0,1]; table (total, F) float tensor (may require
    grad).  Returns (S, L*F) f32, feature index l*F + f."""
    x01 = x01.float()
    feats = []
    for l in range(meta.n_levels):
        scale = meta.scale[l]
        pos = x01 * scale + 0.5                       # pos_fract
        fl = torch.floor(pos)
        w = pos - fl
        pg = fl.to(torch.int64)
        idxs = _corner_indices(meta, l, pg)
        acc = 0
        for c in range(8):
            wx = w[:, 0] if (c & 1) else 1 - w[:, 0]
            wy = w[:, 1] if ((c >> 1) & 1) else 1 - w[:, 1]
            wz = w[:, 2] if (c >> 2) else 1 - w[:, 2]
            val = table[meta.offset[l] + idxs[c]].float()
            acc = acc + (wx * wy * wz)[:, None] * val
        feats.append(acc)
    out = torch.cat(feats, 1)
    return q16(out) if quantize else out


def q16(t):
    """Round through f16 (straight-through for autograd)."""
    return t + (t.half().float() - t).detach()


SH_C = [0.28209479177387814, 0.48860251190291987, 1.0925484305920792, 0.94617469575755997, 0.31539156525251999,
        0.54627421529603959, 0.59004358992664352, 2.8906114426405538, 0.45704579946446572, 0.3731763325901154,
        1.4453057213202769]


def sh4(d):
    """Degree-4 real SH of unit vectors d (S,3) -> (S,16); spherical_harmonics.h, SURVEY.md 8a."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    return torch.stack([
        torch.full_like(x, 0.28209479177387814),
        -0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x,
        1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.94617469575755997 * z2 - 0.31539156525251999,
        -1.0925484305920792 * xz, 0.54627421529603959 * x2 - 0.54627421529603959 * y2,
        0.59004358992664352 * y * (-3.0 * x2 + y2), 2.8906114426405538 * xy * z,
        0.45704579946446572 * y * (1.0 - 5.0 * z2), 0.3731763325901154 * z * (5.0 * z2 - 3.0),
        0.45704579946446572 * x * (1.0 - 5.0 * z2), 1.4453057213202769 * z * (x2 - y2),
        0.59004358992664352 * x * (-x2 + 3.0 * y2)], 1)


def split_mlp_params(params, n_in, n_hidden, n_out_padded=16, width=64):
    """tiny-cuda-nn parameter order: layers in order, each (out,in) row-major."""
    ws, off = [], 0
    dims = [n_in] + [width] * n_hidden + [n_out_padded]
    for i in range(len(dims) - 1):
        n = dims[i + 1] * dims[i]
        ws.append(params[off:off + n].view(dims[i + 1], dims[i]))
        off += n
    assert off == params.numel()
    return ws


def matmul_acc16(h, w):
    """h (S,K) @ w (O,K)^T with an F16 ACCUMULATOR, as tiny-cuda-nn's fully-fused MLP runs it (wmma fragments
    `accumulator, 16, 16, 16, __half`): the K dimension is consumed in 16-wide MMA steps; inside a step the products are
    summed at higher precision, the running sum is rounded to f16 after every step.  (The native kernels here accumulate
    in f32 over the whole K -- wider; this mode bounds how far real tiny-cuda-nn output sits from both.)"""
    acc = torch.zeros(h.shape[0], w.shape[0])
    for k0 in range(0, h.shape[1], 16):
        acc = (acc + h[:, k0:k0 + 16] @ w[:, k0:k0 + 16].t()).half().float()
    return acc


def mlp(x, params, n_in, n_hidden, n_out, out_act="None", quantize=False, acc16=False):
    """FullyFusedMLP: ReLU hidden, no bias.  Returns (S, n_out) PRE-rounding f32 (after out_act).
    acc16 (forward only, implies quantize): f16 accumulators, see matmul_acc16."""
    ws = split_mlp_params(params, n_in, n_hidden)
    quantize = quantize or acc16
    h = q16(x.float()) if quantize else x.float()
    for i, w in enumerate(ws):
        w = q16(w.float()) if quantize else w.float()
        h = matmul_acc16(h, w) if acc16 else h @ w.t()
        if i < len(ws) - 1:
            h = torch.relu(h)
            if quantize:
                h = q16(h)
    h = h[:, :n_out]
    if out_act == "Sigmoid":
        h = torch.sigmoid(h)
    return h


class TruncExp(torch.autograd.Function):
    """custom_functions.py:162-173."""
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        x = ctx.saved_tensors[0]
        return g * torch.exp(x.clamp(-15, 15))


class Field:
    """NGP.forward / NGP.density (networks.py:94-107,132-153) with the tiny-cuda-nn parts above."""

    def __init__(self, scale=0.5, seed=1337, exact_levels=False):
        self.scale = scale
        b = math.exp(math.log(2048 * scale / 16) / 15)
        self.meta = GridMeta(16, 2, 19, 16, b, exact=exact_levels)
        g = torch.Generator().manual_seed(seed)
        # tiny-cuda-nn init: xavier-uniform MLP weights, U(-1e-4,1e-4) grid
        def xavier(o, i):
            a = math.sqrt(6.0 / (i + o))
            return (torch.rand(o, i, generator=g) * 2 - 1) * a
        self.density_w = torch.cat([xavier(64, 32).reshape(-1), xavier(16, 64).reshape(-1)])
        self.rgb_w = torch.cat([xavier(64, 32).reshape(-1), xavier(64, 64).reshape(-1), xavier(16, 64).reshape(-1)])
        self.table = (torch.rand(self.meta.total, 2, generator=g) * 2 - 1) * 1e-4

    def parameters(self):
        return [self.density_w, self.rgb_w, self.table]

    def density(self, x, quantize=False, acc16=False):
        quantize = quantize or acc16
        x01 = (x - (-self.scale)) / (self.scale - (-self.scale))          # networks.py:103
        table = q16(self.table) if quantize else self.table
        feats = hash_encode(x01, table, self.meta, quantize)
        h = mlp(feats, self.density_w, 32, 1, 16, "None", quantize, acc16)
        if quantize:
            h = q16(h)
        sigma = TruncExp.apply(h[:, 0])                                    # networks.py:105
        return sigma, h, feats

    def forward(self, x, d, quantize=False, acc16=False):
        quantize = quantize or acc16
        sigma, h, _ = self.density(x, quantize, acc16)
        dn = d / torch.norm(d, dim=1, keepdim=True)                        # networks.py:143
        sh = sh4(dn)                                                       # (d+1)/2 then *2-1 inside tcnn == identity
        if quantize:
            sh = q16(sh)
        rgb = mlp(torch.cat([sh, h], 1), self.rgb_w, 32, 2, 3, "Sigmoid", quantize, acc16)
        if quantize:
            rgb = q16(rgb)
        return sigma, rgb, h
