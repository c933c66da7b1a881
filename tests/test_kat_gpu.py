"""GPU: the product's kernels against known answers that do NOT come from oracle/tcnn_oracle.py (tests/kat_independent.py: scipy's
spherical harmonics, hand-computed hash indices).  The tiny-cuda-nn half of the oracle stays "parity unpinned" (no fixture of the real
package exists); these pin a8 (SH degree 4) and the hash-grid geometry of a6 to an independent source."""
import ctypes as C

import numpy as np
import pytest
import torch

import kat_independent as K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available()
    from ngp_pl_amd import _lib
    return _lib


def test_sh4_kernel_against_scipy(lib):
    """ngp_sh4_fwd (f16 out) on 10 000 directions incl. poles and axes vs scipy.special: within half an f16 ulp of the exact value
    (+ 2e-4 for the kernel's f32 polynomial evaluation of inputs passed through the (d+1)/2 -> *2-1 round trip, networks.py:143-144)."""
    d = K.unit_directions(10000, seed=3)
    want = K.sh4_scipy(d)
    d01 = ((torch.from_numpy(d).float() + 1) / 2).cuda().contiguous()
    out = torch.empty(d.shape[0], 16, dtype=torch.float16, device="cuda")
    lib.call("ngp_sh4_fwd", lib.ptr(d01), d.shape[0], lib.ptr(out), lib.stream())
    got = out.float().cpu().numpy().astype(np.float64)
    ulp = np.maximum(2.0 ** (np.floor(np.log2(np.maximum(np.abs(want), 2.0 ** -14))) - 10), 2.0 ** -24)      # f16 spacing at the exact value
    err = np.abs(got - want)
    assert (err <= 0.5 * ulp + 2e-4).all(), (err.max(), np.unravel_index(np.argmax(err - 0.5 * ulp), err.shape))
    # and through the module the reference instantiates (tcnn.Encoding, SphericalHarmonics degree 4)
    from ngp_pl_amd import tcnn
    enc = tcnn.Encoding(3, {"otype": "SphericalHarmonics", "degree": 4})
    got2 = enc(d01).float().cpu().numpy()
    np.testing.assert_array_equal(got2, out.float().cpu().numpy())


def test_hash_grid_gathers_the_hand_computed_entries(lib):
    """ngp_hashgrid_fwd at level 15 (res 1025, hashed, 2^19 entries): a position just inside cell (x, y, z) weights corner 0 with ~1;
    with a distinctive value at the hand-computed index of that cell and zeros elsewhere the level's feature must be that value."""
    from ngp_pl_amd import tcnn
    meta = tcnn.make_grid_meta({"otype": "Grid", "type": "Hash", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19,
                                "base_resolution": 16, "per_level_scale": K.per_level_scale(0.5), "interpolation": "Linear"})
    assert [meta.offset[i] for i in range(17)] == K.LEVELS[(0.5, "float32")]["offset"]
    scale15 = float(meta.scale[15])
    cells = [c for c, _ in K.HASH_KAT if max(c) <= 1023]
    table = torch.zeros(meta.offset[16], 2, dtype=torch.float16)
    for k, (c, idx) in enumerate(K.HASH_KAT):
        if max(c) <= 1023:
            table[meta.offset[15] + idx] = torch.tensor([1.0 + k / 16.0, -(1.0 + k / 16.0)], dtype=torch.float16)
    # pos = x01 * scale + 0.5 -> cell + 0.002 on every axis; x = x01 - 0.5 in the [-0.5, 0.5] box
    x01 = torch.tensor([[(c[0] + 0.002 - 0.5) / scale15, (c[1] + 0.002 - 0.5) / scale15, (c[2] + 0.002 - 0.5) / scale15] for c in cells], dtype=torch.float64)
    x = (x01 - 0.5).float().cuda().contiguous()
    n = x.shape[0]
    feats = torch.empty(16, n, 2, dtype=torch.float16, device="cuda")
    mn = torch.full((3,), -0.5, device="cuda"); mx = torch.full((3,), 0.5, device="cuda")
    lib.call("ngp_hashgrid_fwd", lib.ptr(x), lib.ptr(mn), lib.ptr(mx), lib.ptr(table.cuda()), C.byref(meta), n, lib.ptr(feats), lib.stream())
    got = feats[15].float().cpu().numpy()
    j = 0
    for k, (c, idx) in enumerate(K.HASH_KAT):
        if max(c) > 1023:
            continue
        v = 1.0 + k / 16.0
        # corner-0 weight (1 - f)^3 with f within [0, 0.01] (float32 rounding of x01 * scale at 1023 is ~6e-5 per axis)
        assert 0.96 * v <= got[j, 0] <= v + 1e-3 and -v - 1e-3 <= got[j, 1] <= -0.96 * v, (c, idx, got[j])
        j += 1


@pytest.mark.parametrize("n_hidden", [1, 2])
def test_fully_fused_mlp_with_hand_computable_weights(lib, n_hidden):
    """tcnn.Network (FullyFusedMLP, 64 neurons, ReLU, no bias, weights (out, in) row-major layer after layer, output rows padded to 16
    -- models/networks.py:49-55,67-77) with weights whose result is known without any restatement:
        layer 0:  h[i] = relu(x[i]), h[32 + i] = relu(-x[i])                      (W0[i][i] = 1, W0[32 + i][i] = -1)
        layer 1:  (two hidden layers only) a reversal, h'[j] = h[63 - j]           (W1[j][63 - j] = 1)
        output :  y[k] = 2 relu(x[k]) - 2 relu(-x[k]) + 0.5 relu(x[k + 16]) = 2 x[k] + 0.5 max(x[k + 16], 0),  k < 16
    on inputs that are exact in f16.  A transposed layout, a swapped layer order, a bias or a different padding all change y."""
    from ngp_pl_amd import tcnn
    net = tcnn.Network(32, 16, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64,
                                "n_hidden_layers": n_hidden}).cuda()
    W0 = torch.zeros(64, 32); W1 = torch.zeros(64, 64); Wo = torch.zeros(16, 64)
    for i in range(32):
        W0[i, i] = 1.0; W0[32 + i, i] = -1.0
    pos = (lambda j: 63 - j) if n_hidden == 2 else (lambda j: j)             # where unit j of layer 0 sits in the last hidden layer
    for j in range(64):
        W1[j, 63 - j] = 1.0
    for k in range(16):
        Wo[k, pos(k)] = 2.0; Wo[k, pos(32 + k)] = -2.0; Wo[k, pos(16 + k)] = 0.5
    blob = torch.cat([W0.reshape(-1)] + ([W1.reshape(-1)] if n_hidden == 2 else []) + [Wo.reshape(-1)])
    assert blob.numel() == net.params.numel() == 64 * 32 + (64 * 64 if n_hidden == 2 else 0) + 16 * 64
    with torch.no_grad():
        net.params.copy_(blob.cuda())
    net._half.invalidate()
    g = torch.Generator().manual_seed(1)
    x = (torch.randint(-32, 33, (4099, 32), generator=g).float() / 8.0)      # multiples of 1/8 in [-4, 4]: exact in f16, sums exact too
    want = 2.0 * x[:, :16] + 0.5 * x[:, 16:32].clamp(min=0)
    got = net(x.cuda().half()).float().cpu()
    assert got.shape == (4099, 16) and torch.equal(got, want)
