"""GPU parity of the tiny-cuda-nn replacements (hash grid, fused MLPs, SH) against the fp32
torch oracle with the kernels' f16 rounding points inserted (oracle/tcnn_oracle.py, quantize=True).

Tolerances (stated per check) are f16-level: features/activations are stored as f16 (ulp 2^-11
relative), MFMA accumulates in f32 in a different order than torch.matmul, gradients of the table
are accumulated in packed f16 (as tiny-cuda-nn does).  This is parity against OUR restatement of
tiny-cuda-nn's published algorithm -- the reference does not pin it (see the oracle's header).
"""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

from oracle import tcnn_oracle as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available()
    from ngp_pl_amd import _lib
    return _lib


@pytest.fixture(scope="module")
def field():
    """Oracle field with trained-like magnitudes (table O(1), not the 1e-4 init) so that every
    level and every weight matters in the comparisons."""
    f = T.Field(scale=0.5, seed=7)
    g = torch.Generator().manual_seed(8)
    f.table = ((torch.rand(f.meta.total, 2, generator=g) * 2 - 1) * 0.8).half().float()
    f.density_w = (f.density_w * 1.5).half().float()
    f.rgb_w = (f.rgb_w * 1.5).half().float()
    return f


def native_meta(lib):
    meta = lib.GridMeta()
    b = math.exp(math.log(2048 * 0.5 / 16) / 15)
    lib.call("ngp_grid_meta_init", C.byref(meta), 16, 2, 19, 16, float(b))
    return meta


def sample_points(n, seed=0, edges=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, 3, generator=g) - 0.5
    if edges and n >= 8:
        x[0] = torch.tensor([-0.5, -0.5, -0.5]); x[1] = torch.tensor([0.5, 0.5, 0.5])     # box corners: x01 = 0 / 1
        x[2] = torch.tensor([0.0, 0.0, 0.0]); x[3] = torch.tensor([0.5, -0.5, 0.25])
    d = torch.randn(n, 3, generator=g) * 1.3       # un-normalised directions, like get_rays produces
    return x, d


def run_hash_fwd(lib, meta, x, table_h, mn=-0.5, mx=0.5):
    n = x.shape[0]
    xs = x.cuda().contiguous()
    feats = torch.empty(16, n, 2, dtype=torch.float16, device="cuda")
    mnt = torch.full((3,), mn, device="cuda"); mxt = torch.full((3,), mx, device="cuda")
    lib.call("ngp_hashgrid_fwd", lib.ptr(xs), lib.ptr(mnt), lib.ptr(mxt), lib.ptr(table_h), C.byref(meta), n, lib.ptr(feats), lib.stream())
    return feats


def test_grid_meta_matches_oracle(lib, field):
    meta = native_meta(lib)
    assert [meta.resolution[i] for i in range(16)] == field.meta.resolution
    assert [meta.offset[i] for i in range(17)] == field.meta.offset
    np.testing.assert_allclose([meta.scale[i] for i in range(16)], field.meta.scale, rtol=0, atol=0)
    # sizes: 16^3 dense at level 0, 2^19 hashed entries from level 6 on
    assert meta.offset[1] == 4096 and meta.offset[7] - meta.offset[6] == 1 << 19


def test_hashgrid_forward(lib, field):
    meta = native_meta(lib)
    x, _ = sample_points(20000, seed=1)
    table_h = field.table.half().cuda()
    feats = run_hash_fwd(lib, meta, x, table_h)
    got = feats.permute(1, 0, 2).reshape(x.shape[0], 32).float().cpu()
    want = T.hash_encode(x + 0.5, field.table, field.meta, quantize=True)
    # |values| <= 0.8, f16 storage: 1 ulp at that magnitude is 4.9e-4; allow 2 ulp
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=1e-3)
    # row-major converters round-trip
    rm = torch.empty(x.shape[0], 32, dtype=torch.float16, device="cuda")
    lib.call("ngp_feats_to_rowmajor", lib.ptr(feats), 16, x.shape[0], lib.ptr(rm), lib.stream())
    assert torch.equal(rm.cpu().float(), got)
    back = torch.empty_like(feats)
    lib.call("ngp_feats_from_rowmajor", lib.ptr(rm), 16, x.shape[0], lib.ptr(back), lib.stream())
    assert torch.equal(back, feats)


@pytest.mark.parametrize("mode", ["sliced", "binned", "atomic_f16", "atomic_f32"])
def test_hashgrid_backward(lib, field, mode):
    meta = native_meta(lib)
    n = 30000
    # ray-like coherent samples (runs of consecutive points) + random ones: collisions and contention
    x, _ = sample_points(n, seed=2)
    o = torch.rand(200, 1, 3) - 0.5; dd = torch.randn(200, 1, 3); dd /= dd.norm(dim=-1, keepdim=True)
    x[:20000] = (o + dd * (torch.arange(100).view(1, 100, 1) * 1.7e-3)).clamp(-0.5, 0.5).reshape(-1, 3)
    g = torch.Generator().manual_seed(3)
    dfe = (torch.randn(n, 32, generator=g) * 0.5).half()
    dfe[::7] = 0                                    # samples with exactly zero gradient are skipped
    table = field.table.clone().requires_grad_(True)
    feats = T.hash_encode(x + 0.5, table, field.meta)
    feats.backward(dfe.float())
    want = table.grad
    xs = x.cuda().contiguous()
    dfl = dfe.view(n, 16, 2).permute(1, 0, 2).contiguous().cuda()
    mnt = torch.full((3,), -0.5, device="cuda"); mxt = torch.full((3,), 0.5, device="cuda")
    if mode == "sliced":
        grad = torch.full((field.meta.total, 2), float("nan"), dtype=torch.float16, device="cuda")   # must be fully overwritten
        lib.call("ngp_hashgrid_bwd_sliced", lib.ptr(xs), lib.ptr(mnt), lib.ptr(mxt), lib.ptr(dfl), C.byref(meta), n, None, None, lib.ptr(grad), lib.stream())
    elif mode == "binned":
        grad = torch.full((field.meta.total, 2), float("nan"), dtype=torch.float16, device="cuda")
        nbytes = lib.lib().ngp_hashgrid_bwd_binned_workspace_bytes(C.byref(meta), n)
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        lib.call("ngp_hashgrid_bwd_binned", lib.ptr(xs), lib.ptr(mnt), lib.ptr(mxt), lib.ptr(dfl), C.byref(meta), n, None, None,
                 lib.ptr(ws), nbytes, lib.ptr(grad), lib.stream())
        # fixed-point integer accumulation: the result does not depend on the order in which updates arrive
        again = torch.full_like(grad, float("nan"))
        ws.zero_()
        lib.call("ngp_hashgrid_bwd_binned", lib.ptr(xs), lib.ptr(mnt), lib.ptr(mxt), lib.ptr(dfl), C.byref(meta), n, None, None,
                 lib.ptr(ws), nbytes, lib.ptr(again), lib.stream())
        assert torch.equal(grad, again)
    else:
        f32 = mode == "atomic_f32"
        grad = torch.zeros(field.meta.total, 2, dtype=torch.float32 if f32 else torch.float16, device="cuda")
        lib.call("ngp_hashgrid_bwd", lib.ptr(xs), lib.ptr(mnt), lib.ptr(mxt), lib.ptr(dfl), C.byref(meta), n, lib.ptr(grad), int(f32), lib.stream())
    got = grad.float().cpu()
    assert torch.isfinite(got).all()
    # packed-f16 accumulation: each add rounds to 2^-11 relative of the running sum
    tol = {"atomic_f32": 3e-5, "binned": 1e-3}.get(mode, 1e-2)   # f16 atomics: up to ~100 contributions per entry, each add rounds at 2^-11; binned: exact sums, one final f16 rounding (+ the f16 dfeats x f32 weights products in 2^-24 units)
    scale = want.abs().max().item()
    err = (got - want).abs().max().item() / scale
    assert err < tol, "max error %g of max |grad| %g" % (err, scale)
    assert (got != 0).sum() == (want != 0).sum() or mode != "atomic_f32"


def outlier_frac(got, want, tol):
    """fraction of elements whose error exceeds tol * max|want|"""
    return ((got - want).abs() > tol * want.abs().max()).float().mean().item()


def field_native(lib, field, x, d):
    meta = native_meta(lib)
    n = x.shape[0]
    table_h = field.table.half().cuda()
    feats = run_hash_fwd(lib, meta, x, table_h)
    dw, rw = field.density_w.half().cuda(), field.rgb_w.half().cuda()
    sig = torch.empty(n, device="cuda"); rgb = torch.empty(n, 3, device="cuda")
    h = torch.empty(n, 16, dtype=torch.float16, device="cuda")
    ds = d.cuda().contiguous()
    lib.call("ngp_field_fwd", lib.ptr(feats), lib.ptr(ds), lib.ptr(dw), lib.ptr(rw), n, lib.ptr(sig), lib.ptr(rgb), lib.ptr(h), lib.stream())
    return feats, sig, rgb, h, dw, rw, ds


def test_field_forward(lib, field):
    for n in (1, 31, 32, 33, 4097, 50000):          # ragged tails around the 32-sample MFMA tile
        x, d = sample_points(n, seed=4)
        feats, sig, rgb, h, *_ = field_native(lib, field, x, d)
        ws, wr, wh = field.forward(x, d, quantize=True)
        # h: f16 output of a 64-wide dot product of f16 activations; 2 ulp of the largest |h| + summation noise
        hmax = wh.abs().max().item()
        np.testing.assert_allclose(h.float().cpu().numpy(), wh.detach().numpy(), rtol=2e-3, atol=2e-3 * max(hmax, 1.0))
        np.testing.assert_allclose(rgb.cpu().numpy(), wr.detach().numpy(), rtol=0, atol=2e-3)      # north-star asks 1e-4 vs CUDA; f16 ulp at 1.0 is 4.9e-4
        np.testing.assert_allclose(sig.cpu().numpy(), ws.detach().numpy(), rtol=1e-2, atol=1e-6)   # sigma = exp(h0): one f16 ulp of h0 ~ 4 moves sigma by 0.4 %


def test_sh_encoding(lib):
    g = torch.Generator().manual_seed(5)
    d = torch.randn(5000, 3, generator=g); d /= d.norm(dim=1, keepdim=True)
    out = torch.empty(5000, 16, dtype=torch.float16, device="cuda")
    d01 = ((d + 1) / 2).cuda().contiguous()
    lib.call("ngp_sh4_fwd", lib.ptr(d01), 5000, lib.ptr(out), lib.stream())
    want = T.sh4(((d + 1) / 2) * 2 - 1)
    np.testing.assert_allclose(out.float().cpu().numpy(), want.numpy(), rtol=2e-3, atol=1e-3)
    # closed forms: constant term and the z-only band at the pole
    pole = torch.tensor([[0.5, 0.5, 1.0]]).cuda()
    o1 = torch.empty(1, 16, dtype=torch.float16, device="cuda")
    lib.call("ngp_sh4_fwd", lib.ptr(pole), 1, lib.ptr(o1), lib.stream())
    assert abs(o1[0, 0].item() - 0.2821) < 1e-3 and abs(o1[0, 2].item() - 0.4886) < 1e-3 and abs(o1[0, 6].item() - 0.6308) < 1e-3


def test_hashgrid_input_gradient(lib, field):
    """dL/dx through the encoding (pose optimisation, train.py:86-89): native kernel vs autograd of the
    fp32 oracle.  The interpolant is piecewise linear: away from cell faces the derivative is exact up
    to f16 rounding of the table/dfeats products."""
    meta = native_meta(lib)
    n = 6000
    x, _ = sample_points(n, seed=11, edges=False)
    g = torch.Generator().manual_seed(12)
    dfe = (torch.randn(n, 32, generator=g) * 0.05).half()            # upstream gradient, f16 like the MLP's dgrad
    x01 = (x + 0.5).clone().requires_grad_(True)
    out = T.hash_encode(x01, field.table, field.meta)
    (out * dfe.float()).sum().backward()
    want = x01.grad / 1.0                                              # d/dx = d/dx01 / (max - min), extent 1
    table_h = field.table.half().cuda()
    dfeats = dfe.view(n, 16, 2).permute(1, 0, 2).contiguous().cuda()
    xs = x.cuda().contiguous()
    mnt = torch.full((3,), -0.5, device="cuda"); mxt = torch.full((3,), 0.5, device="cuda")
    dx = torch.empty(n, 3, device="cuda")
    lib.call("ngp_hashgrid_bwd_input", lib.ptr(xs), lib.ptr(mnt), lib.ptr(mxt), lib.ptr(table_h), lib.ptr(dfeats), C.byref(meta), n, 1.0,
             lib.ptr(dx), lib.stream())
    scale = want.abs().max().item()
    np.testing.assert_allclose(dx.cpu().numpy(), want.numpy(), rtol=2e-3, atol=2e-4 * scale)
    # scaling argument and a non-unit box: d/dx scales with 1/(max-min)
    mnt2 = torch.full((3,), -1.0, device="cuda"); mxt2 = torch.full((3,), 1.0, device="cuda")
    xs2 = (xs * 2).contiguous()
    dx2 = torch.empty(n, 3, device="cuda")
    lib.call("ngp_hashgrid_bwd_input", lib.ptr(xs2), lib.ptr(mnt2), lib.ptr(mxt2), lib.ptr(table_h), lib.ptr(dfeats), C.byref(meta), n, 4.0,
             lib.ptr(dx2), lib.stream())
    np.testing.assert_allclose(dx2.cpu().numpy(), 2.0 * dx.cpu().numpy(), rtol=1e-5, atol=1e-6 * scale)


def test_sh_input_gradient(lib):
    g = torch.Generator().manual_seed(13)
    d = torch.randn(4000, 3, generator=g); d /= d.norm(dim=1, keepdim=True)
    d01 = ((d + 1) / 2).clone().requires_grad_(True)
    gsh = (torch.randn(4000, 16, generator=g) * 0.1).half()
    (T.sh4(d01 * 2 - 1) * gsh.float()).sum().backward()
    out = torch.empty(4000, 3, device="cuda")
    d01c = d01.detach().cuda().contiguous(); gc = gsh.cuda().contiguous()
    lib.call("ngp_sh4_bwd", lib.ptr(d01c), lib.ptr(gc), 4000, 1.0, lib.ptr(out), lib.stream())
    np.testing.assert_allclose(out.cpu().numpy(), d01.grad.numpy(), rtol=1e-4, atol=1e-5)


def test_field_backward(lib, field):
    n = 20011
    x, d = sample_points(n, seed=6)
    feats, sig, rgb, h, dw, rw, ds = field_native(lib, field, x, d)
    g = torch.Generator().manual_seed(7)
    dsig = torch.randn(n, generator=g) * 1e-3
    drgb = torch.randn(n, 3, generator=g) * 1e-2
    dsig[::5] = 0; drgb[::5] = 0
    # oracle: gradients w.r.t. both weight blobs and the (quantised) features
    dwp = field.density_w.clone().requires_grad_(True); rwp = field.rgb_w.clone().requires_grad_(True)
    f_in = T.hash_encode(x + 0.5, field.table, field.meta, quantize=True).detach().requires_grad_(True)
    hh = T.q16(T.mlp(f_in, dwp, 32, 1, 16, "None", True))
    s_o = T.TruncExp.apply(hh[:, 0])
    dn = d / d.norm(dim=1, keepdim=True)
    c_o = T.q16(T.mlp(torch.cat([T.q16(T.sh4(dn)), hh], 1), rwp, 32, 2, 3, "Sigmoid", True))
    ((s_o * dsig).sum() + (c_o * drgb).sum()).backward()
    scale = 128.0
    n_part = lib.call("ngp_field_bwd_partials", n)
    partials = torch.empty(n_part * 10240, device="cuda")
    dh = torch.empty(n, 16, dtype=torch.float16, device="cuda"); dfeats = torch.empty(16, n, 2, dtype=torch.float16, device="cuda")
    dsig_d, drgb_d = dsig.cuda(), drgb.cuda().contiguous()      # named: a temporary would be freed before the kernel runs
    lib.call("ngp_field_bwd", lib.ptr(feats), lib.ptr(ds), lib.ptr(h), lib.ptr(dw), lib.ptr(rw), lib.ptr(dsig_d), lib.ptr(drgb_d),
             scale, n, None, None, lib.ptr(dh), lib.ptr(dfeats), lib.ptr(partials), lib.stream())
    gd = torch.empty(3072, device="cuda"); gr = torch.empty(7168, device="cuda")
    lib.call("ngp_reduce_partials", lib.ptr(partials), n_part, 3072, lib.ptr(gd), lib.stream())
    lib.call("ngp_reduce_partials", lib.ptr(partials[n_part * 3072:]), n_part, 7168, lib.ptr(gr), lib.stream())

    def rel(got, want):
        return ((got - want).abs().max() / want.abs().max()).item()
    # activation gradients pass through f16 (x128): 2^-11 per rounding, a handful of roundings per path
    assert rel(gr.cpu() / scale, rwp.grad) < 1e-2, "rgb net weight grad"
    assert rel(gd.cpu() / scale, dwp.grad) < 1e-2, "density net weight grad"
    got_df = dfeats.permute(1, 0, 2).reshape(n, 32).float().cpu() / scale
    # per-sample gradients: a hidden unit whose pre-activation is within rounding of 0 can land on
    # either side of the ReLU in the two implementations, which changes that ONE sample's gradient
    # by a whole unit's contribution; so bound the bulk tightly and the outliers loosely
    assert outlier_frac(got_df, f_in.grad, 1e-2) < 1e-3 and rel(got_df, f_in.grad) < 0.2, "feature grad"
    assert (got_df[::5] == 0).all()                 # zero seeds -> exact zeros (the grid scatter skips them)
    assert (rwp.grad[64 * 32 + 64 * 64 + 3 * 64:] == 0).all() and (gr.cpu()[64 * 32 + 64 * 64 + 3 * 64:] == 0).all()   # padded output rows


@pytest.mark.parametrize("n_in,n_hidden,n_out,act", [(32, 2, 3, 1), (16, 1, 1, 1), (32, 1, 16, 0), (64, 2, 8, 0), (16, 2, 16, 0), (64, 1, 4, 1)])
def test_generic_mlp(lib, n_in, n_hidden, n_out, act):
    g = torch.Generator().manual_seed(n_in + n_hidden + n_out)
    n = 10007
    from ngp_pl_amd.tcnn import mlp_init
    w = (mlp_init(g, n_in, n_hidden) * 1.5).half().float()
    x = (torch.randn(n, n_in, generator=g)).half()
    out = torch.empty(n, n_out, dtype=torch.float16, device="cuda")
    wh = w.half().cuda(); xc = x.cuda()
    lib.call("ngp_mlp_fwd", lib.ptr(xc), lib.ptr(wh), n_in, n_hidden, n_out, act, n, lib.ptr(out), lib.stream())
    wp = w.clone().requires_grad_(True); xp = x.float().requires_grad_(True)
    want = T.mlp(xp, wp, n_in, n_hidden, n_out, "Sigmoid" if act else "None", quantize=True)
    np.testing.assert_allclose(out.float().cpu().numpy(), want.detach().numpy(), rtol=3e-3, atol=3e-3 * max(1.0, want.abs().max().item()))
    dout = (torch.randn(n, n_out, generator=g) * 0.1).half()
    want.backward(dout.float())
    n_part = lib.call("ngp_mlp_bwd_partials", n)
    partials = torch.empty(n_part, w.numel(), device="cuda")
    din = torch.empty(n, n_in, dtype=torch.float16, device="cuda")
    dout_d = dout.cuda()
    lib.call("ngp_mlp_bwd", lib.ptr(xc), lib.ptr(wh), lib.ptr(dout_d), n_in, n_hidden, n_out, act, n, lib.ptr(din), lib.ptr(partials), lib.stream())
    gw = partials.sum(0).cpu()
    assert ((gw - wp.grad).abs().max() / wp.grad.abs().max()).item() < 1e-2
    dgot = din.float().cpu()
    assert outlier_frac(dgot, xp.grad, 1e-2) < 1e-3 and ((dgot - xp.grad).abs().max() / xp.grad.abs().max()).item() < 0.2   # ReLU-boundary flips, see test_field_backward


def test_unsupported_config_is_loud(lib):
    with pytest.raises(lib.NgpError):
        x = torch.zeros(16, device="cuda")
        lib.call("ngp_mlp_fwd", lib.ptr(x), lib.ptr(x), 48, 1, 3, 0, 10, lib.ptr(x), lib.stream())
    from ngp_pl_amd import tcnn
    with pytest.raises(NotImplementedError):
        tcnn.Network(32, 3, {"otype": "FullyFusedMLP", "activation": "Tanh", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2})


def test_active_sample_compaction(lib, field):
    """Backward over the compacted active-sample list == backward over all samples when the
    skipped samples carry zero seeds (what compositing guarantees past a ray's early stop)."""
    meta = native_meta(lib)
    g = torch.Generator().manual_seed(12)
    n_rays = 700
    counts = torch.randint(0, 60, (n_rays,), generator=g)
    counts[5] = 0
    start = torch.cumsum(counts, 0) - counts
    n = int(counts.sum())
    rays_a = torch.stack([torch.arange(n_rays), start, counts], 1).contiguous()
    total = torch.minimum(counts, torch.randint(0, 70, (n_rays,), generator=g))      # samples before the stop
    n_act = torch.minimum(counts, total + 1)
    mask = torch.zeros(n, dtype=torch.bool)
    for r in range(n_rays):
        mask[start[r]:start[r] + n_act[r]] = True
    x, d = sample_points(n, seed=13, edges=False)
    feats, sig, rgb, h, dw, rw, ds = field_native(lib, field, x, d)
    dsig = torch.randn(n, generator=g) * 1e-3 * mask
    drgb = torch.randn(n, 3, generator=g) * 1e-2 * mask[:, None]
    xs = x.cuda().contiguous()
    mnt = torch.full((3,), -0.5, device="cuda"); mxt = torch.full((3,), 0.5, device="cuda")
    active = torch.full((n,), -1, dtype=torch.int32, device="cuda"); n_active = torch.zeros(1, dtype=torch.int32, device="cuda")
    rays_a_d, total_d = rays_a.cuda(), total.cuda()
    n_act_d = torch.empty(n_rays, dtype=torch.int32, device="cuda")
    lib.call("ngp_active_samples", lib.ptr(rays_a_d), lib.ptr(total_d), n_rays, lib.ptr(n_act_d), lib.ptr(active), lib.ptr(n_active), lib.stream())
    assert int(n_active.item()) == int(mask.sum())
    assert torch.equal(active[:int(n_active.item())].cpu().long(), torch.nonzero(mask)[:, 0])      # ray order, bit-exact
    outs = []
    for use_active in (False, True):
        n_part = lib.call("ngp_field_bwd_partials", n)
        partials = torch.zeros(n_part * 10240, device="cuda")
        dh = torch.zeros(n, 16, dtype=torch.float16, device="cuda"); dfeats = torch.zeros(16, n, 2, dtype=torch.float16, device="cuda")
        a, na = (lib.ptr(active), lib.ptr(n_active)) if use_active else (None, None)
        dsig_d, drgb_d = dsig.cuda(), drgb.cuda().contiguous()
        lib.call("ngp_field_bwd", lib.ptr(feats), lib.ptr(ds), lib.ptr(h), lib.ptr(dw), lib.ptr(rw), lib.ptr(dsig_d), lib.ptr(drgb_d),
                 128.0, n, a, na, lib.ptr(dh), lib.ptr(dfeats), lib.ptr(partials), lib.stream())
        grad = torch.full((field.meta.total, 2), float("nan"), dtype=torch.float16, device="cuda")
        lib.call("ngp_hashgrid_bwd_sliced", lib.ptr(xs), lib.ptr(mnt), lib.ptr(mxt), lib.ptr(dfeats), C.byref(meta), n, a, na, lib.ptr(grad), lib.stream())
        wsum = torch.cat([partials[:n_part * 3072].view(n_part, 3072).sum(0), partials[n_part * 3072:].view(n_part, 7168).sum(0)])
        outs.append((wsum.cpu(), grad.float().cpu()))
    (w0, g0), (w1, g1) = outs
    assert ((w0 - w1).abs().max() / w0.abs().max()).item() < 1e-4       # same terms, different partial-sum grouping
    assert ((g0 - g1).abs().max() / g0.abs().max()).item() < 5e-3       # f16 accumulation order differs


# ---------------------------------------------------------------------------------------------------------
# distance to the UN-QUANTISED fp32 field: how far is the f16-storage / f32-accumulate
# native field from "any correct tiny-cuda-nn", with numbers.  north_star asks RGB / sigma within 1e-4 abs of the
# CUDA path; tiny-cuda-nn itself stores features, activations and outputs in f16 (ulp at 1.0: 4.9e-4, at 0.5: 2.4e-4),
# so neither it nor this path can hold 1e-4 against exact arithmetic -- these tests state what IS held.
# ---------------------------------------------------------------------------------------------------------
def _dist(name, err, fh=None):
    err = err.flatten().double()
    q = torch.quantile(err, torch.tensor([0.5, 0.99], dtype=torch.float64))
    rec = dict(median=float(q[0]), p99=float(q[1]), max=float(err.max()))
    print("  %-34s median %.3e   p99 %.3e   max %.3e" % (name, rec["median"], rec["p99"], rec["max"]))
    return rec


def test_field_error_against_the_unquantised_fp32_oracle(lib, field):
    """Error distribution (median / p99 / max) of the native field against the fp32 oracle WITHOUT rounding points,
    next to (a) the oracle with the kernels' f16 rounding points and (b) the oracle with tiny-cuda-nn's f16
    accumulators.  The asserted bounds are what DESIGN.md section 2 quotes."""
    n = 60000
    x, d = sample_points(n, seed=21)
    feats, sig, rgb, h, *_ = field_native(lib, field, x, d)
    got_f = feats.permute(1, 0, 2).reshape(n, 32).float().cpu()
    got_s, got_c = sig.cpu(), rgb.cpu()
    want_f = T.hash_encode(x + 0.5, field.table, field.meta)                 # fp32, no rounding
    ws, wc, _ = field.forward(x, d)                                          # fp32 end to end
    qs, qc, _ = field.forward(x, d, quantize=True)                           # f16 storage, f32 accumulation (this path's model)
    as_, ac, _ = field.forward(x, d, acc16=True)                             # f16 storage, f16 accumulators (tiny-cuda-nn's model)
    print("\nnative HIP field vs oracle variants, %d samples, |table| <= 0.8, |h0| up to %.1f:" % (n, float(field.density(x)[1][:, 0].abs().max())))
    r = {}
    r["feat"] = _dist("features abs (vs fp32)", (got_f - want_f).abs())
    r["rgb"] = _dist("rgb abs (vs fp32)", (got_c - wc.detach()).abs())
    r["sig"] = _dist("sigma rel (vs fp32)", ((got_s - ws.detach()).abs() / ws.detach().clamp(min=1e-6)))
    r["rgb_q"] = _dist("rgb abs (vs f16-storage oracle)", (got_c - qc.detach()).abs())
    r["sig_q"] = _dist("sigma rel (vs f16-storage oracle)", ((got_s - qs.detach()).abs() / qs.detach().clamp(min=1e-6)))
    r["rgb_a"] = _dist("rgb abs (vs f16-accumulate oracle)", (got_c - ac.detach()).abs())
    r["sig_a"] = _dist("sigma rel (vs f16-accumulate)", ((got_s - as_.detach()).abs() / as_.detach().clamp(min=1e-6)))
    r["tcnn_rgb"] = _dist("f16-accumulate oracle vs fp32: rgb", (ac.detach() - wc.detach()).abs())
    r["tcnn_sig"] = _dist("f16-accumulate oracle vs fp32: sig", ((as_.detach() - ws.detach()).abs() / ws.detach().clamp(min=1e-6)))
    # features: one f16 rounding of a value |v| <= 0.8 on top of exact f32 interpolation: <= 2^-12 = 2.44e-4
    assert r["feat"]["max"] <= 2.6e-4 and r["feat"]["median"] <= 7e-5      # 2^-12 plus the f32 interpolation noise
    # rgb in [0,1] stored as f16 after a sigmoid: half an ulp (2.4e-4 near 1, 1.2e-4 near 0.5) + propagated feature/activation rounding
    assert r["rgb"]["median"] <= 2.5e-4 and r["rgb"]["p99"] <= 1.5e-3 and r["rgb"]["max"] <= 4e-3
    # sigma = exp(h0), h0 stored as f16: |h0| in [4, 8) has ulp 3.9e-3 -> up to 0.2 % from that rounding alone
    assert r["sig"]["median"] <= 2e-3 and r["sig"]["p99"] <= 1e-2 and r["sig"]["max"] <= 3e-2
    # this path sits CLOSER to exact arithmetic than the f16-accumulator model of tiny-cuda-nn does
    assert r["rgb"]["p99"] <= r["tcnn_rgb"]["p99"] * 1.05 and r["sig"]["p99"] <= r["tcnn_sig"]["p99"] * 1.05


def test_gradient_error_against_the_unquantised_fp32_oracle(lib, field):
    """Same question for the backward: weight gradients and per-sample feature gradients of the native fused
    backward against fp32 autograd of the oracle WITHOUT rounding points (the existing tests compare with the
    rounding points inserted)."""
    n = 20000
    x, d = sample_points(n, seed=22)
    feats, sig, rgb, h, dw, rw, ds = field_native(lib, field, x, d)
    g = torch.Generator().manual_seed(23)
    dsig = torch.randn(n, generator=g) * 1e-3; drgb = torch.randn(n, 3, generator=g) * 1e-2
    dwp = field.density_w.clone().requires_grad_(True); rwp = field.rgb_w.clone().requires_grad_(True)
    f_in = T.hash_encode(x + 0.5, field.table, field.meta).detach().requires_grad_(True)
    hh = T.mlp(f_in, dwp, 32, 1, 16)
    s_o = T.TruncExp.apply(hh[:, 0])
    c_o = T.mlp(torch.cat([T.sh4(d / d.norm(dim=1, keepdim=True)), hh], 1), rwp, 32, 2, 3, "Sigmoid")
    ((s_o * dsig).sum() + (c_o * drgb).sum()).backward()
    n_part = lib.call("ngp_field_bwd_partials", n)
    partials = torch.empty(n_part * 10240, device="cuda")
    dh = torch.empty(n, 16, dtype=torch.float16, device="cuda"); dfeats = torch.empty(16, n, 2, dtype=torch.float16, device="cuda")
    dsig_d, drgb_d = dsig.cuda(), drgb.cuda().contiguous()
    lib.call("ngp_field_bwd", lib.ptr(feats), lib.ptr(ds), lib.ptr(h), lib.ptr(dw), lib.ptr(rw), lib.ptr(dsig_d), lib.ptr(drgb_d),
             128.0, n, None, None, lib.ptr(dh), lib.ptr(dfeats), lib.ptr(partials), lib.stream())
    gd = partials[:n_part * 3072].view(n_part, 3072).sum(0).cpu() / 128.0
    gr = partials[n_part * 3072:].view(n_part, 7168).sum(0).cpu() / 128.0
    got_df = dfeats.permute(1, 0, 2).reshape(n, 32).float().cpu() / 128.0
    print("\nnative fused backward vs fp32 autograd without rounding points, %d samples (errors relative to max |gradient|):" % n)
    r_d = _dist("density-net weight grad", (gd - dwp.grad).abs() / dwp.grad.abs().max())
    r_r = _dist("rgb-net weight grad", (gr - rwp.grad).abs() / rwp.grad.abs().max())
    r_f = _dist("per-sample feature grad", (got_df - f_in.grad).abs() / f_in.grad.abs().max())
    # measured on MI355X: density net median 3.4e-3 / max 2.2e-2, rgb net median 3.4e-4 / max 1.5e-2 of the largest entry (the
    # density net's gradient passes through the f16 dL/dh and the exp() of an f16 h0; with the rounding points inserted in the
    # oracle the same comparison holds 1e-2, test_field_backward)
    assert r_d["max"] <= 4e-2 and r_d["median"] <= 6e-3 and r_r["max"] <= 3e-2 and r_r["median"] <= 1e-3
    assert r_f["p99"] <= 1e-2 and r_f["max"] <= 0.25                 # a ReLU unit within rounding of 0 flips for single samples


def test_points_outside_the_box_stay_in_bounds(lib, field):
    """positions outside [xyz_min, xyz_max] (public NGP.density()/forward() callers) used to index past
    the dense levels.  They now clamp to the border cell: forward, all table backwards and the input gradient stay in
    bounds (finite results, untouched guard bands), and in-box rows of a mixed batch are exactly what they are alone."""
    meta = native_meta(lib)
    n = 8192
    g = torch.Generator().manual_seed(31)
    x_in = torch.rand(n, 3, generator=g) - 0.5
    x_out = (torch.rand(n, 3, generator=g) - 0.5) * 8.0                 # up to 4x the half extent, both signs
    x_out[:4] = torch.tensor([[-0.5001, 0.0, 0.0], [0.5001, 0.5001, 0.5001], [-30.0, 40.0, -50.0], [1e6, -1e6, 1e6]])
    x = torch.cat([x_in, x_out])
    guard = 4096
    total = field.meta.total
    buf = torch.zeros(total + 2 * guard, 2, dtype=torch.float16, device="cuda")
    buf[guard:guard + total] = field.table.half().cuda()
    buf[:guard] = float("nan"); buf[guard + total:] = float("nan")       # a read past either end poisons the output
    table_h = buf[guard:guard + total]
    feats = run_hash_fwd(lib, meta, x, table_h)
    alone = run_hash_fwd(lib, meta, x_in, table_h)
    assert torch.isfinite(feats.float()).all()
    assert torch.equal(feats[:, :n], alone)
    dfe = (torch.randn(2 * n, 32, generator=g) * 0.1).half()
    dfl = dfe.view(2 * n, 16, 2).permute(1, 0, 2).contiguous().cuda()
    xs = x.cuda().contiguous()
    mnt = torch.full((3,), -0.5, device="cuda"); mxt = torch.full((3,), 0.5, device="cuda")
    for mode in ("sliced", "binned", "atomic"):
        gbuf = torch.zeros(total + 2 * guard, 2, dtype=torch.float16, device="cuda")
        gbuf[:guard] = 7.0; gbuf[guard + total:] = 7.0
        grad = gbuf[guard:guard + total]
        if mode == "sliced":
            lib.call("ngp_hashgrid_bwd_sliced", lib.ptr(xs), lib.ptr(mnt), lib.ptr(mxt), lib.ptr(dfl), C.byref(meta), 2 * n, None, None, lib.ptr(grad), lib.stream())
        elif mode == "binned":
            nbytes = lib.lib().ngp_hashgrid_bwd_binned_workspace_bytes(C.byref(meta), 2 * n)
            ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            lib.call("ngp_hashgrid_bwd_binned", lib.ptr(xs), lib.ptr(mnt), lib.ptr(mxt), lib.ptr(dfl), C.byref(meta), 2 * n, None, None,
                     lib.ptr(ws), nbytes, lib.ptr(grad), lib.stream())
        else:
            lib.call("ngp_hashgrid_bwd", lib.ptr(xs), lib.ptr(mnt), lib.ptr(mxt), lib.ptr(dfl), C.byref(meta), 2 * n, lib.ptr(grad), 0, lib.stream())
        torch.cuda.synchronize()
        assert torch.isfinite(grad.float()).all(), mode
        assert (gbuf[:guard] == 7.0).all() and (gbuf[guard + total:] == 7.0).all(), "%s wrote outside the table" % mode
    dx = torch.empty(2 * n, 3, device="cuda")
    lib.call("ngp_hashgrid_bwd_input", lib.ptr(xs), lib.ptr(mnt), lib.ptr(mxt), lib.ptr(table_h), lib.ptr(dfl), C.byref(meta), 2 * n, 1.0,
             lib.ptr(dx), lib.stream())
    assert torch.isfinite(dx).all()
    # the module-level entry points the advisor named
    from ngp_pl_amd.networks import NGP
    m = NGP(scale=0.5).cuda()
    with torch.no_grad():
        s = m.density(xs)
        s2, c2 = m(xs, torch.randn(2 * n, 3, device="cuda"))
    assert torch.isfinite(s).all() and torch.isfinite(s2).all() and torch.isfinite(c2.float()).all()


def test_exact_level_table_option(lib):
    """NGP(level_table='exact') (checkpoints whose xyz_encoder.params has the exact-arithmetic length, ngp_pl_amd/utils.py):
    the kernels take the table from the meta record, so forward and table backward must match the oracle built on the same table."""
    from ngp_pl_amd import tcnn
    b = math.exp(math.log(2048 * 0.5 / 16) / 15)
    meta = tcnn.make_grid_meta({"otype": "Grid", "type": "Hash", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19,
                                "base_resolution": 16, "per_level_scale": b, "interpolation": "Linear"}, level_table="exact")
    om = T.GridMeta(16, 2, 19, 16, b, exact=True)
    assert [meta.resolution[i] for i in range(16)] == om.resolution and [meta.offset[i] for i in range(17)] == om.offset
    assert om.total * 2 + 3072 == 11_423_136
    g = torch.Generator().manual_seed(41)
    table = ((torch.rand(om.total, 2, generator=g) * 2 - 1) * 0.8).half().float()
    x, _ = sample_points(20000, seed=42)
    feats = run_hash_fwd(lib, meta, x, table.half().cuda())
    got = feats.permute(1, 0, 2).reshape(x.shape[0], 32).float().cpu()
    want = T.hash_encode(x + 0.5, table, om, quantize=True)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=1e-3)
    n = x.shape[0]
    dfe = (torch.randn(n, 32, generator=g) * 0.5).half()
    tb = table.clone().requires_grad_(True)
    T.hash_encode(x + 0.5, tb, om).backward(dfe.float())
    dfl = dfe.view(n, 16, 2).permute(1, 0, 2).contiguous().cuda()
    xs = x.cuda().contiguous()
    mnt = torch.full((3,), -0.5, device="cuda"); mxt = torch.full((3,), 0.5, device="cuda")
    grad = torch.full((om.total, 2), float("nan"), dtype=torch.float16, device="cuda")
    nbytes = lib.lib().ngp_hashgrid_bwd_binned_workspace_bytes(C.byref(meta), n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    lib.call("ngp_hashgrid_bwd_binned", lib.ptr(xs), lib.ptr(mnt), lib.ptr(mxt), lib.ptr(dfl), C.byref(meta), n, None, None,
             lib.ptr(ws), nbytes, lib.ptr(grad), lib.stream())
    err = (grad.float().cpu() - tb.grad).abs().max().item() / tb.grad.abs().max().item()
    assert err < 1e-3, err


def test_binned_backward_in_launch_groups(lib, field):
    """ngp_hashgrid_bwd_binned_group: n launch groups that each complete a contiguous table range == the single launch, bit
    for bit; a group's range is final as soon as its launch is done (what the multi-GPU exchange hands on piecewise)."""
    meta = native_meta(lib)
    n = 40000
    x, _ = sample_points(n, seed=51)
    g = torch.Generator().manual_seed(52)
    dfe = (torch.randn(n, 32, generator=g) * 0.3).half()
    dfl = dfe.view(n, 16, 2).permute(1, 0, 2).contiguous().cuda()
    xs = x.cuda().contiguous()
    mnt = torch.full((3,), -0.5, device="cuda"); mxt = torch.full((3,), 0.5, device="cuda")
    nbytes = lib.lib().ngp_hashgrid_bwd_binned_workspace_bytes(C.byref(meta), n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    ref = torch.full((field.meta.total, 2), float("nan"), dtype=torch.float16, device="cuda")
    lib.call("ngp_hashgrid_bwd_binned", lib.ptr(xs), lib.ptr(mnt), lib.ptr(mxt), lib.ptr(dfl), C.byref(meta), n, None, None,
             lib.ptr(ws), nbytes, lib.ptr(ref), lib.stream())
    for ng in (2, 3, 5):
        out = torch.full_like(ref, float("nan"))
        covered = 0
        for grp in range(ng):
            a, b = C.c_int64(), C.c_int64()
            lib.call("ngp_hashgrid_bwd_binned_group_entries", C.byref(meta), n, ng, grp, C.byref(a), C.byref(b))
            assert a.value == covered and b.value > a.value
            lib.call("ngp_hashgrid_bwd_binned_group", lib.ptr(xs), lib.ptr(mnt), lib.ptr(mxt), lib.ptr(dfl), C.byref(meta), n, None, None,
                     lib.ptr(ws), nbytes, lib.ptr(out), ng, grp, lib.stream())
            torch.cuda.synchronize()
            assert torch.equal(out[a.value:b.value], ref[a.value:b.value]), (ng, grp)        # final right behind its own launch
            if grp + 1 < ng:
                assert torch.isnan(out[b.value:].float()).all()                                # later ranges untouched so far
            covered = b.value
        assert covered == field.meta.total and torch.equal(out, ref)


def test_exchange_helper_kernels(lib):
    """ngp_reduce_partials2 (both MLP blocks' partial rows in one launch) and ngp_found_inf2 (non-finite check over two
    buffers, alternating flags)."""
    g = torch.Generator(device="cuda").manual_seed(61)
    n_part = 37
    pa = torch.randn(n_part, 3072, device="cuda", generator=g); pb = torch.randn(n_part, 7168, device="cuda", generator=g)
    out = torch.empty(10240, device="cuda")
    lib.call("ngp_reduce_partials2", lib.ptr(pa), 3072, lib.ptr(pb), 7168, n_part, lib.ptr(out), lib.stream())
    one = torch.empty(3072, device="cuda")
    lib.call("ngp_reduce_partials", lib.ptr(pa), n_part, 3072, lib.ptr(one), lib.stream())
    assert torch.equal(out[:3072], one)
    np.testing.assert_allclose(out.cpu().numpy(), torch.cat([pa.sum(0), pb.sum(0)]).cpu().numpy(), rtol=1e-5, atol=1e-5)
    flags = torch.zeros(8, dtype=torch.int32, device="cuda")
    g16 = torch.randn(11445040, device="cuda", generator=g).half(); small = torch.randn(10240, device="cuda", generator=g)
    cur, nxt = flags[0:], flags[4:]
    nxt.fill_(7)
    lib.call("ngp_found_inf2", lib.ptr(g16), 0, g16.numel(), lib.ptr(small), 1, small.numel(), lib.ptr(cur), lib.ptr(nxt), lib.stream())
    assert int(flags[0]) == 0 and int(flags[4]) == 0                       # finite; the other flag was cleared
    for buf, idx, val in ((g16, 11445039, float("inf")), (g16, 5, float("nan")), (small, 10239, float("-inf")), (small, 0, float("nan"))):
        keep = buf[idx].clone(); buf[idx] = val
        flags.zero_()
        lib.call("ngp_found_inf2", lib.ptr(g16), 0, g16.numel(), lib.ptr(small), 1, small.numel(), lib.ptr(cur), lib.ptr(nxt), lib.stream())
        assert int(flags[0]) == 1, (idx, val)
        buf[idx] = keep
    f1 = torch.zeros(1, dtype=torch.int32, device="cuda")
    g16[77] = float("inf")
    lib.call("ngp_found_inf", lib.ptr(g16), 0, g16.numel(), lib.ptr(f1), 1, lib.stream())
    assert int(f1) == 1
    lib.call("ngp_found_inf", lib.ptr(small), 1, small.numel(), lib.ptr(f1), 1, lib.stream())
    assert int(f1) == 0


def test_forward_launch_forms_agree_bit_for_bit(lib):
    """The hash forward's three launch forms -- host-sized (cost-balanced XCD map), device-sized (`_n`: the count and the level
    stride come from device memory, pair map) and over an explicit sample list (`_list`: the two-round forward) -- are scheduling
    only: the same features, bit for bit, on ray-ordered samples with runs of lanes in one cell (incl. a run longer than a wave)."""
    import ctypes as C, math
    from ngp_pl_amd._lib import GridMeta, call, ptr, stream
    torch.manual_seed(0)
    meta = GridMeta()
    call("ngp_grid_meta_init", C.byref(meta), 16, 2, 19, 16, float(math.exp(math.log(2048 * 0.5 / 16) / 15)))
    R, K = 700, 41
    o = torch.rand(R, 1, 3, device="cuda") - 0.5
    d = torch.randn(R, 1, 3, device="cuda"); d = d / d.norm(dim=-1, keepdim=True)
    x = (o * 0.7 + d * (torch.arange(K, device="cuda").view(1, K, 1) * 1.7e-3)).clamp(-0.5, 0.5).reshape(-1, 3).contiguous()
    x[1000:1100] = x[1000]                           # a run longer than a wave
    S = x.shape[0]
    table = ((torch.rand(meta.offset[16], 2, device="cuda") - 0.5) * 0.4).half()
    mn = torch.full((3,), -0.5, device="cuda"); mx = torch.full((3,), 0.5, device="cuda")
    f = torch.zeros(16, S, 2, dtype=torch.float16, device="cuda")
    call("ngp_hashgrid_fwd", ptr(x), ptr(mn), ptr(mx), ptr(table), C.byref(meta), S, ptr(f), stream())
    n = S - 777
    n_dev = torch.tensor([n], dtype=torch.int32, device="cuda")
    g = torch.zeros(16 * S * 2, dtype=torch.float16, device="cuda")                     # device-sized launch: level stride = the count
    call("ngp_hashgrid_fwd_n", ptr(x), ptr(mn), ptr(mx), ptr(table), C.byref(meta), S, ptr(n_dev), ptr(g), stream())
    assert torch.equal(g[:16 * n * 2].view(16, n, 2), f[:, :n]) and not bool(g[16 * n * 2:].any())
    lst = torch.arange(S, device="cuda", dtype=torch.int32).view(-1, 7)[::2].reshape(-1).contiguous()   # runs of 7 samples, every other one
    h = torch.zeros(16, S, 2, dtype=torch.float16, device="cuda")
    call("ngp_hashgrid_fwd_list", ptr(x), ptr(mn), ptr(mx), ptr(table), C.byref(meta), S, ptr(lst), lst.numel(), None, ptr(h), stream())
    assert torch.equal(h[:, lst.long()], f[:, lst.long()])
    assert float(f.float().abs().sum()) > 0


@pytest.mark.parametrize("n,n_act", [(20011, None), (40000, 17003), (40000, 31), (300, 0)])
def test_one_launch_field_backward_equals_the_two_launch_chain(lib, field, n, n_act):
    """ngp_field_bwd (round 6: colour net + density net in ONE launch, h recomputed, dL/dh handed over in registers, h / dh_scratch
    NULL) against the two mlp_bwd_kernel launches it replaces (test hook ngp_field_bwd_two_launches: colour net -> dh (S,16) in
    memory -> density net).  dfeats: bit for bit.  Colour-net dW partial rows: bit for bit (same tile -> wave mapping, same
    accumulation order).  Density-net dW: the tiles are summed by 4 waves per workgroup instead of 8 -> f32 summation order only."""
    x, d = sample_points(n, seed=41)
    feats, sig, rgb, h, dw, rw, ds = field_native(lib, field, x, d)
    g = torch.Generator().manual_seed(42)
    dsig_d = (torch.randn(n, generator=g) * 1e-3).cuda(); drgb_d = (torch.randn(n, 3, generator=g) * 1e-2).cuda().contiguous()
    active = n_active = None
    rows = n
    if n_act is not None:
        active = torch.randperm(n, generator=g)[:max(n_act, 1)].sort().values.int().cuda().contiguous()
        n_active = torch.tensor([n_act], dtype=torch.int32, device="cuda")
        rows = n_act
    n_part = lib.call("ngp_field_bwd_partials", n)
    out = {}
    for name in ("ngp_field_bwd", "ngp_field_bwd_two_launches"):
        partials = torch.full((n_part * 10240,), float("nan"), device="cuda")
        dfeats = torch.zeros(16, n, 2, dtype=torch.float16, device="cuda")
        two = name.endswith("two_launches")
        dh = torch.empty(n, 16, dtype=torch.float16, device="cuda") if two else None
        lib.call(name, lib.ptr(feats), lib.ptr(ds), lib.ptr(h) if two else None, lib.ptr(dw), lib.ptr(rw), lib.ptr(dsig_d), lib.ptr(drgb_d),
                 128.0, n, lib.ptr(active), lib.ptr(n_active), lib.ptr(dh), lib.ptr(dfeats), lib.ptr(partials), lib.stream())
        torch.cuda.synchronize()
        out[name] = (dfeats[:, :rows].cpu(), partials[:n_part * 3072].view(n_part, 3072).cpu(), partials[n_part * 3072:].view(n_part, 7168).cpu())
    (df1, pd1, pr1), (df2, pd2, pr2) = out["ngp_field_bwd"], out["ngp_field_bwd_two_launches"]
    assert torch.equal(df1, df2), "dfeats: %d elements differ" % int((df1 != df2).sum())
    assert torch.isfinite(pd1).all() and torch.isfinite(pr1).all()          # every partial row written, also by workgroups without tiles
    assert torch.equal(pr1, pr2), "colour-net dW partial rows: %d elements differ" % int((pr1 != pr2).sum())
    gd1, gd2 = pd1.double().sum(0), pd2.double().sum(0)
    if rows:
        assert float((gd1 - gd2).abs().max()) <= 2e-6 * float(gd2.abs().max()) + 1e-30
        assert float(df1.float().abs().sum()) > 0
    else:
        assert not pd1.any() and not pr1.any() and not df1.any()


def test_binned_backward_keeps_gradients_near_the_top_of_the_f16_range(lib, field):
    """Under the dynamic loss scale the feature gradients sit near the top of the f16 range, not at ~1e-5 as under the fixed 128.  The
    slice owners' fixed-point conversion went through v_cvt_i32_f32 until round 6, which saturates at |w x g| = 128: every larger
    contribution was silently clipped.  Gradients of magnitude 200 .. 30 000 against the fp32 autograd of the oracle: exact sums,
    one f16 rounding at the end (saturated at +-65504 where a sum leaves the range, never inf)."""
    meta = native_meta(lib)
    n = 6000
    x, _ = sample_points(n, seed=71)
    g = torch.Generator().manual_seed(72)
    mag = torch.exp(torch.rand(n, 32, generator=g) * (math.log(30000.0) - math.log(200.0)) + math.log(200.0))
    dfe = (mag * torch.where(torch.rand(n, 32, generator=g) < 0.5, -1.0, 1.0)).half()
    table = field.table.clone().requires_grad_(True)
    T.hash_encode(x + 0.5, table, field.meta).backward(dfe.float())
    want = table.grad
    table_abs = field.table.clone().requires_grad_(True)
    T.hash_encode(x + 0.5, table_abs, field.meta).backward(dfe.float().abs())
    mass = table_abs.grad                              # sum of |contribution| per entry: what the ORACLE's f32 summation error scales with
    xs = x.cuda().contiguous()
    dfl = dfe.view(n, 16, 2).permute(1, 0, 2).contiguous().cuda()
    mnt = torch.full((3,), -0.5, device="cuda"); mxt = torch.full((3,), 0.5, device="cuda")
    grad = torch.full((field.meta.total, 2), float("nan"), dtype=torch.float16, device="cuda")
    nbytes = lib.lib().ngp_hashgrid_bwd_binned_workspace_bytes(C.byref(meta), n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    lib.call("ngp_hashgrid_bwd_binned", lib.ptr(xs), lib.ptr(mnt), lib.ptr(mxt), lib.ptr(dfl), C.byref(meta), n, None, None,
             lib.ptr(ws), nbytes, lib.ptr(grad), lib.stream())
    got = grad.float().cpu()
    assert torch.isfinite(got).all()
    want_sat = want.clamp(-65504.0, 65504.0)
    err = (got - want_sat).abs()
    assert float(want.abs().max()) > 20000.0 and float((want.abs() > 128.0).float().mean()) > 0.001      # the case is in the data
    # exact sums + ONE f16 rounding (2^-11 of the value) on this side; f32 summation noise (1e-6 of the entry's mass) on the oracle's
    # ... and 3.6 absolute: a sample's fractional cell position differs by up to ~1.2e-4 between the two fp32 evaluations at the fine
    # levels ((x - min) * (1 / extent) * scale with scale ~2000: a few ulps of 2000), so do its corner weights, times gradients of up
    # to 30 000.  (What rounds 1-5 would have produced is 70 .. 30 000 away: asserted below.)
    bound = 6e-4 * want_sat.abs() + 2e-6 * mass + 3.6
    worst = int((err / bound).flatten().argmax())
    assert bool((err <= bound).all()), (float((err / bound).max()), float(err.max()), worst // 2, float(got.flatten()[worst]), float(want.flatten()[worst]),
                                        float(mass.flatten()[worst]), [int(o) for o in field.meta.offset[:17]])
    clipped = want.clamp(-128.0, 128.0)                # what rounds 1-5 would have produced for single large contributions: far outside the bound
    assert float(((clipped - want_sat).abs() > bound).float().mean()) > 0.001
