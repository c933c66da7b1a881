"""The tiny-cuda-nn pin, once it exists: tests/golden/tcnn_golden.npz is written by tests/golden/make_tcnn_golden.py on a machine that
has the reference's real `tinycudann` (it cannot exist in this environment: SURVEY.md 8(c), DESIGN.md section 2 "parity unpinned").
While the file is absent these tests are SKIPPED and the tiny-cuda-nn half of the oracle stays unpinned; with it, the oracle
(CPU) and the native kernels (GPU) are held to tiny-cuda-nn's own outputs on identical parameters and inputs, at the tolerances
SURVEY.md 8(c) adopts (features / h 2e-3 rel of the largest entry, rgb 1e-3 abs -- f16 outputs -- and gradients 2 % of the largest)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden", "tcnn_golden.npz")
needs_golden = pytest.mark.skipif(not os.path.exists(GOLDEN), reason="tests/golden/tcnn_golden.npz absent: run tests/golden/make_tcnn_golden.py "
                                  "where the real tinycudann is installed (parity of the tiny-cuda-nn half stays unpinned until then)")


def _recipe():
    spec = importlib.util.spec_from_file_location("make_tcnn_golden", os.path.join(HERE, "golden", "make_tcnn_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _close(got, want, rel, what):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)
    assert err <= rel, "%s: %.3g of the largest entry (allowed %.3g)" % (what, err, rel)


def test_the_recipe_regenerates_its_inputs_deterministically():
    """(runs everywhere) both sides of the pin rebuild parameters and inputs from seeds: the generators must be reproducible."""
    r = _recipe()
    a, b = r.make_inputs(), r.make_inputs()
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and a[0].shape == (r.N, 3) and a[0].min() >= 0 and a[0].max() <= 1
    assert np.array_equal(r.make_params(5000, 3072, 1), r.make_params(5000, 3072, 1))
    enc, net1, net2 = r.configs(0.5)
    assert abs(enc["per_level_scale"] - 1.3195079) < 1e-6 and net2["n_hidden_layers"] == 2


def _level_table(gold):
    n = int(gold["s0.5_xyz_params_len"][0])
    assert n in (11448112, 11423136), "xyz_encoder.params has %d entries: neither the float32 nor the exact level table (DESIGN.md section 2)" % n
    return "float32" if n == 11448112 else "exact"


@needs_golden
def test_oracle_against_tiny_cuda_nn():
    from oracle import tcnn_oracle as T
    r = _recipe()
    gold = np.load(GOLDEN)
    x01, d, feat32, seed_h, seed_sh, seed_rgb = r.make_inputs()
    assert np.array_equal(x01, gold["x01"]) and np.array_equal(feat32, gold["feat32"])
    f = T.Field(scale=0.5, exact_levels=_level_table(gold) == "exact")
    n_mlp = 3072
    p = torch.from_numpy(r.make_params(int(gold["s0.5_xyz_params_len"][0]), n_mlp, r.SEED_PARAMS))
    assert f.meta.total * 2 + n_mlp == p.numel()
    f.density_w = p[:n_mlp].clone().requires_grad_(True)
    f.table = p[n_mlp:].view(-1, 2).clone().requires_grad_(True)
    x = (torch.from_numpy(x01) - 0.5).requires_grad_(True)                   # x01 = (x + 0.5) / 1 at scale 0.5
    _, h, _ = f.density(x, quantize=True)
    _close(h.detach().numpy(), gold["h"].astype(np.float32), 2e-3, "xyz_encoder output")
    h.backward(torch.from_numpy(seed_h))
    _close(f.density_w.grad.numpy(), gold["xyz_grad_mlp"], 2e-2, "density-net weight gradient")
    gg = f.table.grad.reshape(-1).numpy()
    _close(gg[gold["xyz_grad_grid_pick_idx"]], gold["xyz_grad_grid_pick"], 2e-2, "grid gradient (4096 entries)")
    _close(T.sh4(torch.from_numpy(d)).numpy(), gold["sh"].astype(np.float32), 2e-3, "SH degree 4")
    f.rgb_w = torch.from_numpy(r.make_params(int(gold["rgb_params_len"][0]), int(gold["rgb_params_len"][0]), r.SEED_PARAMS + 1))
    c = T.mlp(T.q16(torch.from_numpy(feat32)), f.rgb_w, 32, 2, 3, "Sigmoid", True, False)
    assert np.abs(c.detach().numpy()[:, :3] - gold["rgb"].astype(np.float32)).max() <= 1e-3


@needs_golden
@pytest.mark.gpu
def test_native_modules_against_tiny_cuda_nn():
    from ngp_pl_amd import tcnn
    r = _recipe()
    gold = np.load(GOLDEN)
    x01, d, feat32, seed_h, seed_sh, seed_rgb = r.make_inputs()
    enc, net1, net2 = r.configs(0.5)
    table = _level_table(gold)
    xyz = tcnn.NetworkWithInputEncoding(3, 16, enc, net1, level_table=table).cuda()
    assert xyz.params.numel() == int(gold["s0.5_xyz_params_len"][0])
    enc16, _, _ = r.configs(16.0)
    assert tcnn.NetworkWithInputEncoding(3, 16, enc16, net1, level_table=table).params.numel() == int(gold["s16_xyz_params_len"][0])
    with torch.no_grad():
        xyz.params.copy_(torch.from_numpy(r.make_params(xyz.params.numel(), 3072, r.SEED_PARAMS)).cuda())
    xin = torch.from_numpy(x01).cuda().requires_grad_(True)
    h = xyz(xin)
    _close(h.detach().float().cpu().numpy(), gold["h"].astype(np.float32), 2e-3, "xyz_encoder output")
    h.backward(torch.from_numpy(seed_h).cuda().to(h.dtype))
    gp = xyz.params.grad.float().cpu().numpy()
    _close(gp[:3072], gold["xyz_grad_mlp"], 2e-2, "density-net weight gradient")
    _close(gp[3072:][gold["xyz_grad_grid_pick_idx"]], gold["xyz_grad_grid_pick"], 2e-2, "grid gradient (4096 entries)")
    _close(xin.grad.float().cpu().numpy(), gold["xyz_grad_x"], 2e-2, "input gradient")
    sh = tcnn.Encoding(3, {"otype": "SphericalHarmonics", "degree": 4}).cuda()
    assert sh.params.numel() == int(gold["sh_params_len"][0])
    y = sh((torch.from_numpy(d).cuda() + 1) / 2)
    _close(y.float().cpu().numpy(), gold["sh"].astype(np.float32), 2e-3, "SH degree 4")
    rgb = tcnn.Network(32, 3, net2).cuda()
    assert rgb.params.numel() == int(gold["rgb_params_len"][0])
    with torch.no_grad():
        rgb.params.copy_(torch.from_numpy(r.make_params(rgb.params.numel(), rgb.params.numel(), r.SEED_PARAMS + 1)).cuda())
    fin = torch.from_numpy(feat32).cuda().requires_grad_(True)
    c = rgb(fin)
    assert np.abs(c.detach().float().cpu().numpy() - gold["rgb"].astype(np.float32)).max() <= 1e-3
    c.backward(torch.from_numpy(seed_rgb).cuda().to(c.dtype))
    _close(rgb.params.grad.float().cpu().numpy(), gold["rgb_grad_params"], 2e-2, "colour-net weight gradient")
    _close(fin.grad.float().cpu().numpy(), gold["rgb_grad_in"], 2e-2, "colour-net input gradient")
