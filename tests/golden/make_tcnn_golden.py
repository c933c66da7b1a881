"""PIN RECIPE for the tiny-cuda-nn half of the oracle ("parity unpinned" -- tiny-cuda-nn is an un-vendored,
unpinned, CUDA-only dependency of the reference, README.md:39; it cannot be installed in the development container or on the
MI355X boxes).  Run this ONCE on any machine that has the reference's real dependency:

    pip install git+https://github.com/NVlabs/tiny-cuda-nn/#subdirectory=bindings/torch     (README.md:39)
    python tests/golden/make_tcnn_golden.py            # writes tests/golden/tcnn_golden.npz  (~1 MB)

and commit the file.  tests/test_tcnn_golden_gpu.py (skipped while the file is absent) then holds the native kernels -- and
oracle/tcnn_oracle.py (tests/test_tcnn_golden_cpu.py) -- to tiny-cuda-nn's own outputs on identical parameters and inputs:
  * `params` lengths of the three modules the reference instantiates (models/networks.py:36-77) at scale 0.5 and scale 16: the
    known-answer check that decides between the float32 level table (11 448 112 at scale 0.5) and the exact one (11 423 136), DESIGN.md section 2;
  * xyz_encoder (hash grid -> 32 -> 64 -> 16), dir_encoder (SH degree 4), rgb_net (32 -> 64 -> 64 -> 3, sigmoid) forward outputs;
  * their backward: input gradients and parameter gradients for fixed output seeds (the grid gradient as a digest: sum, |sum|,
    and 4096 fixed entries -- the table has 11.4 M parameters).
Parameters are NOT stored: both sides regenerate them from numpy RandomState seeds recorded in the file (layout assumed:
[MLP weights layer by layer, (out, in) row-major | grid entries level-major], SURVEY.md 8(a) [3P]; the lengths above verify the
split).  Everything is float32 / float16 numpy in the file; no pickles."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "tcnn_golden.npz")
N = 4096                      # samples (tiny-cuda-nn pads batches to multiples of 128 itself)
SEED_PARAMS, SEED_INPUTS = 4242, 4243


def configs(scale):
    L, F, log2_T, N_min = 16, 2, 19, 16
    b = float(np.exp(np.log(2048 * scale / N_min) / (L - 1)))                  # networks.py:33
    enc = {"otype": "Grid", "type": "Hash", "n_levels": L, "n_features_per_level": F, "log2_hashmap_size": log2_T,
           "base_resolution": N_min, "per_level_scale": b, "interpolation": "Linear"}
    net1 = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 1}
    net2 = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "Sigmoid", "n_neurons": 64, "n_hidden_layers": 2}
    return enc, net1, net2


def make_params(n, n_mlp, seed):
    """The parameter vector both sides use: MLP block uniform in +-0.25 (trained-like magnitudes), grid block uniform in +-0.5."""
    g = np.random.RandomState(seed)
    p = np.empty(n, np.float32)
    p[:n_mlp] = g.uniform(-0.25, 0.25, n_mlp).astype(np.float32)
    if n > n_mlp:
        p[n_mlp:] = g.uniform(-0.5, 0.5, n - n_mlp).astype(np.float32)
    return p


def make_inputs():
    g = np.random.RandomState(SEED_INPUTS)
    x01 = g.uniform(0.0, 1.0, (N, 3)).astype(np.float32)                       # (x - xyz_min) / (xyz_max - xyz_min), networks.py:103
    d = g.normal(size=(N, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    feat32 = g.uniform(-1.0, 1.0, (N, 32)).astype(np.float32)                  # rgb_net input: cat([sh, h]) stand-in
    seed_h = g.normal(size=(N, 16)).astype(np.float32) * 1e-2                  # dL/dy seeds
    seed_rgb = g.normal(size=(N, 3)).astype(np.float32) * 1e-2
    seed_sh = g.normal(size=(N, 16)).astype(np.float32) * 1e-2
    return x01, d, feat32, seed_h, seed_sh, seed_rgb


def main():
    try:
        import torch
        import tinycudann as tcnn
    except ImportError as e:
        sys.exit("this recipe needs the reference's real dependency (tinycudann on an NVIDIA GPU): %s" % e)
    dev = "cuda"
    out = {"n": np.array([N]), "seed_params": np.array([SEED_PARAMS]), "seed_inputs": np.array([SEED_INPUTS]),
           "tcnn_version": np.frombuffer(str(getattr(tcnn, "__version__", "unknown")).encode(), np.uint8)}
    x01, d, feat32, seed_h, seed_sh, seed_rgb = make_inputs()
    for name, arr in (("x01", x01), ("d", d), ("feat32", feat32), ("seed_h", seed_h), ("seed_sh", seed_sh), ("seed_rgb", seed_rgb)):
        out[name] = arr
    for scale in (0.5, 16.0):
        enc, net1, net2 = configs(scale)
        tag = "s%g" % scale
        xyz = tcnn.NetworkWithInputEncoding(n_input_dims=3, n_output_dims=16, encoding_config=enc, network_config=net1)
        out[tag + "_xyz_params_len"] = np.array([xyz.params.numel()])
        if scale != 0.5:
            continue
        n_mlp = 32 * 64 + 64 * 16
        p = make_params(xyz.params.numel(), n_mlp, SEED_PARAMS)
        with torch.no_grad():
            xyz.params.copy_(torch.from_numpy(p).to(dev))
        xin = torch.from_numpy(x01).to(dev).requires_grad_(True)
        h = xyz(xin)
        out["h"] = h.detach().float().cpu().numpy().astype(np.float16)
        h.backward(torch.from_numpy(seed_h).to(dev).to(h.dtype))
        gp = xyz.params.grad.detach().float().cpu().numpy()
        out["xyz_grad_mlp"] = gp[:n_mlp]
        gg = gp[n_mlp:]
        pick = np.random.RandomState(7).randint(0, gg.size, 4096)
        out["xyz_grad_grid_pick_idx"] = pick.astype(np.int64)
        out["xyz_grad_grid_pick"] = gg[pick]
        out["xyz_grad_grid_digest"] = np.array([gg.astype(np.float64).sum(), np.abs(gg.astype(np.float64)).sum(), float((gg != 0).sum())])
        out["xyz_grad_x"] = xin.grad.detach().float().cpu().numpy()
        # direction encoding (networks.py:58-65, called with (d + 1) / 2: :143-144)
        sh = tcnn.Encoding(n_input_dims=3, encoding_config={"otype": "SphericalHarmonics", "degree": 4})
        out["sh_params_len"] = np.array([sh.params.numel()])
        din = ((torch.from_numpy(d).to(dev) + 1) / 2).requires_grad_(True)
        y = sh(din)
        out["sh"] = y.detach().float().cpu().numpy().astype(np.float16)
        y.backward(torch.from_numpy(seed_sh).to(dev).to(y.dtype))
        out["sh_grad_d01"] = din.grad.detach().float().cpu().numpy()
        # colour network (networks.py:67-77)
        rgb = tcnn.Network(n_input_dims=32, n_output_dims=3, network_config=net2)
        out["rgb_params_len"] = np.array([rgb.params.numel()])
        pr = make_params(rgb.params.numel(), rgb.params.numel(), SEED_PARAMS + 1)
        with torch.no_grad():
            rgb.params.copy_(torch.from_numpy(pr).to(dev))
        fin = torch.from_numpy(feat32).to(dev).requires_grad_(True)
        c = rgb(fin)
        out["rgb"] = c.detach().float().cpu().numpy().astype(np.float16)
        c.backward(torch.from_numpy(seed_rgb).to(dev).to(c.dtype))
        out["rgb_grad_params"] = rgb.params.grad.detach().float().cpu().numpy()
        out["rgb_grad_in"] = fin.grad.detach().float().cpu().numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: (v.shape, str(v.dtype)) for k, v in out.items()})


if __name__ == "__main__":
    main()
