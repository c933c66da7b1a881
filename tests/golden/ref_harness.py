"""Runs the reference's OWN Python hot path (models/rendering.py, models/networks.py, models/custom_functions.py,
losses.py) on the CPU of this container, to generate golden vectors for the oracle (tests/golden/make_render_golden.py).
Works only where /root/reference is mounted.  The reference imports four packages that do not exist here; each gets a
stand-in that is NOT a re-implementation of the path under test:

  vren          -> the reference's own .cu kernels compiled for the host (oracle/_ref, oracle/build_ref.sh), torch
                   tensors in and out, in-place arguments updated in place like the extension does
  tinycudann    -> modules with tiny-cuda-nn's constructor/`params` contract whose arithmetic is oracle/tcnn_oracle.py;
                   outputs are rounded to f16 values like tiny-cuda-nn's but handed over as float32, which is what the
                   reference's custom_fwd(cast_inputs=float32) operators see under CUDA autocast (on the CPU that cast
                   is inactive).  tiny-cuda-nn itself is CUDA-only and absent: that half stays "parity unpinned"; what
                   this harness pins is everything the reference's Python does AROUND it
  torch_scatter -> segment_csr by index_add (only RayMarcher.backward uses it)
  kornia        -> create_meshgrid / create_meshgrid3d as documented for normalized_coordinates=False
"""
import os
import sys
import types

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import tcnn_oracle as T            # noqa: E402
from oracle.vren_oracle import Reference       # noqa: E402

REFERENCE = "/root/reference"


def _np(t, dtype=None):
    a = t.detach().cpu().numpy()
    return a if dtype is None else np.ascontiguousarray(a, dtype)


def make_vren(fma=True):
    vr = Reference(fma)
    m = types.ModuleType("vren")
    tt = torch.from_numpy

    def ray_aabb_intersect(rays_o, rays_d, centers, half_sizes, max_hits):
        return tuple(tt(a) for a in vr.ray_aabb_intersect(_np(rays_o), _np(rays_d), _np(centers), _np(half_sizes), max_hits))

    def ray_sphere_intersect(rays_o, rays_d, centers, radii, max_hits):
        return tuple(tt(a) for a in vr.ray_sphere_intersect(_np(rays_o), _np(rays_d), _np(centers), _np(radii), max_hits))

    def morton3D(coords):
        return tt(vr.morton3D(_np(coords, np.int32)))

    def morton3D_invert(indices):
        return tt(vr.morton3D_invert(_np(indices, np.int32)))

    def packbits(density_grid, density_threshold, density_bitfield):
        vr.packbits(_np(density_grid, np.float32).reshape(-1), float(density_threshold), density_bitfield.numpy())   # in place

    def raymarching_train(rays_o, rays_d, hits_t, density_bitfield, cascades, scale, exp_step_factor, noise, grid_size, max_samples):
        out = vr.raymarching_train(_np(rays_o), _np(rays_d), _np(hits_t), _np(density_bitfield), cascades, scale, exp_step_factor,
                                   _np(noise), grid_size, max_samples)
        return tuple(tt(a) for a in out)

    def raymarching_test(rays_o, rays_d, hits_t, alive_indices, density_bitfield, cascades, scale, exp_step_factor, grid_size,
                         max_samples, N_samples):
        assert hits_t.is_contiguous()
        out = vr.raymarching_test(_np(rays_o), _np(rays_d), hits_t.numpy(), _np(alive_indices), _np(density_bitfield), cascades, scale,
                                  exp_step_factor, grid_size, max_samples, N_samples)     # hits_t advanced in place
        return tuple(tt(a) for a in out)

    def composite_train_fw(sigmas, rgbs, deltas, ts, rays_a, T_threshold):
        return tuple(tt(a) for a in vr.composite_train_fw(_np(sigmas, np.float32), _np(rgbs, np.float32), _np(deltas), _np(ts), _np(rays_a),
                                                          T_threshold))

    def composite_train_bw(dL_dopacity, dL_ddepth, dL_drgb, dL_dws, sigmas, rgbs, ws, deltas, ts, rays_a, opacity, depth, rgb, T_threshold):
        return tuple(tt(a) for a in vr.composite_train_bw(*[_np(a) for a in (dL_dopacity, dL_ddepth, dL_drgb, dL_dws, sigmas, rgbs, ws, deltas,
                                                                            ts, rays_a, opacity, depth, rgb)], T_threshold))

    def composite_test_fw(sigmas, rgbs, deltas, ts, hits_t, alive_indices, T_threshold, N_eff_samples, opacity, depth, rgb):
        for a in (alive_indices, opacity, depth, rgb):
            assert a.is_contiguous()
        vr.composite_test_fw(_np(sigmas, np.float32), _np(rgbs, np.float32), _np(deltas), _np(ts), hits_t.numpy(), alive_indices.numpy(),
                             T_threshold, _np(N_eff_samples), opacity.numpy(), depth.numpy(), rgb.numpy())   # alive/opacity/depth/rgb in place

    def distortion_loss_fw(ws, deltas, ts, rays_a):
        return tuple(tt(a) for a in vr.distortion_loss_fw(_np(ws), _np(deltas), _np(ts), _np(rays_a)))

    def distortion_loss_bw(dL_dloss, ws_inclusive_scan, wts_inclusive_scan, ws, deltas, ts, rays_a):
        return tt(vr.distortion_loss_bw(*[_np(a) for a in (dL_dloss, ws_inclusive_scan, wts_inclusive_scan, ws, deltas, ts, rays_a)]))

    for f in (ray_aabb_intersect, ray_sphere_intersect, morton3D, morton3D_invert, packbits, raymarching_train, raymarching_test,
              composite_train_fw, composite_train_bw, composite_test_fw, distortion_loss_fw, distortion_loss_bw):
        setattr(m, f.__name__, f)
    return m


def make_tinycudann():
    m = types.ModuleType("tinycudann")

    class NetworkWithInputEncoding(nn.Module):
        def __init__(self, n_input_dims, n_output_dims, encoding_config, network_config):
            super().__init__()
            ec, nc = encoding_config, network_config
            assert n_input_dims == 3 and ec["otype"] == "Grid" and ec["type"] == "Hash" and nc["n_neurons"] == 64
            self.meta = T.GridMeta(ec["n_levels"], ec["n_features_per_level"], ec["log2_hashmap_size"], ec["base_resolution"],
                                   float(ec["per_level_scale"]))
            self.n_in, self.n_hidden, self.n_out = ec["n_levels"] * ec["n_features_per_level"], nc["n_hidden_layers"], n_output_dims
            self.n_mlp = self.n_in * 64 + 64 * 64 * (self.n_hidden - 1) + 64 * 16
            self.params = nn.Parameter(torch.zeros(self.n_mlp + self.meta.total * 2))

        def forward(self, x):
            table = T.q16(self.params[self.n_mlp:].view(-1, 2))
            feats = T.hash_encode(x.float(), table, self.meta, True)
            return T.q16(T.mlp(feats, self.params[:self.n_mlp], self.n_in, self.n_hidden, self.n_out, "None", True))

    class Encoding(nn.Module):
        def __init__(self, n_input_dims, encoding_config):
            super().__init__()
            assert encoding_config["otype"] == "SphericalHarmonics" and encoding_config["degree"] == 4

        def forward(self, x):
            return T.q16(T.sh4(x.float() * 2 - 1))                 # tiny-cuda-nn maps its [0,1] input back to [-1,1]

    class Network(nn.Module):
        def __init__(self, n_input_dims, n_output_dims, network_config):
            super().__init__()
            nc = network_config
            self.n_in, self.n_out, self.n_hidden, self.act = n_input_dims, n_output_dims, nc["n_hidden_layers"], nc["output_activation"]
            self.params = nn.Parameter(torch.zeros(self.n_in * 64 + 64 * 64 * (self.n_hidden - 1) + 64 * 16))

        def forward(self, x):
            return T.q16(T.mlp(x.float(), self.params, self.n_in, self.n_hidden, self.n_out, self.act, True))

    m.NetworkWithInputEncoding, m.Encoding, m.Network = NetworkWithInputEncoding, Encoding, Network
    return m


def make_torch_scatter():
    m = types.ModuleType("torch_scatter")

    def segment_csr(src, indptr, reduce="sum"):
        counts = indptr[1:] - indptr[:-1]
        owner = torch.repeat_interleave(torch.arange(len(counts)), counts)
        return torch.zeros((len(counts),) + tuple(src.shape[1:]), dtype=src.dtype).index_add_(0, owner, src)
    m.segment_csr = segment_csr
    return m


def make_kornia():
    m = types.ModuleType("kornia")

    def create_meshgrid(height, width, normalized_coordinates=True, device=None, dtype=torch.float32):
        assert not normalized_coordinates
        gy, gx = torch.meshgrid(torch.arange(height, dtype=dtype), torch.arange(width, dtype=dtype), indexing="ij")
        return torch.stack([gx, gy], -1).unsqueeze(0)

    def create_meshgrid3d(depth, height, width, normalized_coordinates=True, device=None, dtype=torch.float32):
        assert not normalized_coordinates
        gz, gy, gx = torch.meshgrid(torch.arange(depth, dtype=dtype), torch.arange(height, dtype=dtype), torch.arange(width, dtype=dtype),
                                    indexing="ij")
        return torch.stack([gx, gy, gz], -1).unsqueeze(0)           # (1, D, H, W, 3), xyz last
    m.create_meshgrid, m.create_meshgrid3d = create_meshgrid, create_meshgrid3d
    utils = types.ModuleType("kornia.utils"); grid = types.ModuleType("kornia.utils.grid")
    grid.create_meshgrid3d = create_meshgrid3d; utils.grid = grid; m.utils = utils
    return m, utils, grid


def load_reference(fma=True):
    """Returns (networks, rendering, losses) modules of the reference, imported over the stand-ins."""
    kornia, kutils, kgrid = make_kornia()
    sys.modules.update({"vren": make_vren(fma), "tinycudann": make_tinycudann(), "torch_scatter": make_torch_scatter(),
                        "kornia": kornia, "kornia.utils": kutils, "kornia.utils.grid": kgrid})
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    import importlib
    networks = importlib.import_module("models.networks")
    rendering = importlib.import_module("models.rendering")
    losses = importlib.import_module("losses")
    return networks, rendering, losses


def load_field_params(model, field):
    """Give the reference's NGP the parameters of an oracle Field (tiny-cuda-nn layout: MLP weights, then the table)."""
    with torch.no_grad():
        model.xyz_encoder.params.copy_(torch.cat([field.density_w, field.table.reshape(-1)]))
        model.rgb_net.params.copy_(field.rgb_w)
