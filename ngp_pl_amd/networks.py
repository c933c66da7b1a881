"""`NGP` field model of the reference (/root/reference/models/networks.py:12-269): same constructor,
buffers, state-dict keys and methods, with the tiny-cuda-nn modules replaced by `ngp_pl_amd.tcnn`
and the per-sample hot path (hash grid -> density MLP -> TruncExp -> SH -> rgb MLP) fused into one
autograd node that calls the gfx950 kernels directly.

state_dict keys (utils.py:4-39 relies on them): xyz_encoder.params, dir_encoder.params,
rgb_net.params, tonemapper_net_{0,1,2}.params, center, xyz_min, xyz_max, half_size,
density_bitfield (+ density_grid, grid_coords once `register_training_buffers()` ran, as
train.py:73-76 does).
"""
import ctypes as C

import os

import numpy as np
import torch
from torch import nn

from . import _lib, tcnn, vren
from ._lib import call, ptr, stream
from .custom_functions import TruncExp

NEAR_DISTANCE = 0.01   # rendering.py:8 (imported from there by the reference, networks.py:9)


class _FusedField(torch.autograd.Function):
    """(x, d) -> (sigma f32 (S), rgb f32 (S,3)); NGP.forward for rgb_act == 'Sigmoid'."""

    @staticmethod
    def forward(ctx, x, d, enc_params, rgb_params, model):
        x = x.detach().float().contiguous(); d = d.detach().float().contiguous()
        _lib.require_cuda(x, d)
        n, dev = x.shape[0], x.device
        enc, net = model.xyz_encoder, model.rgb_net
        eh, rh = enc._half.get(enc_params), net._half.get(rgb_params)
        feats = torch.empty(16, n, 2, dtype=torch.float16, device=dev)
        h = torch.empty(n if _lib.field_bwd_uses_h() else 0, 16, dtype=torch.float16, device=dev)    # (the one-launch backward recomputes it)
        sigmas = torch.empty(n, dtype=torch.float32, device=dev)
        rgbs = torch.empty(n, 3, dtype=torch.float32, device=dev)
        if n > 0:
            with torch.cuda.device(dev):
                call("ngp_hashgrid_fwd", ptr(x), ptr(model.xyz_min), ptr(model.xyz_max), ptr(eh[enc.n_mlp:]),
                     C.byref(enc.meta), n, ptr(feats), stream())
                call("ngp_field_fwd", ptr(feats), ptr(d), ptr(eh), ptr(rh), n, ptr(sigmas), ptr(rgbs), ptr(h) if h.numel() else None, stream())
        ctx.model = model
        ctx.save_for_backward(x, d, feats, h)
        return sigmas, rgbs

    @staticmethod
    def backward(ctx, dL_dsigmas, dL_drgbs):
        model = ctx.model
        x, d, feats, h = ctx.saved_tensors
        enc, net = model.xyz_encoder, model.rgb_net
        n, dev = x.shape[0], x.device
        scale = tcnn.LOSS_SCALE
        if n == 0:
            return None, None, torch.zeros_like(enc.params), torch.zeros_like(net.params), None
        dL_dsigmas = dL_dsigmas.float().contiguous(); dL_drgbs = dL_drgbs.float().contiguous()
        eh, rh = enc._half.get(enc.params), net._half.get(net.params)
        with torch.cuda.device(dev):
            n_part = call("ngp_field_bwd_partials", n)
            partials = torch.empty(n_part * (enc.n_mlp + net.params.numel()), dtype=torch.float32, device=dev)
            dh = torch.empty(n, 16, dtype=torch.float16, device=dev) if h.numel() else None
            dfeats = torch.empty(16, n, 2, dtype=torch.float16, device=dev)
            call("ngp_field_bwd", ptr(feats), ptr(d), ptr(h) if h.numel() else None, ptr(eh), ptr(rh), ptr(dL_dsigmas), ptr(dL_drgbs), scale, n,
                 None, None, ptr(dh), ptr(dfeats), ptr(partials), stream())
            g16 = model._grid_grad16(dev)
            tcnn.grid_backward(x, model.xyz_min, model.xyz_max, dfeats, enc.meta, n, g16)
            p_density = partials[:n_part * enc.n_mlp]
            p_rgb = partials[n_part * enc.n_mlp:]
            if model.native_grads:
                # hand the native buffers to ngp_pl_amd.optim.FusedAdam (no f32 materialisation)
                model.hand_over_native(dict(grid16=g16, density_partials=p_density, rgb_partials=p_rgb, n_partials=n_part, scale=scale))
                return None, None, None, None, None
            g_enc = torch.empty_like(enc.params)
            g_enc[:enc.n_mlp] = tcnn.reduce_partials(p_density, n_part, enc.n_mlp) / scale
            call("ngp_cast_f16_to_f32", ptr(g16), enc.n_grid, 1.0 / scale, ptr(g_enc[enc.n_mlp:]), stream())
            g_rgb = tcnn.reduce_partials(p_rgb, n_part, net.params.numel()) / scale
        return None, None, g_enc, g_rgb, None


class NGP(nn.Module):
    def __init__(self, scale, rgb_act="Sigmoid", level_table="float32"):
        """`level_table`: arithmetic of the hash grid's level table (tcnn.make_grid_meta); "exact" is the opt-in for
        checkpoints whose `xyz_encoder.params` has the exact-arithmetic length (11 423 136 at scale 0.5)."""
        super().__init__()
        self.rgb_act = rgb_act
        # scene bounding box (networks.py:19-23)
        self.scale = scale
        self.register_buffer("center", torch.zeros(1, 3))
        self.register_buffer("xyz_min", -torch.ones(1, 3) * scale)
        self.register_buffer("xyz_max", torch.ones(1, 3) * scale)
        self.register_buffer("half_size", (self.xyz_max - self.xyz_min) / 2)
        # cascade k covers [-2^(k-1), 2^(k-1)]^3 (networks.py:25-29)
        self.cascades = max(1 + int(np.ceil(np.log2(2 * scale))), 1)
        self.grid_size = 128
        self.register_buffer("density_bitfield", torch.zeros(self.cascades * self.grid_size ** 3 // 8, dtype=torch.uint8))

        L, F, log2_T, N_min = 16, 2, 19, 16
        b = np.exp(np.log(2048 * scale / N_min) / (L - 1))
        self.xyz_encoder = tcnn.NetworkWithInputEncoding(
            n_input_dims=3, n_output_dims=16,
            encoding_config={"otype": "Grid", "type": "Hash", "n_levels": L, "n_features_per_level": F,
                             "log2_hashmap_size": log2_T, "base_resolution": N_min, "per_level_scale": b,
                             "interpolation": "Linear"},
            network_config={"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None",
                            "n_neurons": 64, "n_hidden_layers": 1}, level_table=level_table)
        self.dir_encoder = tcnn.Encoding(n_input_dims=3, encoding_config={"otype": "SphericalHarmonics", "degree": 4})
        self.rgb_net = tcnn.Network(
            n_input_dims=32, n_output_dims=3,
            network_config={"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": self.rgb_act,
                            "n_neurons": 64, "n_hidden_layers": 2})
        if self.rgb_act == "None":          # HDR branch: per-channel tonemappers (networks.py:79-92)
            for i in range(3):
                setattr(self, f"tonemapper_net_{i}", tcnn.Network(
                    n_input_dims=1, n_output_dims=1,
                    network_config={"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "Sigmoid",
                                    "n_neurons": 64, "n_hidden_layers": 1}))
        # fused-path switches
        self.fused = True            # one autograd node for the whole field (False: module by module, as the reference)
        self.native_grads = False    # leave gradients in native f16/partial buffers for optim.FusedAdam
        # render()'s native training node (native_grads) runs its f16 backward under a dynamic loss scale = GradScaler's rule on the
        # device, on top of tiny-cuda-nn's 128 (what Lightning's precision=16 gives the reference, train.py:274; None: the fixed 128 alone)
        self.native_loss_scaler = dict(init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000)
        self.sync_free_sampling = True   # occupancy-cell sampling without the reference's nonzero() host sync
        self._native = None
        self._g16 = None
        from .optim import tag_parameters
        tag_parameters(self)         # an optimizer built from bare parameter tensors (train.py:123-131) finds this model through them

    # -- helpers -------------------------------------------------------------------------------
    def hand_over_native(self, record):
        """A backward of the fused field leaves its gradients in native buffers for optim.FusedAdam (native_grads=True).
        The record holds ONE backward: a second one before the optimizer consumed the first (gradient accumulation, two
        render() calls per step) would silently replace it -- refuse instead."""
        if self._native is not None:
            raise RuntimeError("the native gradient record of the previous backward has not been consumed by FusedAdam.step(): "
                               "gradient accumulation over several backward passes needs model.native_grads = False "
                               "(f32 .grad tensors, any torch optimizer)")
        self._native = record

    def _grid_grad16(self, dev):
        if self._g16 is None or self._g16.device != dev:
            self._g16 = torch.empty(self.xyz_encoder.n_grid, dtype=torch.float16, device=dev)
        return self._g16

    def register_training_buffers(self):
        """density_grid / grid_coords exactly as train.py:73-76 registers them on the model."""
        G = self.grid_size
        dev = self.density_bitfield.device
        self.register_buffer("density_grid", torch.zeros(self.cascades, G ** 3, device=dev))
        ax = torch.arange(G, dtype=torch.int32, device=dev)
        # kornia.create_meshgrid3d(G,G,G,False).reshape(-1,3): last dim (x,y,z) with x fastest
        zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
        self.register_buffer("grid_coords", torch.stack([xx, yy, zz], -1).reshape(-1, 3).contiguous())

    # -- field ---------------------------------------------------------------------------------
    def density(self, x, return_feat=False):
        """x (N,3) in [-scale,scale] -> sigmas (N) [, h (N,16) f16]   (networks.py:94-107)."""
        if not torch.is_grad_enabled() or not self.xyz_encoder.params.requires_grad:
            return self._density_nograd(x, return_feat)
        x = (x - self.xyz_min) / (self.xyz_max - self.xyz_min)
        h = self.xyz_encoder(x)
        sigmas = TruncExp.apply(h[:, 0])
        return (sigmas, h) if return_feat else sigmas

    @torch.no_grad()
    def _density_nograd(self, x, return_feat=False):
        x = x.float().contiguous()
        _lib.require_cuda(x)
        n, dev = x.shape[0], x.device
        enc = self.xyz_encoder
        eh = enc._half.get(enc.params)
        feats = torch.empty(16, n, 2, dtype=torch.float16, device=dev)
        sigmas = torch.empty(n, dtype=torch.float32, device=dev)
        h = torch.empty(n, 16, dtype=torch.float16, device=dev) if return_feat else None
        if n > 0:
            with torch.cuda.device(dev):
                call("ngp_hashgrid_fwd", ptr(x), ptr(self.xyz_min), ptr(self.xyz_max), ptr(eh[enc.n_mlp:]), C.byref(enc.meta), n,
                     ptr(feats), stream())
                call("ngp_density_fwd", ptr(feats), ptr(eh), n, ptr(sigmas), ptr(h), stream())
        return (sigmas, h) if return_feat else sigmas

    def log_radiance_to_rgb(self, log_radiances, **kwargs):
        """HDR-NeRF tonemapping, rgb_act == 'None' only (networks.py:109-130)."""
        log_exposure = torch.log(kwargs["exposure"]) if "exposure" in kwargs else 0
        out = [getattr(self, f"tonemapper_net_{i}")(log_radiances[:, i:i + 1] + log_exposure) for i in range(3)]
        return torch.cat(out, 1)

    def forward(self, x, d, **kwargs):
        """x (N,3), d (N,3) -> sigmas (N), rgbs (N,3)   (networks.py:132-153)."""
        rays_need_grad = torch.is_grad_enabled() and (x.requires_grad or d.requires_grad)     # pose optimisation (train.py:86-89)
        if self.fused and self.rgb_act == "Sigmoid" and not rays_need_grad:
            return _FusedField.apply(x, d, self.xyz_encoder.params, self.rgb_net.params, self)
        sigmas, h = self.density(x, return_feat=True)
        d = d / torch.norm(d, dim=1, keepdim=True)
        d = self.dir_encoder((d + 1) / 2)
        rgbs = self.rgb_net(torch.cat([d, h], 1))
        if self.rgb_act == "None":
            if kwargs.get("output_radiance", False):
                rgbs = TruncExp.apply(rgbs)
            else:
                rgbs = self.log_radiance_to_rgb(rgbs, **kwargs)
        return sigmas, rgbs

    # -- occupancy grid ------------------------------------------------------------------------
    @torch.no_grad()
    def get_all_cells(self):
        """[(indices, coords)] * cascades over the whole grid (networks.py:155-167)."""
        indices = vren.morton3D(self.grid_coords).long()
        return [(indices, self.grid_coords)] * self.cascades

    @torch.no_grad()
    def sample_uniform_and_occupied_cells(self, M, density_threshold):
        """Per cascade: M uniform cells + M cells drawn from those above the threshold
        (networks.py:169-195)."""
        cells = []
        dev = self.density_grid.device
        for c in range(self.cascades):
            coords1 = torch.randint(self.grid_size, (M, 3), dtype=torch.int32, device=dev)
            indices1 = vren.morton3D(coords1).long()
            occ = self.density_grid[c] > density_threshold
            if self.sync_free_sampling:
                # the reference's nonzero() + randint(len) costs a host sync (and a pipeline drain)
                # every update; the same draw -- uniform over the occupied cells, with replacement --
                # by inverse-CDF lookup stays on the device.  Empty grid: rank 0 maps past the end,
                # clamped to the last cell (the reference adds no occupied samples then).
                csum = torch.cumsum(occ, 0, dtype=torch.int32)
                rank = (torch.rand(M, device=dev) * csum[-1]).to(torch.int32)
                indices2 = torch.searchsorted(csum, rank, right=True).clamp_(max=occ.numel() - 1)
            else:
                indices2 = torch.nonzero(occ)[:, 0]
                if len(indices2) > 0:
                    indices2 = indices2[torch.randint(len(indices2), (M,), device=dev)]
            coords2 = vren.morton3D_invert(indices2.int())
            cells.append((torch.cat([indices1, indices2]), torch.cat([coords1, coords2])))
        return cells

    @torch.no_grad()
    def mark_invisible_cells(self, K, poses, img_wh, chunk=64 ** 3):
        """density_grid = -1 for cells no training camera sees or that are closer than the near plane to one, count_grid = the
        fraction of cameras covering a cell; run once before training (networks.py:197-238, train.py:155-158).  One launch
        (`ngp_mark_invisible_cells`: a thread per cell walks the cameras, which pass through LDS 1024 at a time -- any number of
        training cameras).  GPU only, like every other operator of this package; `chunk` is accepted for the reference's
        signature and unused (the kernel has no temporaries to bound)."""
        n_cams = poses.shape[0]
        _lib.require_cuda(self.density_grid)
        self.count_grid = torch.zeros_like(self.density_grid)
        dev = self.density_grid.device
        Kd = K.to(device=dev, dtype=torch.float32).contiguous()
        Pd = poses[:, :3, :4].to(device=dev, dtype=torch.float32).contiguous()
        with torch.cuda.device(dev):
            call("ngp_mark_invisible_cells", ptr(Kd), ptr(Pd), n_cams, int(img_wh[0]), int(img_wh[1]), NEAR_DISTANCE,
                 self.cascades, self.grid_size, float(self.scale), ptr(self.count_grid), ptr(self.density_grid), stream())

    @torch.no_grad()
    def _update_density_grid_native(self, density_threshold, warmup, decay, erode):
        """The whole update as one library call (`ngp_occupancy_update`): cell sampling, jitter,
        density-only forward with scatter epilogue, merge, mean, bit packing -- 7 launches per
        cascade instead of ~100 torch ops, no host sync.  Same distribution of sampled cells as
        the torch path below; the random stream is the library's counter-based one."""
        dev = self.density_grid.device
        enc = self.xyz_encoder
        eh = enc._half.get(enc.params)
        lib = _lib.lib()
        nbytes = lib.ngp_occupancy_update_workspace_bytes(self.cascades, self.grid_size)
        ws = getattr(self, "_occ_ws", None)
        if ws is None or ws.numel() < nbytes or ws.device != dev:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self._occ_ws = ws
        decay_grid = None
        if erode:
            decay_grid = torch.clamp(decay ** (1 / self.count_grid), 0.1, 0.95).contiguous()
        self._occ_updates = getattr(self, "_occ_updates", 0) + 1
        seed = int(getattr(self, "occ_seed", 0)) * 1000003 + self._occ_updates
        with torch.cuda.device(dev):
            call("ngp_occupancy_update", ptr(self.density_grid), ptr(self.density_bitfield), self.cascades, self.grid_size,
                 float(self.scale), float(density_threshold), float(decay), ptr(decay_grid), 1 if warmup else 0, seed,
                 ptr(self.xyz_min), ptr(self.xyz_max), ptr(eh[enc.n_mlp:]), C.byref(enc.meta), ptr(eh),
                 ptr(ws), nbytes, stream())

    @torch.no_grad()
    def update_density_grid(self, density_threshold, warmup=False, decay=0.95, erode=False):
        """Every 16 steps (train.py:160-163): sigma at jittered centres of the chosen cells,
        grid = max(grid*decay, sigma) where grid >= 0, threshold = min(mean(grid>0), thr), pack
        (networks.py:240-269).  The merge + mean + pack run as two kernels with the mean kept on
        device -- no .item() sync."""
        if getattr(self, "native_grid_update", True) and self.density_grid.is_cuda and self.fused:
            return self._update_density_grid_native(density_threshold, warmup, decay, erode)
        tmp = torch.zeros_like(self.density_grid)
        cells = self.get_all_cells() if warmup else \
            self.sample_uniform_and_occupied_cells(self.grid_size ** 3 // 4, density_threshold)
        for c in range(self.cascades):
            indices, coords = cells[c]
            s = min(2 ** (c - 1), self.scale)
            half_grid_size = s / self.grid_size
            xyzs_w = (coords / (self.grid_size - 1) * 2 - 1) * (s - half_grid_size)
            xyzs_w += (torch.rand_like(xyzs_w) * 2 - 1) * half_grid_size
            tmp[c, indices] = self.density(xyzs_w)
        decay_grid = None
        if erode:
            decay_grid = torch.clamp(decay ** (1 / self.count_grid), 0.1, 0.95).contiguous()
        stats = torch.zeros(2, dtype=torch.float32, device=tmp.device)
        with torch.cuda.device(tmp.device):
            call("ngp_density_grid_update", ptr(self.density_grid), ptr(tmp), ptr(decay_grid), float(decay),
                 self.density_grid.numel(), ptr(stats), stream())
            call("ngp_packbits_auto", ptr(self.density_grid), self.density_bitfield.numel(), ptr(stats),
                 float(density_threshold), ptr(self.density_bitfield), stream())
