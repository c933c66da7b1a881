"""`tinycudann` surface used by the reference (/root/reference/models/networks.py:36-92), backed by
the gfx950 kernels of libngp_hip.so:

    tcnn.NetworkWithInputEncoding(n_input_dims, n_output_dims, encoding_config, network_config)
    tcnn.Encoding(n_input_dims, encoding_config)
    tcnn.Network(n_input_dims, n_output_dims, network_config)

Same contract as tiny-cuda-nn's torch bindings: every module owns ONE float32 `params`
Parameter (state-dict key `<name>.params`, layout [MLP weights layer by layer (out,in) row-major,
then grid entries level-major]), casts it to f16 for the kernels, returns f16 of shape
(N, n_output_dims), and runs its backward with loss scale 128.  Supported configurations are the
ones the reference instantiates (hash grid F=2, L<=16, linear; FullyFusedMLP 64 neurons, ReLU,
1-2 hidden layers, <=16 outputs; SH degree 4); anything else raises -- there is no fallback.
"""
import ctypes as C
import math

import weakref

import torch
from torch import nn

from . import _lib
from ._lib import call, ptr, stream

LOSS_SCALE = 128.0          # tiny-cuda-nn's default for f16 params
_SEED = 1337                # tiny-cuda-nn's default seed


def _unsupported(what):
    raise NotImplementedError("ngp_pl_amd.tcnn: unsupported configuration: %s" % what)


def make_grid_meta(encoding_config, level_table="float32"):
    """Per-level scale / resolution / offset table of tiny-cuda-nn's GridEncodingTemplated constructor.

    level_table="float32" (default): evaluated in float32 exactly as grid.h does (`exp2f(l * log2f(b)) * N_min - 1`), by
    the library's ngp_grid_meta_init.  level_table="exact": the same formula in exact arithmetic (Python float64;
    `b**l * N_min - 1` lands on integers where float32 lands a few 1e-6 above them).  The two differ for the reference's
    b = exp(ln(2048 s / 16) / 15) at levels 5, 10 and 15 (resolution 65 / 257 / 1025 vs 64 / 256 / 1024; only level 5 is below the 2^19 cap), i.e. in the length of
    `xyz_encoder.params` (11 448 112 vs 11 423 136 at scale 0.5): a checkpoint only loads into the table it was trained
    with -- see ngp_pl_amd.utils.load_ckpt."""
    ec = encoding_config
    if ec.get("otype") not in ("Grid", "HashGrid") or ec.get("type", "Hash") != "Hash":
        _unsupported("encoding %r" % (ec,))
    if ec.get("interpolation", "Linear") != "Linear" or int(ec.get("n_features_per_level", 2)) != 2:
        _unsupported("grid needs F=2 and linear interpolation")
    n_levels, log2_T = int(ec.get("n_levels", 16)), int(ec.get("log2_hashmap_size", 19))
    n_min, b = int(ec.get("base_resolution", 16)), float(ec.get("per_level_scale", 2.0))
    meta = _lib.GridMeta()
    if level_table == "float32":
        call("ngp_grid_meta_init", C.byref(meta), n_levels, 2, log2_T, n_min, b)
    elif level_table == "exact":
        if not 1 <= n_levels <= _lib.NGP_MAX_LEVELS:
            _unsupported("n_levels %d" % n_levels)
        meta.n_levels, meta.n_features = n_levels, 2
        off = 0
        for l in range(n_levels):
            scale = round(2.0 ** (l * math.log2(b)) * n_min - 1.0, 9)      # 1e-9: absorbs the float64 noise around the integer hits
            res = int(math.ceil(scale)) + 1
            n = min((res ** 3 + 7) // 8 * 8, 1 << log2_T)
            meta.offset[l], meta.resolution[l], meta.scale[l] = off, res, scale
            off += n
        for l in range(n_levels, _lib.NGP_MAX_LEVELS + 1):
            meta.offset[l] = off
    else:
        raise ValueError("level_table must be 'float32' or 'exact', not %r" % (level_table,))
    return meta


def _check_network(nc, n_out):
    if nc.get("otype") not in ("FullyFusedMLP", "CutlassMLP") or nc.get("activation", "ReLU") != "ReLU":
        _unsupported("network %r" % (nc,))
    if int(nc.get("n_neurons", 64)) != 64 or int(nc.get("n_hidden_layers", 1)) not in (1, 2) or n_out > 16:
        _unsupported("network needs 64 neurons, 1-2 hidden layers, <=16 outputs")
    act = nc.get("output_activation", "None")
    if act not in ("None", "Sigmoid"):
        _unsupported("output activation %s" % act)
    return int(nc.get("n_hidden_layers", 1)), (1 if act == "Sigmoid" else 0)


def _xavier(gen, out_f, in_f):
    a = math.sqrt(6.0 / (in_f + out_f))
    return ((torch.rand(out_f, in_f, generator=gen) * 2 - 1) * a).reshape(-1)


def mlp_init(gen, n_in, n_hidden):
    dims = [n_in] + [64] * n_hidden + [16]
    return torch.cat([_xavier(gen, dims[i + 1], dims[i]) for i in range(len(dims) - 1)])


class _HalfCache:
    """f16 working copy of the f32 master params, refreshed when the Parameter changes
    (tiny-cuda-nn re-casts on every training forward)."""

    def __init__(self):
        self.t = None
        self.key = None

    def get(self, params):
        key = (params.data_ptr(), params._version)
        if self.t is None or self.key != key or self.t.device != params.device:
            if self.t is None or self.t.shape != params.shape or self.t.device != params.device:
                self.t = torch.empty(params.shape, dtype=torch.float16, device=params.device)
            with torch.cuda.device(params.device):
                call("ngp_cast_f32_to_f16", ptr(params.detach()), params.numel(), ptr(self.t), stream())
            self.key = key
        return self.t

    def invalidate(self):
        """The master params were rewritten behind autograd's back (`.data` writes do not bump the version)."""
        self.key = None

    def mark_fresh(self, params):
        """The fused optimizer already rewrote the f16 copy."""
        self.key = (params.data_ptr(), params._version)


_BIN_WS = {}


def binned_enabled():
    import os
    return os.environ.get("NGP_BINNED_BWD", "1") != "0"


def binned_workspace(device, nbytes):
    """Per-device scratch of the binned table backward (grown geometrically, reused by every call on the device's stream)."""
    ws = _BIN_WS.get(device)
    if ws is None or ws.numel() < nbytes:
        ws = _BIN_WS[device] = torch.empty(int(nbytes * 1.25), dtype=torch.uint8, device=device)
    return ws


def grid_backward(x, xyz_min, xyz_max, dfeats, meta, n, g16):
    """dL/dtable (packed f16, overwritten) for the full batch: the binned kernel (exact fixed-point sums,
    deterministic) when the batch fits it, else the one-pass sliced kernel.  NGP_BINNED_BWD=0 forces the latter."""
    lib = _lib.lib()
    nbytes = lib.ngp_hashgrid_bwd_binned_workspace_bytes(C.byref(meta), n) if binned_enabled() else 0
    if nbytes:
        ws = binned_workspace(x.device, nbytes)
        call("ngp_hashgrid_bwd_binned", ptr(x), ptr(xyz_min), ptr(xyz_max), ptr(dfeats), C.byref(meta), n, None, None,
             ptr(ws), ws.numel(), ptr(g16), stream())
    else:
        call("ngp_hashgrid_bwd_sliced", ptr(x), ptr(xyz_min), ptr(xyz_max), ptr(dfeats), C.byref(meta), n, None, None, ptr(g16), stream())


def reduce_partials(partials, n_partials, n):
    out = torch.empty(n, dtype=torch.float32, device=partials.device)
    call("ngp_reduce_partials", ptr(partials), n_partials, n, ptr(out), stream())
    return out


# ---------------------------------------------------------------------------------------------
class _EncodeAndNet(torch.autograd.Function):
    """hash grid -> density-type MLP (32 -> 64 -> 16)."""

    @staticmethod
    def forward(ctx, x, params, mod):
        x = x.detach().float().contiguous()
        _lib.require_cuda(x)
        n = x.shape[0]
        ph = mod._half.get(params)
        dev = x.device
        feats = torch.empty(mod.n_levels, n, 2, dtype=torch.float16, device=dev)
        h = torch.empty(n, 16, dtype=torch.float16, device=dev)
        sig = torch.empty(n, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            call("ngp_hashgrid_fwd", ptr(x), ptr(mod._zero3), ptr(mod._one3), ptr(ph[mod.n_mlp:]), C.byref(mod.meta), n,
                 ptr(feats), stream())
            call("ngp_density_fwd", ptr(feats), ptr(ph), n, ptr(sig), ptr(h), stream())
        ctx.mod = mod
        ctx.save_for_backward(x, feats)
        return h if mod.n_output_dims == 16 else h[:, :mod.n_output_dims]

    @staticmethod
    def backward(ctx, dL_dh):
        mod = ctx.mod
        x, feats = ctx.saved_tensors
        n = x.shape[0]
        dev = x.device
        grad = torch.zeros(mod.params.numel(), dtype=torch.float32, device=dev)
        want_dx = ctx.needs_input_grad[0]
        if n == 0:
            return (torch.zeros(0, 3, dtype=torch.float32, device=dev) if want_dx else None), grad, None
        dh = torch.zeros(n, 16, dtype=torch.float16, device=dev)
        dh[:, :mod.n_output_dims] = (dL_dh.float() * LOSS_SCALE).half()
        ph = mod._half.get(mod.params)
        with torch.cuda.device(dev):
            n_part = call("ngp_field_bwd_partials", n)
            partials = torch.empty(n_part, mod.n_mlp, dtype=torch.float32, device=dev)
            dfeats = torch.empty(mod.n_levels, n, 2, dtype=torch.float16, device=dev)
            call("ngp_density_bwd", ptr(feats), ptr(ph), ptr(dh), None, 1.0, n, None, None, ptr(dfeats), ptr(partials), stream())
            grad[:mod.n_mlp] = reduce_partials(partials, n_part, mod.n_mlp) / LOSS_SCALE
            g16 = torch.empty(mod.n_grid, dtype=torch.float16, device=dev)
            grid_backward(x, mod._zero3, mod._one3, dfeats, mod.meta, n, g16)
            call("ngp_cast_f16_to_f32", ptr(g16), mod.n_grid, 1.0 / LOSS_SCALE, ptr(grad[mod.n_mlp:]), stream())
            dx = None
            if want_dx:          # pose optimisation: the sample positions carry gradients (train.py:86-89)
                dx = torch.empty(n, 3, dtype=torch.float32, device=dev)
                call("ngp_hashgrid_bwd_input", ptr(x), ptr(mod._zero3), ptr(mod._one3), ptr(ph[mod.n_mlp:]), ptr(dfeats),
                     C.byref(mod.meta), n, 1.0 / LOSS_SCALE, ptr(dx), stream())
        return dx, grad, None


class NetworkWithInputEncoding(nn.Module):
    def __init__(self, n_input_dims, n_output_dims, encoding_config, network_config, seed=_SEED, level_table="float32"):
        super().__init__()
        if n_input_dims != 3:
            _unsupported("grid encoding needs 3 input dims")
        self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims
        self.level_table = level_table
        self.encoding_config = dict(encoding_config)
        self.meta = make_grid_meta(encoding_config, level_table)
        self.n_levels = int(self.meta.n_levels)
        n_hidden, act = _check_network(network_config, n_output_dims)
        if self.n_levels != 16 or n_hidden != 1 or act != 0:
            _unsupported("fused encoder+MLP is built for L=16 and a 32->64->16 linear-output network")
        self.n_mlp = 64 * 32 + 16 * 64
        self.n_grid = int(self.meta.offset[self.n_levels]) * 2
        gen = torch.Generator().manual_seed(seed)
        mlp = mlp_init(gen, 32, 1)
        grid = (torch.rand(self.n_grid, generator=gen) * 2 - 1) * 1e-4
        self.params = nn.Parameter(torch.cat([mlp, grid]))
        self.params._tcnn_module = weakref.ref(self)            # optim.FusedAdam refreshes this module's f16 working copy
        self.register_buffer("_zero3", torch.zeros(3), persistent=False)
        self.register_buffer("_one3", torch.ones(3), persistent=False)
        self._half = _HalfCache()
        self.loss_scale = LOSS_SCALE

    def forward(self, x):
        return _EncodeAndNet.apply(x, self.params, self)


# ---------------------------------------------------------------------------------------------
class Encoding(nn.Module):
    """SphericalHarmonics degree 4 (networks.py:58-65).  No parameters; the empty `params`
    keeps the reference's state-dict key `dir_encoder.params`."""

    def __init__(self, n_input_dims, encoding_config, seed=_SEED):
        super().__init__()
        if encoding_config.get("otype") != "SphericalHarmonics" or int(encoding_config.get("degree", 4)) != 4 or n_input_dims != 3:
            _unsupported("encoding %r" % (encoding_config,))
        self.n_input_dims, self.n_output_dims = 3, 16
        # (no gradient ever reaches an empty tensor: requires_grad=True would make DistributedDataParallel wait for one and fail
        #  in the second iteration -- "Expected to have finished reduction in the prior iteration")
        self.params = nn.Parameter(torch.zeros(0), requires_grad=False)

    def forward(self, x):
        return _SH4.apply(x)


class _SH4(torch.autograd.Function):
    """x in [0,1]^3 = (d+1)/2 -> 16 SH values f16; backward w.r.t. x for pose optimisation."""

    @staticmethod
    def forward(ctx, x):
        xin = x.detach().float().contiguous()
        _lib.require_cuda(xin)
        out = torch.empty(xin.shape[0], 16, dtype=torch.float16, device=xin.device)
        with torch.cuda.device(xin.device):
            call("ngp_sh4_fwd", ptr(xin), xin.shape[0], ptr(out), stream())
        ctx.save_for_backward(xin)
        ctx.in_dtype = x.dtype
        return out

    @staticmethod
    def backward(ctx, dL_dout):
        (xin,) = ctx.saved_tensors
        n, dev = xin.shape[0], xin.device
        dx = torch.empty(n, 3, dtype=torch.float32, device=dev)
        if n:
            g = (dL_dout.float() * LOSS_SCALE).half().contiguous()       # f16 transport with the modules' loss scale
            with torch.cuda.device(dev):
                call("ngp_sh4_bwd", ptr(xin), ptr(g), n, 1.0 / LOSS_SCALE, ptr(dx), stream())
        return dx.to(ctx.in_dtype)


# ---------------------------------------------------------------------------------------------
class _Net(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, mod):
        _lib.require_cuda(x)
        n = x.shape[0]
        xin = x.detach()
        if xin.shape[1] != mod.n_in_padded or xin.dtype != torch.float16 or not xin.is_contiguous():
            pad = torch.zeros(n, mod.n_in_padded, dtype=torch.float16, device=x.device)
            pad[:, :xin.shape[1]] = xin
            xin = pad
        ph = mod._half.get(params)
        out = torch.empty(n, mod.n_output_dims, dtype=torch.float16, device=x.device)
        with torch.cuda.device(x.device):
            call("ngp_mlp_fwd", ptr(xin), ptr(ph), mod.n_in_padded, mod.n_hidden, mod.n_output_dims, mod.out_act, n, ptr(out), stream())
        ctx.mod = mod
        ctx.in_dtype = x.dtype
        ctx.save_for_backward(xin)
        return out

    @staticmethod
    def backward(ctx, dL_dout):
        mod = ctx.mod
        (xin,) = ctx.saved_tensors
        n = xin.shape[0]
        dev = xin.device
        grad = torch.zeros(mod.params.numel(), dtype=torch.float32, device=dev)
        if n == 0:
            return torch.zeros(0, mod.n_input_dims, dtype=ctx.in_dtype, device=dev), grad, None
        dout = (dL_dout.float() * LOSS_SCALE).half().contiguous()
        ph = mod._half.get(mod.params)
        with torch.cuda.device(dev):
            n_part = call("ngp_mlp_bwd_partials", n)
            partials = torch.empty(n_part, mod.params.numel(), dtype=torch.float32, device=dev)
            din = torch.empty(n, mod.n_in_padded, dtype=torch.float16, device=dev)
            call("ngp_mlp_bwd", ptr(xin), ptr(ph), ptr(dout), mod.n_in_padded, mod.n_hidden, mod.n_output_dims, mod.out_act, n,
                 ptr(din), ptr(partials), stream())
            grad = reduce_partials(partials, n_part, mod.params.numel()) / LOSS_SCALE
        dx = (din[:, :mod.n_input_dims].float() / LOSS_SCALE).to(ctx.in_dtype)
        return dx, grad, None


class Network(nn.Module):
    def __init__(self, n_input_dims, n_output_dims, network_config, seed=_SEED):
        super().__init__()
        self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims
        self.n_in_padded = (n_input_dims + 15) // 16 * 16      # tiny-cuda-nn pads the input width to 16
        if self.n_in_padded not in (16, 32, 64):
            _unsupported("network input width %d" % n_input_dims)
        self.n_hidden, self.out_act = _check_network(network_config, n_output_dims)
        gen = torch.Generator().manual_seed(seed)
        self.params = nn.Parameter(mlp_init(gen, self.n_in_padded, self.n_hidden))
        self.params._tcnn_module = weakref.ref(self)
        self._half = _HalfCache()
        self.loss_scale = LOSS_SCALE

    def forward(self, x):
        return _Net.apply(x, self.params, self)
