"""TEST / MEASUREMENT INFRASTRUCTURE -- the reference's OWN Python hot path executed on this package's bindings.

INTEGRATION.md "Option A" run for real: the reference's unmodified `models/rendering.py`, `models/networks.py`,
`models/custom_functions.py` and `losses.py` are imported with nothing but two module aliases in front of them

    vren        -> ngp_pl_amd.vren   (models/csrc/binding.cpp:234-250: the 12 native functions)
    tinycudann  -> ngp_pl_amd.tcnn   (models/networks.py:36-92: NetworkWithInputEncoding / Encoding / Network)

and two small stand-ins for packages the hot path does not need from a GPU library (`torch_scatter.segment_csr`, which
only RayMarcher.backward uses, custom_functions.py:108-110; `kornia.create_meshgrid3d`, which train.py:75 uses once to
enumerate the occupancy cells).  Where the files come from: /root/reference when it is mounted (this container), else the
copies oracle/build_ref.sh staged, byte for byte, under oracle/_ref/py/ (git-ignored, shipped to the GPU box with the
snapshot).  Nothing here computes anything: every kernel that runs is libngp_hip.so's.

Who may import this: tests/ and bench.py's `api_path_reference_files` leg (which times the reference's files over the
product's kernels -- the product is what is measured, this module only loads the caller).  The product itself never does.
"""
import importlib
import os
import sys
import types
import warnings

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
_NAMES = ("vren", "tinycudann", "torch_scatter", "kornia", "kornia.utils", "kornia.utils.grid")
_CACHE = {}


def source_dir():
    """Directory holding the reference's `models/` package and `losses.py`, or None."""
    for d in (os.environ.get("NGP_REFERENCE_DIR", "/root/reference"), os.path.join(HERE, "_ref", "py")):
        if os.path.isfile(os.path.join(d, "models", "rendering.py")) and os.path.isfile(os.path.join(d, "losses.py")):
            return d
    return None


def available():
    return source_dir() is not None


def _torch_scatter():
    from ngp_pl_amd.custom_functions import segment_sum
    m = types.ModuleType("torch_scatter")

    def segment_csr(src, indptr, out=None, reduce="sum"):
        n = indptr.shape[0] - 1
        rays_a = torch.stack([torch.arange(n, device=indptr.device), indptr[:-1], indptr[1:] - indptr[:-1]], 1)
        return segment_sum(src, rays_a)
    m.segment_csr = segment_csr
    return m


def create_meshgrid3d(depth, height, width, normalized_coordinates=True, device=None, dtype=torch.float32):
    """kornia.utils.grid.create_meshgrid3d for normalized_coordinates=False: (1, D, H, W, 3) with xyz last (train.py:75)."""
    assert not normalized_coordinates
    gz, gy, gx = torch.meshgrid(torch.arange(depth, dtype=dtype, device=device), torch.arange(height, dtype=dtype, device=device),
                                torch.arange(width, dtype=dtype, device=device), indexing="ij")
    return torch.stack([gx, gy, gz], -1).unsqueeze(0)


def _kornia():
    ko, ku, kg = types.ModuleType("kornia"), types.ModuleType("kornia.utils"), types.ModuleType("kornia.utils.grid")
    kg.create_meshgrid3d = create_meshgrid3d
    ku.grid = kg
    ko.utils = ku
    ko.create_meshgrid3d = create_meshgrid3d
    return ko, ku, kg


def load():
    """-> namespace(rendering, networks, custom_functions, losses, source): the reference's modules, imported once over the
    product's bindings.  The aliases are removed from sys.modules again (the loaded modules keep their own references)."""
    if "mods" in _CACHE:
        return _CACHE["mods"]
    src = source_dir()
    if src is None:
        raise RuntimeError("the reference's models/*.py are neither at /root/reference nor staged under oracle/_ref/py "
                           "(run oracle/build_ref.sh where /root/reference is mounted)")
    import ngp_pl_amd.tcnn
    import ngp_pl_amd.vren
    ko, ku, kg = _kornia()
    standins = {"vren": ngp_pl_amd.vren, "tinycudann": ngp_pl_amd.tcnn, "torch_scatter": _torch_scatter(), "kornia": ko,
                "kornia.utils": ku, "kornia.utils.grid": kg}
    saved = {k: sys.modules.get(k) for k in _NAMES}
    stale = [k for k in sys.modules if k == "models" or k.startswith("models.") or k == "losses"]
    saved_pkgs = {k: sys.modules.pop(k) for k in stale}
    sys.modules.update(standins)
    sys.path.insert(0, src)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                 # torch.cuda.amp.custom_fwd / autocast deprecation notes of the 2022 sources
            mods = types.SimpleNamespace(rendering=importlib.import_module("models.rendering"),
                                         networks=importlib.import_module("models.networks"),
                                         custom_functions=importlib.import_module("models.custom_functions"),
                                         losses=importlib.import_module("losses"), source=src)
    finally:
        sys.path.remove(src)
        for k in [k for k in sys.modules if k == "models" or k.startswith("models.") or k == "losses"]:
            del sys.modules[k]
        sys.modules.update(saved_pkgs)
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
            else:
                sys.modules.pop(k, None)
    _CACHE["mods"] = mods
    return mods


def make_model(scale, device, rgb_act="Sigmoid", seed=None):
    """The reference's NGP the way NeRFSystem.__init__ builds it (train.py:71-76): the class from models/networks.py plus the two
    buffers train.py registers on it."""
    mods = load()
    if seed is not None:
        torch.manual_seed(seed)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):          # (the constructor prints its grid configuration)
        model = mods.networks.NGP(scale=scale, rgb_act=rgb_act)
    G = model.grid_size
    model.register_buffer("density_grid", torch.zeros(model.cascades, G ** 3))
    model.register_buffer("grid_coords", create_meshgrid3d(G, G, G, False, dtype=torch.int32).reshape(-1, 3))
    return model.to(device)


class TrainingStep:
    """NeRFSystem.training_step (train.py:159-185) + configure_optimizers (train.py:123-137) without Lightning, statement for
    statement, around the reference's own render / NGP / NeRFLoss: occupancy update every 16 steps, render, loss, backward,
    FusedAdam(net_params, lr, eps=1e-15).  `optimizer_cls` is apex.optimizers.FusedAdam in the reference; the caller passes this
    package's drop-in.  `amp=True` (default) reproduces what `Trainer(precision=16)` (train.py:274) wraps around the step: the
    forward -- occupancy update included, which only works BECAUSE of it: NGP.density() returns float32 through TruncExp's
    custom_fwd(cast_inputs=float32) under autocast and float16 (tiny-cuda-nn's output dtype) without, and
    `density_grid_tmp[c, indices] = self.density(xyzs_w)` (networks.py:256) needs float32 -- under torch.autocast, the loss
    scaled by a GradScaler, unscale + skip-on-inf around the optimizer step."""

    def __init__(self, model, optimizer_cls, lr=1e-2, distortion_loss_w=0.0, erode=False, amp=True):
        mods = load()
        self.mods, self.model = mods, model
        self.loss = mods.losses.NeRFLoss(lambda_distortion=distortion_loss_w)
        self.net_opt = optimizer_cls([p for p in model.parameters()], lr, eps=1e-15)
        self.warmup_steps, self.update_interval, self.global_step = 256, 16, 0
        self.erode = erode
        self.kwargs = {"test_time": False, "random_bg": False}
        if model.scale > 0.5:                               # train.py:95-96
            self.kwargs["exp_step_factor"] = 1 / 256
        self.amp = amp
        self.scaler = torch.amp.GradScaler("cuda") if amp else None

    def __call__(self, rays_o, rays_d, rgb):
        MAX_SAMPLES = self.mods.rendering.MAX_SAMPLES
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.amp):
            if self.global_step % self.update_interval == 0:
                self.model.update_density_grid(0.01 * MAX_SAMPLES / 3 ** 0.5, warmup=self.global_step < self.warmup_steps, erode=self.erode)
            results = self.mods.rendering.render(self.model, rays_o, rays_d, **self.kwargs)
            loss_d = self.loss(results, {"rgb": rgb})
            loss = sum(lo.mean() for lo in loss_d.values())
        self.net_opt.zero_grad()
        if self.scaler is not None:
            self.scaler.scale(loss).backward()
            self.scaler.step(self.net_opt)
            self.scaler.update()
        else:
            loss.backward()
            self.net_opt.step()
        self.global_step += 1
        return results, loss
