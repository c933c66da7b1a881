"""TEST INFRASTRUCTURE -- the reference's OWN `train.py`, unmodified, executed end to end on this package's bindings.

north_star: "... keeps the reference's models.rendering.render() / custom_functions autograd.Function API and NGP module surface so
train.py / pytorch-lightning drives it unchanged".  `train.py` cannot be imported in this image as it stands: pytorch_lightning, apex,
torchmetrics, kornia, cv2, imageio and tiny-cuda-nn are not installed and there is no network.  This module supplies what is missing
and NOTHING of the hot path:

    vren, tinycudann              -> ngp_pl_amd.vren / ngp_pl_amd.tcnn            (the product: every kernel that runs)
    apex.optimizers.FusedAdam     -> ngp_pl_amd.optim.FusedAdam                   (the product's drop-in)
    pytorch_lightning             -> ~150 lines below: LightningModule (hparams, log, device, global_step) and a Trainer.fit that does
                                     what Lightning's does for THIS script: setup, configure_optimizers, on_train_start, the epoch loop
                                     over train_dataloader() with native AMP (precision=16: autocast + GradScaler, train.py:274), the
                                     scheduler per epoch, ModelCheckpoint at the end, the validation loop
    torchmetrics                  -> PSNR / SSIM written out; kornia -> the two meshgrid helpers; cv2, imageio -> import-only stubs
                                     (the run passes --no_save_test)
    datasets                      -> `dataset_dict` with ONE procedural dataset (no data exists on any box) that subclasses the
                                     reference's own datasets/base.py:BaseDataset -- ITS __getitem__ draws the batches -- and builds its rays
                                     with the reference's own datasets/ray_utils.py

and runs `train.py` as __main__ (runpy) with a command line.  The reference's files are read from /root/reference where mounted, else
from the byte-for-byte copies oracle/build_ref.sh stages under the git-ignored oracle/_ref/py/.  Only tests/ may import this module.
"""
import contextlib
import importlib.util
import os
import runpy
import sys
import types

import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
NEEDED = ("train.py", "opt.py", "utils.py", "losses.py", os.path.join("models", "networks.py"), os.path.join("datasets", "base.py"),
          os.path.join("datasets", "ray_utils.py"))


def source_dir():
    for d in (os.environ.get("NGP_REFERENCE_DIR", "/root/reference"), os.path.join(HERE, "_ref", "py")):
        if all(os.path.isfile(os.path.join(d, f)) for f in NEEDED):
            return d
    return None


def available():
    return source_dir() is not None


# ---- pytorch_lightning ---------------------------------------------------------------------------------------------------------
class LightningModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.global_step = 0
        self.current_epoch = 0
        self.logged = {}

    def save_hyperparameters(self, hparams):
        self.hparams = hparams

    @property
    def device(self):
        for p in self.parameters():
            return p.device
        return torch.device("cpu")

    def log(self, name, value, prog_bar=False, **kw):
        if hasattr(value, "compute"):                 # a torchmetrics object (train.py:183)
            value = value.compute()
        self.logged[name] = float(value)


class ModelCheckpoint:
    def __init__(self, dirpath, filename="{epoch:d}", **kw):
        self.dirpath, self.filename = dirpath, filename

    def save(self, system, epoch):
        os.makedirs(self.dirpath, exist_ok=True)
        name = self.filename.replace("{epoch:d}", "epoch=%d" % epoch)
        torch.save({"state_dict": system.state_dict(), "epoch": epoch, "global_step": system.global_step}, os.path.join(self.dirpath, name + ".ckpt"))


class Trainer:
    """What `Trainer(max_epochs, check_val_every_n_epoch, callbacks, logger, accelerator='gpu', devices=1, precision=16).fit(system)`
    does for train.py (pytorch-lightning 1.6-1.7 semantics): one device, native AMP, optimizer 0 only (train.py's second optimizer
    exists only with --optimize_ext), LR schedulers stepped per epoch, validation every `check_val_every_n_epoch` epochs."""

    def __init__(self, max_epochs=1, check_val_every_n_epoch=1, callbacks=(), logger=None, precision=32, devices=1, max_steps_per_epoch=None, **kw):
        assert devices in (1, None) and precision == 16
        self.max_epochs, self.val_every, self.callbacks = max_epochs, check_val_every_n_epoch, list(callbacks)
        self.max_steps_per_epoch = max_steps_per_epoch if max_steps_per_epoch is not None else int(os.environ.get("NGP_HARNESS_STEPS_PER_EPOCH", "0")) or None

    @staticmethod
    def _to(batch, dev):
        return {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else (torch.as_tensor(v).to(dev) if hasattr(v, "shape") else v)) for k, v in batch.items()}

    def fit(self, system, ckpt_path=None):
        if os.environ.get("NGP_HARNESS_DRY") == "1":               # CPU suite: everything of the script but the steps themselves
            system.setup("fit")
            for cb in self.callbacks:
                if isinstance(cb, ModelCheckpoint):
                    cb.save(system, self.max_epochs - 1)
            return
        dev = torch.device("cuda", torch.cuda.current_device())
        system.setup("fit")
        system.to(dev)
        opts, schs = system.configure_optimizers()
        system.to(dev)                                            # (buffers / parameters registered by configure_optimizers)
        opt = opts[0]
        scaler = torch.amp.GradScaler("cuda")
        system.on_train_start()
        loader = system.train_dataloader()
        system.train()
        for epoch in range(self.max_epochs):
            system.current_epoch = epoch
            for i, batch in enumerate(loader):
                if self.max_steps_per_epoch is not None and i >= self.max_steps_per_epoch:
                    break
                batch = self._to(batch, dev)
                with torch.autocast("cuda", dtype=torch.float16):
                    loss = system.training_step(batch, i)
                opt.zero_grad()
                scaler.scale(loss).backward()
                scaler.step(opt)
                scaler.update()
                system.global_step += 1
            for s in schs:
                s.step()
            if (epoch + 1) % self.val_every == 0:
                self.validate(system, dev)
        for cb in self.callbacks:
            if isinstance(cb, ModelCheckpoint):
                cb.save(system, self.max_epochs - 1)
        del loader

    @torch.no_grad()
    def validate(self, system, dev):
        system.eval()
        system.on_validation_start()
        outs = []
        for i, batch in enumerate(system.val_dataloader()):
            with torch.autocast("cuda", dtype=torch.float16):
                outs.append(system.validation_step(self._to(batch, dev), i))
        system.validation_epoch_end(outs)
        system.train()


def _pytorch_lightning():
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule, pl.Trainer = LightningModule, Trainer
    plugins = types.ModuleType("pytorch_lightning.plugins")
    plugins.DDPPlugin = type("DDPPlugin", (), {"__init__": lambda self, **kw: None})
    cbs = types.ModuleType("pytorch_lightning.callbacks")
    cbs.ModelCheckpoint = ModelCheckpoint
    cbs.TQDMProgressBar = type("TQDMProgressBar", (), {"__init__": lambda self, **kw: None})
    loggers = types.ModuleType("pytorch_lightning.loggers")
    loggers.TensorBoardLogger = type("TensorBoardLogger", (), {"__init__": lambda self, **kw: None})
    util = types.ModuleType("pytorch_lightning.utilities")
    dist = types.ModuleType("pytorch_lightning.utilities.distributed")
    dist.all_gather_ddp_if_available = lambda t, *a, **k: t
    util.distributed = dist
    pl.plugins, pl.callbacks, pl.loggers, pl.utilities = plugins, cbs, loggers, util
    return {"pytorch_lightning": pl, "pytorch_lightning.plugins": plugins, "pytorch_lightning.callbacks": cbs,
            "pytorch_lightning.loggers": loggers, "pytorch_lightning.utilities": util, "pytorch_lightning.utilities.distributed": dist}


# ---- torchmetrics --------------------------------------------------------------------------------------------------------------
class PeakSignalNoiseRatio(nn.Module):
    def __init__(self, data_range=1.0):
        super().__init__()
        self.data_range = float(data_range)
        self.reset()

    def reset(self):
        self.se, self.n = 0.0, 0

    def forward(self, pred, target):
        self.se += float(((pred.float() - target.float()) ** 2).sum()); self.n += target.numel()

    def compute(self):
        import math
        return torch.tensor(10.0 * math.log10(self.data_range ** 2 / max(self.se / max(self.n, 1), 1e-12)))


class StructuralSimilarityIndexMeasure(nn.Module):
    """SSIM with an 11x11 uniform window (torchmetrics' default is gaussian: the value is logged, nothing is asserted on it)."""

    def __init__(self, data_range=1.0):
        super().__init__()
        self.c1, self.c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
        self.reset()

    def reset(self):
        self.v, self.n = 0.0, 0

    def forward(self, pred, target):
        import torch.nn.functional as F
        p, t = pred.float(), target.float()
        mu_p, mu_t = F.avg_pool2d(p, 11, 1), F.avg_pool2d(t, 11, 1)
        s_pp = F.avg_pool2d(p * p, 11, 1) - mu_p ** 2; s_tt = F.avg_pool2d(t * t, 11, 1) - mu_t ** 2; s_pt = F.avg_pool2d(p * t, 11, 1) - mu_p * mu_t
        ssim = ((2 * mu_p * mu_t + self.c1) * (2 * s_pt + self.c2)) / ((mu_p ** 2 + mu_t ** 2 + self.c1) * (s_pp + s_tt + self.c2))
        self.v += float(ssim.mean()); self.n += 1

    def compute(self):
        return torch.tensor(self.v / max(self.n, 1))


def _torchmetrics():
    tm = types.ModuleType("torchmetrics")
    tm.PeakSignalNoiseRatio, tm.StructuralSimilarityIndexMeasure = PeakSignalNoiseRatio, StructuralSimilarityIndexMeasure
    img = types.ModuleType("torchmetrics.image"); lp = types.ModuleType("torchmetrics.image.lpip")

    class LearnedPerceptualImagePatchSimilarity(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError("LPIPS needs pretrained VGG weights: not available offline (--eval_lpips is not part of the hot path)")
    lp.LearnedPerceptualImagePatchSimilarity = LearnedPerceptualImagePatchSimilarity
    img.lpip = lp; tm.image = img
    return {"torchmetrics": tm, "torchmetrics.image": img, "torchmetrics.image.lpip": lp}


# ---- kornia / cv2 / imageio / apex ---------------------------------------------------------------------------------------------
def create_meshgrid(height, width, normalized_coordinates=True, device=None, dtype=torch.float32):
    """kornia.create_meshgrid for normalized_coordinates=False: (1, H, W, 2) with (x, y) last (datasets/ray_utils.py:26)."""
    assert not normalized_coordinates
    ys, xs = torch.meshgrid(torch.arange(height, dtype=dtype, device=device), torch.arange(width, dtype=dtype, device=device), indexing="ij")
    return torch.stack([xs, ys], -1).unsqueeze(0)


def _small_standins():
    from oracle.ref_on_binding import _torch_scatter, create_meshgrid3d
    import ngp_pl_amd.optim
    import ngp_pl_amd.tcnn
    import ngp_pl_amd.vren
    ko, ku, kg = types.ModuleType("kornia"), types.ModuleType("kornia.utils"), types.ModuleType("kornia.utils.grid")
    kg.create_meshgrid3d = create_meshgrid3d; ku.grid = kg; ko.utils = ku
    ko.create_meshgrid, ko.create_meshgrid3d = create_meshgrid, create_meshgrid3d
    cv2 = types.ModuleType("cv2"); cv2.COLORMAP_TURBO = 20
    cv2.applyColorMap = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("cv2 stand-in: run with --no_save_test"))
    imageio = types.ModuleType("imageio")
    apex, apex_opt = types.ModuleType("apex"), types.ModuleType("apex.optimizers")
    apex_opt.FusedAdam = ngp_pl_amd.optim.FusedAdam; apex.optimizers = apex_opt
    return {"vren": ngp_pl_amd.vren, "tinycudann": ngp_pl_amd.tcnn, "torch_scatter": _torch_scatter(), "kornia": ko, "kornia.utils": ku,
            "kornia.utils.grid": kg, "cv2": cv2, "imageio": imageio, "apex": apex, "apex.optimizers": apex_opt}


# ---- datasets ------------------------------------------------------------------------------------------------------------------
def _load_file(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _datasets(src, res, n_train, n_test):
    """A `datasets` package whose ray_utils / base ARE the reference's files and whose dataset_dict holds one procedural dataset."""
    pkg = types.ModuleType("datasets"); pkg.__path__ = []
    sys.modules["datasets"] = pkg
    ray_utils = _load_file("datasets.ray_utils", os.path.join(src, "datasets", "ray_utils.py"))
    base = _load_file("datasets.base", os.path.join(src, "datasets", "base.py"))
    pkg.ray_utils, pkg.base = ray_utils, base
    from ngp_pl_amd import synthetic as syn
    from ngp_pl_amd.bench_support import surface_ground_truth

    class ProceduralDataset(base.BaseDataset):
        """The procedural Lego-like scene of bench.py as a `BaseDataset` (datasets/nsvf.py's shape: K, img_wh, directions, poses,
        rays (N, H*W, 3)); batches are drawn by the reference's own BaseDataset.__getitem__."""

        def __init__(self, root_dir, split="train", downsample=1.0, **kw):
            super().__init__(root_dir, split, downsample)
            w = h = int(res * downsample)
            self.K = syn.intrinsics(w)
            self.img_wh = (w, h)
            self.directions = ray_utils.get_ray_directions(h, w, self.K)          # the reference's own (datasets/ray_utils.py:8-47)
            n, seed = (n_train, 0) if split.startswith("train") else (n_test, 999)
            self.poses = syn.hemisphere_poses(n, seed=seed)
            rays = []
            for i in range(n):
                ro, rd = ray_utils.get_rays(self.directions, self.poses[i])
                rays.append(surface_ground_truth(ro.contiguous(), rd.contiguous()))
            self.rays = torch.stack(rays)
    pkg.dataset_dict = {"nsvf": ProceduralDataset, "nerf": ProceduralDataset}
    return ["datasets", "datasets.ray_utils", "datasets.base"]


@contextlib.contextmanager
def _environment(src, res, n_train, n_test, workdir):
    names = {}
    names.update(_pytorch_lightning()); names.update(_torchmetrics()); names.update(_small_standins())
    touched = list(names) + ["datasets", "datasets.ray_utils", "datasets.base", "models", "models.networks", "models.rendering",
                            "models.custom_functions", "losses", "opt", "utils", "metrics"]
    saved = {k: sys.modules.get(k) for k in touched}
    for k in touched:
        sys.modules.pop(k, None)
    sys.modules.update(names)
    cwd, argv = os.getcwd(), list(sys.argv)
    sys.path.insert(0, src)
    try:
        _datasets(src, res, n_train, n_test)
        os.chdir(workdir)
        yield
    finally:
        os.chdir(cwd)
        sys.argv = argv
        sys.path.remove(src)
        for k in [k for k in sys.modules if k in touched or k.startswith("models.") or k.startswith("datasets.") or k.startswith("pytorch_lightning") or k.startswith("torchmetrics")]:
            del sys.modules[k]
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v


def run_train(argv, workdir, res=200, n_train=24, n_test=4):
    """`python train.py <argv>` with the stand-ins in place, in `workdir` (ckpts/ and logs/ land there).  Returns the module's globals:
    `system` (the NeRFSystem, with `.logged`), `hparams`, ..."""
    src = source_dir()
    if src is None:
        raise RuntimeError("the reference's train.py / opt.py / utils.py / datasets/{base,ray_utils}.py are neither at /root/reference nor "
                           "staged under oracle/_ref/py (run oracle/build_ref.sh where /root/reference is mounted)")
    import warnings
    with _environment(src, res, n_train, n_test, workdir):
        sys.argv = ["train.py"] + list(argv)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return runpy.run_path(os.path.join(src, "train.py"), run_name="__main__")
