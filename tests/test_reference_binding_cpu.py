"""CPU, only where /root/reference is mounted: INTEGRATION.md's "Option A" as a test -- the reference's OWN modules
(models/rendering.py, models/networks.py, models/custom_functions.py, losses.py) import and construct on top of THIS
package's bindings after nothing but two module aliases (`vren` -> ngp_pl_amd.vren, `tinycudann` -> ngp_pl_amd.tcnn;
torch_scatter / kornia, which the hot path does not need from a GPU, get two-line stand-ins).  What it pins: every name,
constructor argument and attribute the reference uses from the two native dependencies exists in the binding with a
compatible signature, and a model built by the reference's own NGP class has exactly the state dict of ngp_pl_amd's.
The numerics of the binding are the -m gpu tests' business (no GPU work happens here)."""
import importlib
import inspect
import os
import sys
import types

import pytest
import torch

REFERENCE = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "models")), reason="needs /root/reference")


@pytest.fixture(scope="module")
def ref():
    import ngp_pl_amd.tcnn
    import ngp_pl_amd.vren
    saved = {k: sys.modules.get(k) for k in ("vren", "tinycudann", "torch_scatter", "kornia", "models", "losses")}
    sys.modules["vren"] = ngp_pl_amd.vren
    sys.modules["tinycudann"] = ngp_pl_amd.tcnn
    ts = types.ModuleType("torch_scatter")
    from ngp_pl_amd.custom_functions import segment_sum
    ts.segment_csr = lambda src, indptr, out=None, reduce="sum": segment_sum(src, torch.stack([torch.arange(len(indptr) - 1), indptr[:-1], indptr[1:] - indptr[:-1]], 1))
    sys.modules["torch_scatter"] = ts
    ko = types.ModuleType("kornia"); ku = types.ModuleType("kornia.utils"); kg = types.ModuleType("kornia.utils.grid")
    def create_meshgrid3d(d, h, w, normalized_coordinates=True, device="cpu", dtype=torch.float32):
        zz, yy, xx = torch.meshgrid(torch.arange(d, dtype=dtype), torch.arange(h, dtype=dtype), torch.arange(w, dtype=dtype), indexing="ij")
        return torch.stack([xx, yy, zz], -1)[None]
    kg.create_meshgrid3d = create_meshgrid3d
    ku.grid = kg; ko.utils = ku
    ko.create_meshgrid = lambda h, w, normalized_coordinates=True, device="cpu": torch.stack(torch.meshgrid(torch.arange(w), torch.arange(h), indexing="xy"), -1)[None].float()
    sys.modules["kornia"], sys.modules["kornia.utils"], sys.modules["kornia.utils.grid"] = ko, ku, kg
    sys.path.insert(0, REFERENCE)
    for k in list(sys.modules):
        if k == "models" or k.startswith("models.") or k == "losses":
            del sys.modules[k]
    mods = types.SimpleNamespace(rendering=importlib.import_module("models.rendering"), networks=importlib.import_module("models.networks"),
                                 custom_functions=importlib.import_module("models.custom_functions"), losses=importlib.import_module("losses"))
    yield mods
    sys.path.remove(REFERENCE)
    for k in list(sys.modules):
        if k == "models" or k.startswith("models.") or k in ("losses", "kornia", "kornia.utils", "kornia.utils.grid", "torch_scatter"):
            del sys.modules[k]
    for k, v in saved.items():
        if v is not None:
            sys.modules[k] = v
        else:
            sys.modules.pop(k, None)


def test_the_references_modules_import_on_the_binding(ref):
    import ngp_pl_amd.vren as vren
    # every vren.* the reference's sources mention exists in the binding (binding.cpp:234-250 exports 12 functions)
    used = set()
    for mod in (ref.rendering, ref.networks, ref.custom_functions, ref.losses):
        src = inspect.getsource(mod)
        i = 0
        while True:
            i = src.find("vren.", i)
            if i < 0:
                break
            j = i + 5
            while j < len(src) and (src[j].isalnum() or src[j] == "_"):
                j += 1
            used.add(src[i + 5:j]); i = j
    assert used == {"ray_aabb_intersect", "ray_sphere_intersect", "packbits", "morton3D", "morton3D_invert", "raymarching_train",
                    "raymarching_test", "composite_train_fw", "composite_train_bw", "composite_test_fw", "distortion_loss_fw",
                    "distortion_loss_bw"}
    for name in used:
        assert callable(getattr(vren, name)), name
    assert ref.rendering.MAX_SAMPLES == 1024 and ref.rendering.NEAR_DISTANCE == 0.01


def test_the_references_ngp_constructs_on_the_binding_with_the_same_state_dict(ref):
    from ngp_pl_amd.networks import NGP as OurNGP
    for kwargs in (dict(scale=0.5), dict(scale=16.0), dict(scale=0.5, rgb_act="None")):
        theirs = ref.networks.NGP(**kwargs)                     # the reference's class, tinycudann = ngp_pl_amd.tcnn
        ours = OurNGP(**kwargs)
        a, b = theirs.state_dict(), ours.state_dict()
        assert list(a) == list(b), (list(a), list(b))
        for k in a:
            assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype, k
        assert theirs.cascades == ours.cascades and theirs.grid_size == ours.grid_size
        assert torch.equal(a["xyz_encoder.params"], b["xyz_encoder.params"])        # same seed, same initialisation (tiny-cuda-nn's default seed 1337)
    # the operator classes the reference defines bind the same native names with the same positional arguments
    for name in ("RayAABBIntersector", "RaySphereIntersector", "RayMarcher", "VolumeRenderer", "TruncExp"):
        assert hasattr(ref.custom_functions, name)
    theirs_sig = inspect.signature(ref.rendering.render)
    from ngp_pl_amd.rendering import render
    assert list(theirs_sig.parameters) == list(inspect.signature(render).parameters)
    # NeRFLoss: same constructor defaults and dictionary keys on CPU tensors
    lt, lo = ref.losses.NeRFLoss(), importlib.import_module("ngp_pl_amd.losses").NeRFLoss()
    assert (lt.lambda_opacity, lt.lambda_distortion) == (lo.lambda_opacity, lo.lambda_distortion)
