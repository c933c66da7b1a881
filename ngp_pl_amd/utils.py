"""Checkpoint surface of the reference (/root/reference/utils.py:4-39): `extract_model_state_dict`, `load_ckpt`,
`slim_ckpt` with the same signatures and key handling (Lightning checkpoints keep the model under the `model.`
prefix; the slim checkpoint drops `directions`, `model.density_grid`, `model.grid_coords`, `poses`, `val_lpips*`).

One thing the reference cannot tell us: the length of `xyz_encoder.params` in a released checkpoint.  tiny-cuda-nn
builds its level table in float32 (`exp2f(l * log2f(b)) * 16 - 1`), which for the reference's b lands a few 1e-6 ABOVE
the integers 63, 255 and 1023 at levels 5, 10 and 15 (resolution 65 / 257 / 1025: 11 448 112 parameters at scale 0.5);
evaluated in exact arithmetic the same formula gives 64 / 256 / 1024 (11 423 136, the figure in SURVEY.md).  The two layouts are not
convertible (different grid spacing on three levels), so `load_ckpt` refuses a mismatch with the remedy spelled out:
construct `NGP(scale, level_table=...)` with the table the checkpoint was trained with.
"""
import torch

from . import tcnn


def _open(ckpt):
    """A checkpoint given as a path or as the loaded dictionary."""
    return ckpt if isinstance(ckpt, dict) else torch.load(ckpt, map_location="cpu")


def extract_model_state_dict(ckpt_path, model_name="model", prefixes_to_ignore=()):
    """The entries of a checkpoint that belong to `<model_name>.`, with that prefix removed; a Lightning checkpoint keeps its
    tensors under "state_dict".  Keys that start with one of `prefixes_to_ignore` (after the model prefix) are left out."""
    tensors = _open(ckpt_path)
    tensors = tensors.get("state_dict", tensors)
    head = model_name + "."
    skip = tuple(prefixes_to_ignore)
    return {k[len(head):]: v for k, v in tensors.items()
            if k.startswith(model_name) and not (skip and k[len(head):].startswith(skip))}


def grid_param_count(encoding_config, level_table):
    meta = tcnn.make_grid_meta(encoding_config, level_table)
    return int(meta.offset[int(meta.n_levels)]) * 2


def check_encoder_length(model, n_ckpt):
    """Raise with a precise, actionable message if a checkpoint's `xyz_encoder.params` does not fit this model."""
    enc = model.xyz_encoder
    n_model = enc.params.numel()
    if n_ckpt == n_model:
        return
    for other in ("float32", "exact"):
        if other != enc.level_table and n_ckpt == enc.n_mlp + grid_param_count(enc.encoding_config, other):
            raise RuntimeError(
                "checkpoint xyz_encoder.params has %d entries: that is the hash grid's level table evaluated in %s arithmetic "
                "(this model uses %r: %d entries; the tables differ in the resolution of three levels, so the parameters cannot "
                "be converted).  Construct the model with NGP(scale=%g, level_table=%r) to load this checkpoint."
                % (n_ckpt, other, enc.level_table, n_model, model.scale, other))
    raise RuntimeError("checkpoint xyz_encoder.params has %d entries, this model (scale %g, level_table %r) has %d: different "
                       "scale / hash-grid configuration?" % (n_ckpt, model.scale, enc.level_table, n_model))


def load_ckpt(model, ckpt_path, model_name="model", prefixes_to_ignore=()):
    if not ckpt_path:
        return
    model_dict = model.state_dict()
    checkpoint_ = extract_model_state_dict(ckpt_path, model_name, prefixes_to_ignore)
    if "xyz_encoder.params" in checkpoint_:
        check_encoder_length(model, checkpoint_["xyz_encoder.params"].numel())
    model_dict.update(checkpoint_)
    model.load_state_dict(model_dict)
    for mod in model.modules():          # f16 working copies follow the new master parameters
        if hasattr(mod, "_half"):
            mod._half.invalidate()


_NOT_IN_A_SLIM_CKPT = ("directions", "model.density_grid", "model.grid_coords")     # rebuilt from the dataset / by training start


def slim_ckpt(ckpt_path, save_poses=False):
    """The state dict of a Lightning checkpoint without what inference does not need: ray directions, the float occupancy grid
    and its cell coordinates, the LPIPS network, and (unless `save_poses`) the camera poses.  Edits the checkpoint in place, as
    the reference does."""
    sd = _open(ckpt_path)["state_dict"]
    doomed = set(_NOT_IN_A_SLIM_CKPT) | ({"poses"} if not save_poses else set()) | {k for k in sd if k.startswith("val_lpips")}
    for k in doomed:
        sd.pop(k, None)
    return sd
