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


def extract_model_state_dict(ckpt_path, model_name="model", prefixes_to_ignore=[]):
    checkpoint = torch.load(ckpt_path, map_location="cpu") if not isinstance(ckpt_path, dict) else ckpt_path
    checkpoint_ = {}
    if "state_dict" in checkpoint:      # a pytorch-lightning checkpoint
        checkpoint = checkpoint["state_dict"]
    for k, v in checkpoint.items():
        if not k.startswith(model_name):
            continue
        k = k[len(model_name) + 1:]
        for prefix in prefixes_to_ignore:
            if k.startswith(prefix):
                break
        else:
            checkpoint_[k] = v
    return checkpoint_


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


def load_ckpt(model, ckpt_path, model_name="model", prefixes_to_ignore=[]):
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


def slim_ckpt(ckpt_path, save_poses=False):
    ckpt = torch.load(ckpt_path, map_location="cpu") if not isinstance(ckpt_path, dict) else ckpt_path
    keys_to_pop = ["directions", "model.density_grid", "model.grid_coords"]
    if not save_poses:
        keys_to_pop += ["poses"]
    for k in ckpt["state_dict"]:
        if k.startswith("val_lpips"):
            keys_to_pop += [k]
    for k in keys_to_pop:
        ckpt["state_dict"].pop(k, None)
    return ckpt["state_dict"]
