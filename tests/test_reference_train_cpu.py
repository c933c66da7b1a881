"""CPU: the reference's own train.py up to (not including) its training steps -- oracle/ref_train_harness.py in dry mode: the script is
executed as __main__, parses its command line with the reference's opt.py, builds NeRFSystem (the reference's NGP over this package's
`tinycudann` surface), sets up the procedural dataset through the reference's datasets/base.py + ray_utils.py, writes the Lightning-shaped
checkpoint and slims it with the reference's utils.slim_ckpt.  The steps themselves run on the GPU: tests/test_reference_train_gpu.py."""
import os

import pytest
import torch


def test_the_references_train_py_runs_up_to_its_steps(tmp_path, monkeypatch):
    from oracle import ref_train_harness as H
    if not H.available():
        pytest.skip("the reference's train.py is neither mounted nor staged (oracle/build_ref.sh)")
    monkeypatch.setenv("NGP_HARNESS_DRY", "1")
    g = H.run_train(["--root_dir", "procedural", "--dataset_name", "nsvf", "--exp_name", "dry", "--num_epochs", "2", "--batch_size", "512", "--no_save_test"],
                    str(tmp_path), res=48, n_train=3, n_test=2)
    system, hp = g["system"], g["hparams"]
    assert type(system).__name__ == "NeRFSystem" and type(system.model).__module__ == "models.networks"
    assert hp.batch_size == 512 and hp.num_epochs == 2 and system.warmup_steps == 256 and system.update_interval == 16
    ds = system.train_dataset
    assert ds.batch_size == 512 and ds.rays.shape == (3, 48 * 48, 3) and ds.directions.shape == (48 * 48, 3) and len(ds) == 1000
    sample = ds[0]                                                         # drawn by the reference's BaseDataset.__getitem__
    assert sample["rgb"].shape == (512, 3) and sample["img_idxs"].shape == (512,) and sample["pix_idxs"].max() < 48 * 48
    # the Lightning-shaped checkpoint and its slimmed form (train.py:282-286 with utils.slim_ckpt)
    full = torch.load(tmp_path / "ckpts" / "nsvf" / "dry" / "epoch=1.ckpt")
    slim = torch.load(tmp_path / "ckpts" / "nsvf" / "dry" / "epoch=1_slim.ckpt")
    assert "model.density_grid" in full["state_dict"] and "model.density_grid" not in slim and "model.xyz_encoder.params" in slim
    # ... loads into the product's NGP through the product's utils.load_ckpt
    from ngp_pl_amd.networks import NGP
    from ngp_pl_amd.utils import load_ckpt
    ours = NGP(scale=0.5)
    load_ckpt(ours, str(tmp_path / "ckpts" / "nsvf" / "dry" / "epoch=1_slim.ckpt"))
    assert torch.equal(ours.xyz_encoder.params.detach(), system.model.xyz_encoder.params.detach().cpu())
    assert "pytorch_lightning" not in __import__("sys").modules and "datasets" not in __import__("sys").modules      # the stand-ins are gone again
