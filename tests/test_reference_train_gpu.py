"""GPU: the reference's OWN train.py, unmodified, executed end to end as `python train.py --root_dir ... --num_epochs 1 ...` on this
package's bindings (oracle/ref_train_harness.py: stand-ins for the packages the image lacks -- pytorch_lightning, torchmetrics, kornia,
cv2, imageio; `vren`, `tinycudann` and apex's FusedAdam are the product).  What runs: opt.get_opts, NeRFSystem.__init__ / setup /
configure_optimizers / on_train_start (mark_invisible_cells) / 1000 x training_step under autocast + GradScaler (train.py:159-185 with
update_density_grid every 16 steps) / CosineAnnealingLR / the validation loop with render(test_time=True) (train.py:193-237) /
ModelCheckpoint / utils.slim_ckpt -- every kernel libngp_hip.so's.  north_star: "so train.py / pytorch-lightning drives it unchanged"."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_the_references_train_py_trains_and_validates_on_the_binding(tmp_path):
    from oracle import ref_train_harness as H
    if not H.available():
        pytest.skip("the reference's train.py is neither mounted nor staged (oracle/build_ref.sh)")
    torch.manual_seed(0); np.random.seed(0)
    g = H.run_train(["--root_dir", "procedural", "--dataset_name", "nsvf", "--exp_name", "harness", "--num_epochs", "1", "--batch_size", "4096",
                     "--no_save_test"], str(tmp_path), res=200, n_train=24, n_test=4)
    system = g["system"]
    assert type(system).__name__ == "NeRFSystem" and type(system.model).__module__ == "models.networks"
    assert system.global_step == 1000                                       # datasets/base.py:17-19: an epoch is 1000 batches
    log = system.logged
    for k in ("lr", "train/loss", "train/rm_s", "train/vr_s", "train/psnr", "test/psnr", "test/ssim"):
        assert k in log and math.isfinite(log[k]), (k, log)
    assert log["test/psnr"] > 24.0 and log["train/psnr"] > 20.0, log       # 1000 steps of 4096 rays on 24 images of 200 x 200
    assert 1.0 < log["train/rm_s"] < 200 and 0 < log["train/vr_s"] <= log["train/rm_s"] + 1e-6
    assert abs(log["lr"] - 1e-2) < 1e-9                                     # (the cosine schedule steps per epoch: one epoch, logged before its step)
    # the occupancy grid was maintained by the reference's update_density_grid on the binding
    grid = system.model.density_grid
    assert float((grid > 0).float().mean()) > 0.001 and int(torch.count_nonzero(system.model.density_bitfield)) > 0
    # checkpoint + slimmed checkpoint, loadable by the product
    slim_path = tmp_path / "ckpts" / "nsvf" / "harness" / "epoch=0_slim.ckpt"
    assert slim_path.exists()
    from ngp_pl_amd.networks import NGP
    from ngp_pl_amd.rendering import render
    from ngp_pl_amd.utils import load_ckpt
    from ngp_pl_amd import synthetic as syn
    from ngp_pl_amd.bench_support import surface_ground_truth
    ours = NGP(scale=0.5).cuda()
    ours.register_training_buffers()
    load_ckpt(ours, str(slim_path))
    ours.density_grid.copy_(system.model.density_grid); ours.density_bitfield.copy_(system.model.density_bitfield)
    pose = syn.hemisphere_poses(1, seed=4242)[0].cuda()
    ro, rd = syn.get_rays(syn.get_ray_directions(200, 200, syn.intrinsics(200)).cuda(), pose)
    with torch.no_grad():
        out = render(ours, ro.contiguous(), rd.contiguous(), test_time=True)
    gt = surface_ground_truth(ro, rd)
    psnr = -10.0 * math.log10(float(((out["rgb"] - gt) ** 2).mean()))
    assert psnr > 22.0, psnr                                                # the field the reference's script trained renders through the product's own render()
    line = "reference train.py on the binding: %d steps x 4096 rays, 24 images of 200 x 200: %s; held-out pose through the product's render(): %.2f dB" % (
        system.global_step, ", ".join("%s %.4g" % (k, v) for k, v in sorted(log.items())), psnr)
    print(line)
    if os.environ.get("NGP_PARITY_LOG"):
        with open(os.environ["NGP_PARITY_LOG"], "a") as f:
            f.write(line + "\n")
