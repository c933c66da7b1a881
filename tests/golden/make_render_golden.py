"""Regenerates tests/golden/render_golden.npz by running the reference's OWN Python -- render() (both branches),
RayMarcher / VolumeRenderer, NeRFLoss, NGP.update_density_grid -- on the CPU (tests/golden/ref_harness.py: the
reference's kernels compiled for the host behind `vren`, oracle/tcnn_oracle.py behind `tinycudann`).  Only runs
where /root/reference is mounted; the fixture and this script travel with the repo.  Field parameters and occupancy
bitfields are NOT stored (98 MB): they are regenerated from seeds by the consumer (tests/test_reference_python_cpu.py),
and a checksum in the fixture tells whether the torch/numpy random streams still match."""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H                                   # noqa: E402
from ngp_pl_amd import synthetic as syn                   # noqa: E402
from oracle import tcnn_oracle as T                       # noqa: E402

from render_cases import CONFIGS, field_checksum, make_field, make_rays   # noqa: E402  (shared with the consuming test)


def main():
    warnings.filterwarnings("ignore")
    networks, rendering, losses = H.load_reference(fma=True)
    out = {}
    for tag, c in CONFIGS.items():
        field = make_field(c["scale"])
        model = networks.NGP(scale=c["scale"])
        H.load_field_params(model, field)
        bf = syn.random_blob_bitfield(model.cascades, 128, c["fill"], seed=31)
        model.density_bitfield.copy_(torch.from_numpy(bf))
        ro, rd = make_rays(c["n"], c["scale"], seed=7)
        kw = {} if c["esf"] == 0 else {"exp_step_factor": c["esf"]}
        res = rendering.render(model, ro, rd, test_time=True, **kw)
        for k in ("opacity", "depth", "rgb"):
            out["%s_test_%s" % (tag, k)] = res[k].numpy()
        out[tag + "_test_total_samples"] = np.int64(int(res["total_samples"]))
        # train branch: the marcher draws its jitter with torch.rand_like -- record it
        drawn = []
        real = torch.rand_like
        torch.rand_like = lambda t, *a, **k: drawn.append(real(t, *a, **k)) or drawn[-1]
        try:
            tr = rendering.render(model, ro, rd, test_time=False, **kw)
        finally:
            torch.rand_like = real
        assert len(drawn) == 1
        out[tag + "_noise"] = drawn[0].numpy()
        for k in ("opacity", "depth", "rgb", "ws", "deltas", "ts", "rays_a"):
            out["%s_train_%s" % (tag, k)] = tr[k].detach().numpy()
        out[tag + "_train_rm_samples"] = np.int64(int(tr["rm_samples"])); out[tag + "_train_vr_samples"] = np.int64(int(tr["vr_samples"]))
        # NeRFLoss on the train results (losses.py:40-60), with the distortion term
        gt = torch.rand(c["n"], 3, generator=torch.Generator().manual_seed(9))
        terms = losses.NeRFLoss(lambda_opacity=1e-3, lambda_distortion=1e-3)(tr, {"rgb": gt})
        out[tag + "_gt"] = gt.numpy()
        for k, v in terms.items():
            out["%s_loss_%s" % (tag, k)] = v.detach().numpy()
        out[tag + "_field_checksum"] = field_checksum(field)
        out[tag + "_bitfield_sum"] = np.int64(int(bf.astype(np.int64).sum()))
        n_iter_hint = int((tr["rays_a"][:, 2] > 0).sum())
        print("%s: %d rays (%d hit samples), test total_samples %d, train rm %d vr %d, opacity>0.99: %d" % (
            tag, c["n"], n_iter_hint, int(res["total_samples"]), int(tr["rm_samples"]), int(tr["vr_samples"]), int((res["opacity"] > 0.99).sum())))

    # occupancy maintenance (networks.py:156-195,240-268) on a 32^3 grid so that the fixture stays small
    G = 32
    field = make_field(0.5)
    model = networks.NGP(scale=0.5)
    H.load_field_params(model, field)
    model.grid_size = G
    model.register_buffer("density_bitfield", torch.zeros(model.cascades * G ** 3 // 8, dtype=torch.uint8))
    model.register_buffer("density_grid", torch.zeros(model.cascades, G ** 3))                            # train.py:73-76
    import kornia
    model.register_buffer("grid_coords", kornia.utils.grid.create_meshgrid3d(G, G, G, False, dtype=torch.int32).reshape(-1, 3))
    thr = 0.01 * 1024 / 3 ** 0.5                                                                           # train.py:161
    torch.manual_seed(77)
    for step, warmup in enumerate((True, True, False, False)):       # steps 0 and 2 are stored (one of each kind)
        seen = []
        real_density = model.density
        model.density = lambda x, **k: seen.append((x.clone(), real_density(x, **k))) or seen[-1][1]
        cells = []
        real_all, real_smp = model.get_all_cells, model.sample_uniform_and_occupied_cells
        model.get_all_cells = lambda: cells.append(real_all()) or cells[-1]
        model.sample_uniform_and_occupied_cells = lambda M, t: cells.append(real_smp(M, t)) or cells[-1]
        before = model.density_grid.clone()
        model.update_density_grid(thr, warmup=warmup)
        del model.density, model.get_all_cells, model.sample_uniform_and_occupied_cells
        (idx, coords), = cells[0]
        if step in (0, 2):
            out["occ%d_before" % step] = before[0].numpy(); out["occ%d_after" % step] = model.density_grid[0].numpy()
            out["occ%d_bitfield" % step] = model.density_bitfield.numpy().copy()
            out["occ%d_cells" % step] = idx.numpy().astype(np.int32); out["occ%d_sigma" % step] = seen[0][1].float().numpy()
            # the sampled point of every cell lies inside that cell (networks.py:252-255)
            s_, hgs = 0.5, 0.5 / G
            centre = (coords.float() / (G - 1) * 2 - 1) * (s_ - hgs)
            assert float((seen[0][0] - centre).abs().max()) <= hgs * (1 + 1e-5)
        print("occupancy step %d warmup=%s: %d cells, %d occupied bits" % (step, warmup, len(idx), int(np.unpackbits(model.density_bitfield.numpy()).sum())))
    out["occ_threshold"] = np.float64(thr)

    # NGP.mark_invisible_cells (networks.py:197-238) for a three-cascade scene, again on a 32^3 grid
    vis = networks.NGP(scale=2.0)
    vis.grid_size = G
    vis.register_buffer("density_grid", torch.zeros(vis.cascades, G ** 3))
    vis.register_buffer("grid_coords", kornia.utils.grid.create_meshgrid3d(G, G, G, False, dtype=torch.int32).reshape(-1, 3))
    K = syn.intrinsics(64)
    poses = syn.hemisphere_poses(6, radius=2.5, seed=4)
    vis.mark_invisible_cells(K, poses, (64, 64))
    out["vis_K"] = K.numpy(); out["vis_poses"] = poses.numpy()
    out["vis_density_grid"] = vis.density_grid.numpy().astype(np.int8)              # 0 / -1
    out["vis_count_grid"] = np.round(vis.count_grid.numpy() * 6).astype(np.uint8)  # cameras that see the cell
    print("mark_invisible_cells: %d of %d cells invisible" % (int((vis.density_grid < 0).sum()), vis.density_grid.numel()))

    # RayMarcher.backward (custom_functions.py:98-112): gradients of the packed sample positions/directions -> rays
    from models.custom_functions import RayMarcher
    c = CONFIGS["syn"]
    model = networks.NGP(scale=c["scale"])
    bf = syn.random_blob_bitfield(model.cascades, 128, c["fill"], seed=31)
    ro, rd = make_rays(c["n"], c["scale"], seed=7)
    ro.requires_grad_(True); rd.requires_grad_(True)
    _, hits_t, _ = sys.modules["vren"].ray_aabb_intersect(ro.detach(), rd.detach(), model.center, model.half_size, 1)
    torch.manual_seed(3)
    rays_a, xyzs, dirs, deltas, ts, total = RayMarcher.apply(ro, rd, hits_t[:, 0].contiguous(), torch.from_numpy(bf), model.cascades, c["scale"],
                                                             0.0, 128, 1024)
    gx = torch.randn(xyzs.shape, generator=torch.Generator().manual_seed(12)); gd = torch.randn(dirs.shape, generator=torch.Generator().manual_seed(13))
    torch.autograd.backward([xyzs, dirs], [gx, gd])
    out["rmb_rays_a"] = rays_a.detach().numpy(); out["rmb_ts"] = ts.detach().numpy(); out["rmb_gx"] = gx.numpy(); out["rmb_gd"] = gd.numpy()
    out["rmb_d_rays_o"] = ro.grad.numpy(); out["rmb_d_rays_d"] = rd.grad.numpy()
    path = os.path.join(HERE, "render_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
