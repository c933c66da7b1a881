"""bench.py -- train rays/s (+ render FPS) of the MI355X-native Instant-NGP hot path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line from rank 0.
  * step      = one full optimisation step on a batch of 8192 rays per GPU: occupancy-grid
                update every 16 steps, AABB, ray march, hash-grid encode, fused MLPs, composite,
                loss, full backward, fused Adam (BASELINE.json configs[1]: Lego-like 800x800,
                8192 rays/batch, scale 0.5).  Inputs (rays, ground-truth colours) are resident in
                HBM before the timed region.
  * value     = rays/s over all ranks (weak scaling: 8192 rays per GPU), K timed steps bracketed
                by barrier + synchronize, max over ranks.
  * roofline  = the dominant kernel of the step (hash-grid encode forward+backward by time):
                algorithmic bytes / measured kernel time vs HBM peak (8 TB/s).
  * cpu_baseline = the CPU oracle (reference kernels compiled for the host when available, our
                restatement otherwise) timed on rank 0 on a bounded sample of the same workload.
Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL), per-rank independent ray
batches, gradient all-reduce of the native gradient buffers (f16 grid gradient 22.9 MB + MLP
gradients) every step -- the reference's only collective (DDP, train.py:270-272).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=320)      # past the 256-step occupancy warm-up (train.py:58)
    p.add_argument("--rays", type=int, default=8192)
    p.add_argument("--res", type=int, default=800)
    p.add_argument("--images", type=int, default=100)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-render", action="store_true")
    p.add_argument("--timed-only", action="store_true", help="stop after the timed loop (for rocprofv3 runs: the trace then ends with the K timed steps)")
    return p.parse_args()


HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 achievable
MFMA_F16_PEAK_TFLOPS = 2500.0


def kernel_roofline(trainer, draw, n_steps=10):
    """Stage times of real training steps from HIP events on the stream the kernels run on
    (torch's current stream), then the roofline of the dominant stage.  Algorithmic bytes per
    unit are SURVEY.md section 8(d)'s (restated in DESIGN.md)."""
    n_params = trainer.model.xyz_encoder.params.numel() + trainer.model.rgb_net.params.numel()
    acc, S_acc, R = {}, 0, 0
    cur = draw()
    trainer.events = []
    for _ in range(n_steps):
        nxt = draw()
        trainer.step(cur[0], cur[1], cur[2], next_batch=(nxt[0], nxt[1]))
        for name, ms in trainer.stage_times_ms():
            acc[name] = acc.get(name, 0.0) + ms
        S_acc += trainer.last["rm_samples"]; R = trainer.last["n_rays"]
        cur = nxt
    trainer.events = None
    S = S_acc / n_steps
    algo = {   # bytes per launch
        "march_count(side stream)": 60.0 * R + 4.0 * S, "march_write": 32.0 * S, "hashgrid_fwd": 588.0 * S, "mlp_fwd": 210.0 * S,
        "composite_fw+loss": 28.0 * S + 52.0 * R, "composite_bw": 52.0 * S + 64.0 * R, "mlp_bwd": 300.0 * S,
        "hashgrid_bwd": 1100.0 * S, "adam": 30.0 * n_params, "grid_update": 0.0,
    }
    stages = []
    for name, ms in acc.items():
        ms /= n_steps
        stages.append({"stage": name, "ms": round(ms, 4), "GB/s": round(algo.get(name, 0.0) / (ms * 1e-3) / 1e9, 1) if ms > 0 else None})
    stages.sort(key=lambda d: -d["ms"])
    top = next(d for d in stages if d["stage"] not in ("grid_update", "march_count(side stream)"))   # the march overlaps the main stream
    achieved = top["GB/s"]
    # HBM bytes per launch of that kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of this
    # same command; summary and calibration in profiles/r01_pmc_hbm_traffic.txt).  Not measurable from inside this process.
    # hashgrid_bwd = binning pass + slice owners of the binned variant (merge and gather kernels are below the summary's cut)
    pmc = {"hashgrid_bwd": (116758 + 20018 + 39305 + 41436) * 1024, "hashgrid_fwd": (43901 + 30061) * 1024, "adam": (2 * 78269.2 + 178828.8) * 1024}
    return {"bound": "hbm", "kernel": top["stage"], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": pmc.get(top["stage"]), "traffic_source": "profiles/r01_pmc_hbm_traffic.txt",
            "avg_ms": top["ms"], "samples_per_launch": S, "stages": stages}


def api_path_rate(trainer, draw, n_steps=30):
    """The same step driven through the reference-shaped surface: render() -> NeRFLoss -> torch autograd -> FusedAdam
    (Trainer.step_autograd), i.e. what train.py would exercise.  Secondary number, not `value`."""
    for _ in range(5):
        b = draw(); trainer.step_autograd(b[0], b[1], b[2])
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n_steps):
        b = draw(); trainer.step_autograd(b[0], b[1], b[2])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n_steps
    return {"rays_per_s": b[0].shape[0] / dt, "ms_per_step": dt * 1e3, "what": "render()+NeRFLoss+autograd+FusedAdam, same kernels"}


def cpu_baseline(model, data, budget_s=20.0):
    """The CPU oracle timed on the host cores on BASELINE.json configs[0] (256 rays/batch): full
    step = AABB + march + composite fwd/bwd with the reference's own kernels compiled for the CPU
    (oracle/_ref, falls back to our C restatement) + hash grid / MLPs / SH / Adam as fp32 torch-CPU
    (the tiny-cuda-nn restatement, which is why kind = "port").  Checker code only: this is the one
    place outside tests/ and smoke() that touches oracle/."""
    import numpy as np
    from oracle import tcnn_oracle as T
    from oracle.vren_oracle import Oracle, Reference
    cores = min(os.cpu_count(), 32)     # tiny tensors: more threads only add synchronisation cost
    torch.set_num_threads(cores)
    vr = Reference(True) if Reference.available(True) else Oracle(True)
    field = T.Field(scale=0.5)
    enc = model.xyz_encoder
    field.density_w = enc.params.detach()[:enc.n_mlp].cpu().clone().requires_grad_(True)
    field.table = enc.params.detach()[enc.n_mlp:].cpu().view(-1, 2).clone().requires_grad_(True)
    field.rgb_w = model.rgb_net.params.detach().cpu().clone().requires_grad_(True)
    opt = torch.optim.Adam([field.density_w, field.table, field.rgb_w], lr=1e-2, eps=1e-15)
    bitfield = model.density_bitfield.cpu().numpy()
    gen = torch.Generator(device=data.device); gen.manual_seed(7)
    R = 256
    c = np.zeros((1, 3), np.float32); hs = np.full((1, 3), 0.5, np.float32)
    n_done, S_tot, t0 = -1, 0, time.perf_counter()      # step -1 is an untimed warm-up (lazy inits)
    while n_done < 0 or (time.perf_counter() - t0 < budget_s and n_done < 100):
        ro, rd, gt = (t.cpu() for t in data.sample(R, gen))
        t_step = time.perf_counter()
        _, hits_t, _ = vr.ray_aabb_intersect(ro.numpy(), rd.numpy(), c, hs, 1)
        ht = hits_t[:, 0].copy(); m = (ht[:, 0] >= 0) & (ht[:, 0] < 0.01); ht[m, 0] = 0.01
        noise = np.random.rand(R).astype(np.float32)
        rays_a, xyzs, dirs, deltas, ts, _ = vr.raymarching_train(ro.numpy(), rd.numpy(), ht, bitfield, 1, 0.5, 0.0, noise, 128, 1024)
        sig, rgb, _ = field.forward(torch.from_numpy(xyzs), torch.from_numpy(dirs))
        total, opacity, depth, crgb, ws = vr.composite_train_fw(sig.detach().numpy(), rgb.detach().numpy(), deltas, ts, rays_a, 1e-4)
        o = torch.from_numpy(opacity); col = torch.from_numpy(crgb) + (1 - o)[:, None]
        dcol = 2 * (col - gt) / (3 * R)
        oe = o + 1e-10
        do = -(dcol.sum(1)) + 1e-3 * (-(torch.log(oe) + 1)) / R
        dsig, drgbs = vr.composite_train_bw(do.numpy(), np.zeros(R, np.float32), dcol.numpy(), np.zeros_like(ws), sig.detach().numpy(),
                                            rgb.detach().numpy(), ws, deltas, ts, rays_a, opacity, depth, crgb, 1e-4)
        opt.zero_grad(set_to_none=True)
        torch.autograd.backward([sig, rgb], [torch.from_numpy(dsig), torch.from_numpy(drgbs)])
        opt.step()
        n_done += 1
        if n_done == 0:
            t0 = time.perf_counter()
        else:
            S_tot += ts.shape[0]
    dt = time.perf_counter() - t0
    return {"value": R * n_done / dt, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": "%d full training steps of 256 rays (BASELINE configs[0] batch) on the same scene/occupancy grid, %.1f samples/ray, %.1f s; "
                      "vren kernels = %s, tiny-cuda-nn parts = fp32 torch-CPU restatement with autograd + torch Adam" % (
                          n_done, S_tot / max(n_done, 1) / R, dt,
                          "reference .cu compiled for CPU (oracle/_ref)" if isinstance(vr, Reference) else "oracle/ngp_oracle.c")}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0)); local = int(os.environ.get("LOCAL_RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or "RANK" in os.environ:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    from ngp_pl_amd import synthetic as syn
    from ngp_pl_amd.bench_support import GpuDataset, all_reduce_native, all_reduce_native_mlp, render_fps
    from ngp_pl_amd.networks import NGP
    from ngp_pl_amd.trainer import Trainer

    torch.manual_seed(1337)
    model = NGP(scale=0.5).to(dev)
    model.register_training_buffers()
    trainer = Trainer(model, lr=1e-2, num_epochs=30)
    if dist is not None:   # also with a 1-rank process group (torchrun --nproc-per-node 1): exercises the collective path
        trainer.grad_hook = lambda: all_reduce_native(model, dist, world)
        trainer.mlp_grad_hook = lambda: all_reduce_native_mlp(model, dist)      # small collective hidden under the hash-grid backward
        # identical initial parameters on every rank (DDP broadcasts rank 0's)
        for p in model.parameters():
            dist.broadcast(p.data, 0)
        model.xyz_encoder._half.invalidate(); model.rgb_net._half.invalidate()     # f16 working copies follow the broadcast
    data = GpuDataset(args.res, args.images, dev, seed=0)                 # synthetic Lego-like scene, GT resident in HBM
    gen = torch.Generator(device=dev); gen.manual_seed(1234 + rank)       # per-rank independent batches (base.py:25-29)

    draw_count = [0]
    main_stream = torch.cuda.current_stream()

    def draw(on_side=True):
        # the batch sampler runs on the trainer's marching stream, in front of the march that consumes its rays (the
        # main stream picks the batch up behind that march's event); record_stream: the main stream reads them too
        draw_count[0] += 1
        if trainer.side is None or not on_side:
            return data.sample_native(args.rays, draw_count[0], seed=1234 + rank)
        with torch.cuda.stream(trainer.side):
            batch = data.sample_native(args.rays, draw_count[0], seed=1234 + rank)
        for t in batch:
            t.record_stream(main_stream)
        return batch

    cur = draw()
    for _ in range(args.warmup):
        nxt = draw()
        trainer.step(cur[0], cur[1], cur[2], next_batch=(nxt[0], nxt[1]))
        cur = nxt
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        nxt = draw()
        trainer.step(cur[0], cur[1], cur[2], next_batch=(nxt[0], nxt[1]))
        cur = nxt
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    met = trainer.metrics()
    rays_per_s = args.rays * world * args.steps / dt

    out = {
        "metric": "train rays/sec (800x800 Lego-like, 8192 rays/batch/GPU, full step incl. optimizer)",
        "value": rays_per_s, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16/f32", "dtype_detail": "hash tables, features, MLP operands f16 with f32 MFMA/blend accumulation; march, composite, Adam f32", "data": "synthetic (procedural Lego-like scene, random-init weights)",
        "config": {"workload": "configs[1]: Synthetic-NeRF Lego-like, 1xMI355X per rank, 8192 rays/batch, 800x800, scale 0.5",
                   "rays_per_gpu": args.rays, "image_res": args.res, "n_images": args.images,
                   "samples_per_ray_marched": met["rm_s"], "samples_per_ray_composited": met["vr_s"], "train_psnr": met["psnr"],
                   "parallelism": "dp%d (per-ray data parallel, native-gradient all-reduce)" % world},
    }
    trainer.grad_hook = trainer.mlp_grad_hook = None      # what follows runs on rank 0 only: no collectives from here on
    if rank == 0 and args.timed_only:
        print(json.dumps(out))
    elif rank == 0:
        if not args.no_render:
            # device-driven frame loop; chunk_scale/probe_cap only regroup the SAME per-ray samples into fewer
            # iterations (tests/test_train_gpu.py::test_device_frame_loop_matches_host_loop)
            out["render_fps_800x800"] = render_fps(model, data, n_frames=5, chunk_scale=4, probe_cap=64)
            out["render_fps_800x800"]["loop"] = "ngp_render_test_frame chunk_scale=4 probe_cap=64"
        out["roofline"] = kernel_roofline(trainer, draw)
        out["api_path"] = api_path_rate(trainer, lambda: draw(on_side=False))
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(model, data)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
