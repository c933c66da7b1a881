"""bench.py -- train rays/s (+ render FPS) of the MI355X-native Instant-NGP hot path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line from rank 0.
  * step      = one full optimisation step on a batch of 8192 rays per GPU: occupancy-grid update every 16
                steps, AABB, ray march, hash-grid encode, fused MLPs, composite, loss, full backward, fused
                Adam (BASELINE.json configs[1]: Lego-like 800x800, 8192 rays/batch, scale 0.5).  Inputs
                (poses, directions, ground-truth images) are resident in HBM before any timed region.
  * phases    = (1) UNTIMED, DISCLOSED setup: SETUP_STEPS (320) optimisation steps from the random
                    initialisation, so that whatever K and W are, the timed region sits where SURVEY.md
                    section 8(d) defines the metric (>= 300 steps in, past the 256-step occupancy warm-up).
                    The window [W, W+K) of those steps is timed on the side and reported as `cold_start`
                    (what a literal "W warm-up steps from scratch, then K steps" measures: 15-25x more samples
                    per ray than the steady state, plus first-touch costs);
                (2) W untimed warm-up steps;
                (3) K timed steps bracketed by barrier + synchronize on both sides, max over ranks; the window
                    is repeated until >= MIN_TIMED_STEPS (200) steps are covered and `value` is rays over the
                    summed window time (`timed_windows`, `timed_steps_total` say how many).
  * value     = rays/s over all ranks (weak scaling: 8192 rays per GPU) of `Trainer.step`, the native step
                (direct C-ABI calls, no autograd graph).  `api_path` is the same step driven through the
                reference-shaped surface (render() + NeRFLoss + autograd + FusedAdam).
  * roofline  = the dominant stage of the step by time: algorithmic bytes of the work it ACTUALLY did (backward
                stages run on the active samples only, read back per step) / its HIP-event time on the stream it
                runs on, measured over ROOFLINE_STEPS steps right behind the timed windows; `traffic` = HBM bytes
                from the PMC profile recorded at the same operating point (profiles/*_pmc_traffic.json, made by
                tools/profile_bench.sh + tools/pmc_traffic.py), or null when the operating points differ by more than 10 %.
  * cpu_baseline = the CPU oracle (reference kernels compiled for the host when available, our restatement
                otherwise) timed on rank 0 on a bounded sample of the same workload.
Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL).  `--gpus N` with no RANK in the
environment re-executes itself under torch.distributed.run with N ranks (127.0.0.1 rendezvous); under
torchrun it takes RANK/LOCAL_RANK/WORLD_SIZE from the environment.  Per-rank independent ray batches, one
gradient exchange per step (ngp_pl_amd/ddp.py) -- the reference's only collective (DDP, train.py:270-272).
Without a GPU (`--dry-run`, implied when none is visible) only the launcher and the process group are
exercised (gloo): the product path has no CPU fallback.

Legs behind the headline (each with its own budget; the line is printed with whatever is complete): under a process group, on every
rank, `dp_eval` (more data-parallel steps, then the evaluation sharded over the ranks, train.py:193-237) and `exchange_modes` (the other
exchange modes on the same communicator); then on rank 0 `roofline`, `cpu_baseline`, `full_run`, the FPS legs (incl.
`render_fps_800x800_reference_files`: test.ipynb cell 2 around the reference's own rendering.py on the bindings), `api_path`,
`api_path_plain`, `api_path_reference_files` (train.py:159-185 around the reference's OWN models/*.py + losses.py, staged unmodified by
oracle/build_ref.sh and loaded by oracle/ref_on_binding.py over this package's bindings: every kernel that runs is the product's; no
oracle restatement is executed), `secondary` (configs[3] / configs[2] recipes) and `sensitivity` (rays/s against live samples per ray).

Robustness (round 2's driver run was killed at 1800 s with nothing on stdout): the ONE line is owned by a watchdog thread
(`LineKeeper`).  The headline record is handed to it the moment the timed windows are done; every further object
(`roofline`, `cpu_baseline`, the FPS legs, `api_path`) is a leg with its own wall-clock budget, run in
order of importance.  A leg that exceeds its budget, or the global deadline (NGP_BENCH_DEADLINE_S, default 270 s from
process start), gets `{"error": "timeout ..."}`, the stacks of all Python threads go to stderr (faulthandler), the line is
printed with what is complete and the process leaves through os._exit -- the line is never printed later than the deadline,
and never twice.  Progress goes to stderr unconditionally, one line per leg.
"""
import argparse
import faulthandler
import gc
import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SETUP_STEPS = 320          # untimed, disclosed (>= 300 steps, past the 256-step occupancy warm-up)
MIN_TIMED_STEPS = 200
ROOFLINE_STEPS = 20
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 achievable
WORKLOADS = {
    # name: (scene, scale, rays, lr, erode, description)
    "lego": ("lego", 0.5, 8192, 1e-2, False, "configs[1]: Lego-like 800x800, 8192 rays/batch per MI355X, scale 0.5"),
    # the same recipe on a scene that does NOT flatter early termination (ngp_pl_amd/bench_support.py:lego_hard_scene: studs, treads
    # made of 4 mm bars, a hollow cabin, finite density sigma -> volumetric ground truth); two densities for the sensitivity
    "lego_hard": ("lego_hard:60:1.0", 0.5, 8192, 1e-2, False, "configs[1] recipe on the lego_hard scene (studs, 4 mm tread lattice, hollow cabin), sigma 60"),
    "lego_hard_soft": ("lego_hard:30:1.0", 0.5, 8192, 1e-2, False, "configs[1] recipe on the lego_hard scene, sigma 30"),
    "lego_hard_big": ("lego_hard:30:1.2", 0.5, 8192, 1e-2, False, "configs[1] recipe on the lego_hard scene, sigma 30, object scaled 1.2x (fills the frame)"),
    "lego16k": ("lego", 0.5, 16384, 2e-2, False, "configs[2] recipe (benchmark_synthetic_nerf.sh:25-28): 16384 rays/batch, lr 2e-2, on the Lego-like scene"),
    "unbounded": ("unbounded", 16.0, 8192, 1e-2, True, "configs[3] recipe (benchmark_mipnerf360.sh:21-24): scale 16 -> 6 cascades, exp_step_factor 1/256, "
                  "erode, black background, on a procedural unbounded scene"),
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--workload", choices=sorted(WORKLOADS), default="lego")
    p.add_argument("--rays", type=int, default=0, help="rays per batch and GPU (default: the workload's)")
    p.add_argument("--res", type=int, default=800)
    p.add_argument("--images", type=int, default=100)
    p.add_argument("--setup-steps", type=int, default=SETUP_STEPS)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-render", action="store_true")
    p.add_argument("--secondary", action="store_true", help="(default since round 5; kept for older command lines)")
    p.add_argument("--no-secondary", action="store_true", help="skip the configs[2] / configs[3] recipes and the lego_hard sensitivity (each rebuilds a "
                   "100x800x800 dataset and runs 320 setup steps)")
    p.add_argument("--deadline", type=float, default=float(os.environ.get("NGP_BENCH_DEADLINE_S", "270")),
                   help="seconds from process start after which the line is printed with whatever is complete")
    p.add_argument("--no-api", action="store_true", help="skip the api_path legs")
    p.add_argument("--no-dp-eval", action="store_true", help="under a process group: skip the evaluation sharded over the ranks")
    p.add_argument("--no-full-run", action="store_true", help="skip the literal configs[1] run (30 000 steps from scratch + evaluation over 200 held-out poses, ~15 s)")
    p.add_argument("--timed-only", action="store_true", help="stop after the timed windows (for rocprofv3 runs: the trace then ends with the timed steps)")
    p.add_argument("--dry-run", action="store_true", help="launcher + process group only (gloo, no GPU work)")
    return p.parse_args()


# ---------------------------------------------------------------------------------------------------------
# launcher
# ---------------------------------------------------------------------------------------------------------
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` with N > 1 and no rank environment: start N ranks of this same command
    (one per GPU) the way the driver would, and hand their exit status back."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def dry_run(args, rank, world, out_stream):
    """No GPU: prove the launch path (N ranks, rendezvous on 127.0.0.1, a collective) and nothing else."""
    import torch.distributed as dist
    if world > 1 or "RANK" in os.environ:
        dist.init_process_group("gloo")
        t = torch.tensor([rank + 1.0])
        dist.all_reduce(t)
        assert float(t) == world * (world + 1) / 2
        dist.barrier()
        world_seen = dist.get_world_size()
        dist.destroy_process_group()
    else:
        world_seen = 1
    if rank == 0:
        out_stream.emit(compact_line({"metric": "train rays/sec", "value": None, "unit": "rays/s", "n_gpus": world_seen, "steps": args.steps,
                          "warmup": args.warmup, "dry_run": True, "note": "no GPU visible: launcher and process group only (gloo); "
                          "the product path has no CPU fallback"}))


# ---------------------------------------------------------------------------------------------------------
# measurement pieces
# ---------------------------------------------------------------------------------------------------------
class Loop:
    """The training loop of one workload: model, trainer, HBM-resident data, batch sampler on the marching stream."""

    def __init__(self, workload, args, dev, rank, world, dist, data=None):
        from ngp_pl_amd.bench_support import GpuDataset
        from ngp_pl_amd.networks import NGP
        from ngp_pl_amd.trainer import Trainer
        scene, scale, rays, lr, erode, self.description = WORKLOADS[workload]
        self.rays = args.rays or rays
        self.dev, self.rank, self.world, self.dist = dev, rank, world, dist
        torch.manual_seed(1337)
        self.model = NGP(scale=scale).to(dev)
        self.model.register_training_buffers()
        scene, sigma, size = (scene.split(":") + ["", ""])[:3]
        self.data = data if data is not None else GpuDataset(args.res, args.images, dev, seed=0, scene=scene, sigma=float(sigma) if sigma else None,
                                                             size=float(size) if size else 1.0)                   # ground truth resident in HBM
        if erode:      # train.py:73-76,160-163: the colmap recipe marks the cells no camera sees and erodes by visibility
            self.model.mark_invisible_cells(self.data.K.to(dev), self.data.poses, (self.data.W, self.data.H))
        self.trainer = Trainer(self.model, lr=lr, num_epochs=30 if workload != "lego16k" else 20, erode=erode)
        self.exchange = None
        if dist is not None:       # also with a 1-rank process group (torchrun --nproc-per-node 1): exercises the collective path
            # NGP_DDP_EXCHANGE: "sharded" (default: reduce-scatter -> Adam on the rank's shard -> all-gather of the f16 table),
            # "allreduce" (one all-reduce of the f16 gradient, whole-table Adam on every rank) or "direct" (the sharded schedule over
            # point-to-point transfers, ddp.DirectExchange / ngp_stepper_tail mode 2)
            # NGP_DDP_NATIVE (default 1): the exchange is enqueued by the library on its own RCCL communicator and stream
            # (ddp.NativeExchange, csrc/comm.hip + ngp_stepper_tail); 0: the torch.distributed classes (the host-side mirrors the gloo
            # tests drive).  NGP_DDP_CHUNKS / NGP_DDP_GROUPS: pieces of the grid exchange / launch groups of the table backward.
            from ngp_pl_amd.ddp import GradientExchange, NativeExchange, ShardedExchange
            self.exchange_kind = os.environ.get("NGP_DDP_EXCHANGE", "sharded")
            self.exchange_impl = "native" if os.environ.get("NGP_DDP_NATIVE", "1") != "0" else "torch.distributed"
            # one chunk, one launch group: measured on a 1-rank RCCL group (profiles/archive_r01_r04/r04_pg1_native_exchange.txt) every further launch
            # group of the table backward costs more than the reduce-scatter it could hide (2 groups +33 us, 4 groups +170 us per step)
            n_chunks = int(os.environ.get("NGP_DDP_CHUNKS", "1"))
            if self.exchange_impl == "native":
                try:
                    self.exchange = NativeExchange(self.model, dist, world, rank, mode=self.exchange_kind, n_chunks=n_chunks,
                                                   n_groups=int(os.environ.get("NGP_DDP_GROUPS", n_chunks))).install(self.trainer)
                except Exception as e:                 # noqa: BLE001 -- RCCL could not be loaded / initialised: say so and use torch's communicator
                    progress("native exchange unavailable (%s: %s): falling back to torch.distributed collectives" % (type(e).__name__, str(e)[:200]))
                    self.exchange_impl = "torch.distributed (native exchange failed: %s)" % str(e)[:120]
                    self.exchange = None
                    self.trainer.native_exchange = None
                # all ranks take the same path: one that could not build its communicator takes the others with it
                agree = torch.tensor([1 if self.exchange is not None else 0], device=dev, dtype=torch.int32)
                dist.all_reduce(agree, op=dist.ReduceOp.MIN)
                if int(agree.item()) == 0 and self.exchange is not None:
                    self.exchange.uninstall(self.trainer); self.exchange.close(); self.exchange = None
                    self.exchange_impl = "torch.distributed (native exchange failed on another rank)"
            if self.exchange is None:
                if self.exchange_kind == "sharded":
                    self.exchange = ShardedExchange(self.model, dist, world, rank).install(self.trainer)
                elif self.exchange_kind == "direct":
                    from ngp_pl_amd.ddp import DirectExchange
                    self.exchange = DirectExchange(self.model, dist, world, rank).install(self.trainer)
                else:
                    self.exchange = GradientExchange(self.model, dist, world).install(self.trainer)
            self.exchange.broadcast_parameters()
        self.draws = 0
        # three batch buffers in rotation (the batch being stepped, the one being marched, the one being drawn): nothing is
        # allocated per draw and no allocator bookkeeping across the two streams is needed
        self.ring = [tuple(torch.empty(self.rays, 3, device=dev) for _ in range(3)) for _ in range(3)]
        # The ring is allocated by torch's caching allocator in the MAIN stream's context and written by the sampler on the
        # MARCHING stream: the allocator may hand out a block that a main-stream kernel still queued (a zero-fill of a temporary of
        # the dataset / occupancy set-up above, already freed on the host) is going to write.  Without this hand-over the first batch
        # of a loop built behind bench.py's api leg came back (partly) zeroed -- rays with origin = direction = 0, whose march never
        # ended in round 2's kernels (t_target = inf): the 1800 s stall of BENCH_r02 (profiles/archive_r01_r04/r03_round2_stall_root_cause.txt).
        if self.trainer.side is not None:
            self.trainer.side.wait_stream(torch.cuda.current_stream())
        self.cur = self.draw()

    def draw(self, on_side=True):
        """Next batch (ngp_sample_rays).  It is drawn on the trainer's marching stream, in front of the march that consumes its
        rays: the main stream picks the batch up behind that march's event."""
        self.draws += 1
        tr = self.trainer
        out = self.ring[self.draws % 3]
        side = tr.side is not None and on_side
        # No extra ordering needed for the buffer reuse: the marching stream last waited on the main stream when the march of
        # the previous batch was enqueued (inside the step before this one), i.e. behind every kernel of the step that consumed
        # this buffer's previous batch, three draws back.
        return self.data.sample_native(self.rays, self.draws, seed=1234 + self.rank, out=out,      # per-rank independent batches (base.py:25-29)
                                       stream_handle=tr.side.cuda_stream if side else None)

    def steps(self, n):
        for _ in range(n):
            nxt = self.draw()
            self.trainer.step(self.cur[0], self.cur[1], self.cur[2], next_batch=(nxt[0], nxt[1]))
            self.cur = nxt

    def fence(self):
        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def timed(self, n):
        """n steps bracketed by barrier + synchronize; seconds, max over ranks.  Also the HIP-event time of the same
        window on the main stream (the stream every kernel of the step but the march is launched on).  Python's cyclic
        garbage collector is held off inside the window (a generation-2 pass was seen to stall the enqueueing thread for
        35 ms, i.e. 70 steps' worth, at a fixed step of this script)."""
        self.fence()
        gc.collect()
        gc.disable()
        try:
            return self._timed(n)
        finally:
            gc.enable()

    def _timed(self, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        self.steps(n)
        e1.record()
        self.fence()
        dt = time.perf_counter() - t0
        if self.dist is not None:
            t = torch.tensor([dt], device=self.dev, dtype=torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, e0.elapsed_time(e1) * 1e-3

    def time_exchange(self, n=20):
        """n more steps with device events around the grid exchange: `exchange_ms` = first grid collective -> table ready for the
        next forward; `exposed_exchange_ms` (native exchange) = the part of it behind the end of the table backward, i.e. what
        nothing on the main stream hides."""
        ex, tr = self.exchange, self.trainer
        if hasattr(ex, "sample_times"):
            tr.events = []
            for _ in range(n):
                self.steps(1)
                ex.sample_times()
            tr.events = None
            return {"exchange_ms": ex.exchange_ms(), "exposed_exchange_ms": ex.exposed_ms()}
        ex.timing = True
        self.steps(n)
        ex.timing = False
        return {"exchange_ms": ex.exchange_ms()}

    def other_exchange_modes(self, primary):
        """The other exchange modes on the same communicator and buffers, AFTER the headline has been handed over (a mode that has
        never run on real links must not be able to take the line with it) -- "sharded": reduce-scatter -> Adam on the rank's share
        -> all-gather (rings); "allreduce": gradient all-reduce only, the reference's semantics literally; "direct": the sharded
        schedule with point-to-point transfers over all xGMI links at once and the N slices added in rank order in f32 (round 5).
        Every rank calls this (collectives inside)."""
        modes = {self.exchange_kind: primary}
        for other in ("sharded", "allreduce", "direct"):
            if other == self.exchange_kind:
                continue
            try:
                self.exchange.switch_mode(other)
                self.steps(5)
                t = self.timed(20)[0]
                modes[other] = dict(self.time_exchange(), ms_per_step=t / 20 * 1e3, rays_per_s=self.rays * self.world * 20 / t)
            except Exception as e:             # noqa: BLE001
                modes[other] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        try:
            self.exchange.switch_mode(self.exchange_kind)
        except Exception as e:                 # noqa: BLE001
            modes["switch_back_error"] = "%s: %s" % (type(e).__name__, str(e)[:200])
        return modes

    def run(self, setup_steps, warmup, steps, min_timed=MIN_TIMED_STEPS):
        """setup (with the cold-start window inside) -> warm-up -> timed windows.  Returns the result record."""
        tr = self.trainer
        cold = None
        w0 = min(warmup, max(setup_steps - steps, 0))
        if setup_steps >= w0 + steps:
            self.steps(w0)
            dt, _ = self.timed(steps)
            met = tr.metrics()
            cold = {"rays_per_s": self.rays * self.world * steps / dt, "ms_per_step": dt / steps * 1e3,
                    "window": "steps [%d, %d) from the random initialisation" % (w0, w0 + steps),
                    "samples_per_ray_marched": met["rm_s"],
                    "what": "what `--warmup W --steps K` measures without the setup phase: inside the 256-step occupancy warm-up"}
            self.steps(setup_steps - w0 - steps)
        else:
            self.steps(setup_steps)
        self.steps(warmup)
        if hasattr(tr, "host_times"):
            tr.host_times(reset=True)              # host accounting of the native stepper: the timed windows only
        n_win = max(1, -(-min_timed // steps))
        wins, ev = [], []
        for _ in range(n_win):
            dt, dte = self.timed(steps)
            wins.append(dt); ev.append(dte)
        total = sum(wins)
        met = tr.metrics()
        extra = {}
        host = tr.host_times(reset=True) if hasattr(tr, "host_times") else None
        if host:                       # native stepper only: polling for the march's count (device-bound) vs launches / events / checks
            extra["host_ms_per_step"] = host
        if self.exchange is not None:          # the exchange stage on its own: 20 more steps with device events around it (all ranks alike)
            extra.update({"exchange": self.exchange_kind, "exchange_impl": self.exchange_impl})
            extra.update(self.time_exchange())
        return {**extra, "ms_per_step": total / (n_win * steps) * 1e3, "rays_per_s": self.rays * self.world * n_win * steps / total,
                "timed_windows": n_win, "timed_steps_total": n_win * steps, "window_ms_per_step_min_max": [min(wins) / steps * 1e3, max(wins) / steps * 1e3],
                "ms_per_step_hip_events": sum(ev) / (n_win * steps) * 1e3, "cold_start": cold, "metrics": met,
                "global_step_at_end": tr.global_step}


def kernel_roofline(loop, ms_per_step=None, n_steps=ROOFLINE_STEPS):
    """Stage times of real training steps from HIP events on the stream the kernels run on (torch's current
    stream), then the roofline of the dominant stage.  Algorithmic bytes per unit are SURVEY.md section 8(d)'s
    (restated in DESIGN.md); the backward stages are priced by the samples they process (the active list)."""
    trainer = loop.trainer
    n_params = trainer.model.xyz_encoder.params.numel() + trainer.model.rgb_net.params.numel()
    acc, S_acc, A_acc, R = {}, 0, 0, loop.rays
    trainer.events = []
    gc.collect()
    gc.disable()
    for _ in range(n_steps):
        loop.steps(1)
        for name, ms in trainer.stage_times_ms():
            acc[name] = acc.get(name, 0.0) + ms
            if os.environ.get("NGP_BENCH_DEBUG") and name == "grid_update":
                print("[debug] step %d grid_update %.3f ms" % (trainer.global_step, ms), file=sys.stderr)
        S_acc += trainer.last["rm_samples"]
        A_acc += int(trainer.last["n_active"].item())
    trainer.events = None
    gc.enable()
    S, A = S_acc / n_steps, A_acc / n_steps
    algo = {   # bytes per launch
        "march_count(side stream)": 60.0 * R + 4.0 * S, "march_write": 32.0 * S, "hashgrid_fwd": 588.0 * S, "mlp_fwd": 210.0 * S,
        "composite_fw+loss": 28.0 * S + 52.0 * R, "composite_bw": 52.0 * S + 64.0 * R + 24.0 * A, "mlp_bwd": 300.0 * A,
        "hashgrid_bwd": 1100.0 * A, "adam": 28.0 * n_params, "grid_update": 0.0,
    }
    stages = []
    for name, ms in acc.items():
        ms /= n_steps
        stages.append({"stage": name, "ms": round(ms, 4), "GB/s": round(algo.get(name, 0.0) / (ms * 1e-3) / 1e9, 1) if ms > 0 else None})
    stages.sort(key=lambda d: -d["ms"])
    main = [d for d in stages if d["stage"] not in ("grid_update", "march_count(side stream)")]      # the march overlaps the main stream
    top = main[0]
    achieved = top["GB/s"]
    traffic, source, prof = pmc_traffic(top["stage"], S, A)
    # every algorithmic byte the step processes (all stages, the march included) over the step time of the timed windows
    step_bytes = sum(algo.get(d["stage"], 0.0) for d in stages)
    whole = None
    if ms_per_step:
        gbs = step_bytes / (ms_per_step * 1e-3) / 1e9
        whole = {"bytes_per_step": step_bytes, "ms_per_step": ms_per_step, "achieved": round(gbs, 1), "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                 "what": "sum of the stages' algorithmic bytes (SURVEY.md 8(d) per-unit figures x the units this run processed) / ms_per_step of the timed windows"}
    return {"bound": "hbm", "kernel": top["stage"], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "builder_traffic_source": source,
            "avg_ms": top["ms"], "samples_marched_per_launch": S, "samples_active_per_launch": A,
            "units_priced": "active samples (the backward runs on the samples up to each ray's early stop)" if top["stage"] in ("hashgrid_bwd", "mlp_bwd") else "marched samples",
            "whole_step": whole, "issue_bound": (prof or {}).get("issue_bound", {}).get(top["stage"]),
            "builder_profile": profile_roofline(prof, top["stage"]),
            "main_stream_stage_sum_ms": round(sum(d["ms"] for d in main), 4), "stages": stages}


def profile_roofline(prof, stage):
    """The same roofline entry from the committed rocprofv3 profile alone (profiles/r*_pmc_traffic.json): the stage's kernel
    durations summed from the kernel trace (marching stream running next to them, tracing on) at the PROFILE's operating point."""
    if not prof or stage not in prof.get("kernel_sum_ms", {}):
        return None
    ms = prof["kernel_sum_ms"][stage]
    units = prof["samples_active_per_step"] if stage in ("hashgrid_bwd", "mlp_bwd") else prof["samples_marched_per_step"]
    per_unit = {"hashgrid_bwd": 1100.0, "mlp_bwd": 300.0, "hashgrid_fwd": 588.0, "mlp_fwd": 210.0}.get(stage)
    if per_unit is None or not ms:
        return {"kernel_sum_ms": ms}
    gbs = per_unit * units / (ms * 1e-3) / 1e9
    return {"kernel_sum_ms": ms, "units": units, "achieved": round(gbs, 1), "frac": gbs / HBM_PEAK_GBS,
            "what": "algorithmic bytes at the profile's operating point / sum of the stage's kernel durations in profiles/r*_kernel_trace_summary.txt"}


def pmc_traffic(stage, S, A):
    """HBM bytes per launch of `stage` from the newest PMC profile under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate passes over `bench.py --timed-only`, summarised by tools/pmc_traffic.py).  Not measurable from inside this process;
    only reported when that profile's operating point is close to this run's: marched samples per step within 10 %, active
    samples within 20 % (the active count of step ~540 was seen between 145 k and 169 k from run to run: the first steps of the
    occupancy warm-up add f16 atomics in arrival order); the profile's own operating point is quoted next to the number."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, "no profiles/r*_pmc_traffic.json", None
    with open(files[-1]) as f:
        prof = json.load(f)
    rel = os.path.relpath(files[-1], ROOT)
    rec = prof.get("stages", {}).get(stage)
    if rec is None:
        return None, "%s has no entry for %s" % (rel, stage), prof
    pS, pA = prof["samples_marched_per_step"], prof["samples_active_per_step"]
    if abs(pS - S) > 0.10 * S or abs(pA - A) > 0.20 * A:
        return None, "%s was recorded at %.0f marched / %.0f active samples per step, this run has %.0f / %.0f (too far apart)" % (rel, pS, pA, S, A), prof
    return rec["hbm_bytes_per_launch"], "%s, recorded at %.0f marched / %.0f active samples per step (%s)" % (rel, pS, pA, rec.get("how", "FETCH_SIZE x2 + WRITE_SIZE")), prof


def _loss_scale_note(tr):
    """The dynamic loss scale the native step trains under (GradScaler's rule on the device on top of tiny-cuda-nn's 128, round 6), read
    after the timed windows: 'x<scale> dynamic, <skipped> steps skipped so far'; 'fixed 128' when the scaler is off."""
    if getattr(tr, "loss_scaler", None) is None:
        return "fixed 128"
    sc, clean = tr.loss_scale_state()
    return "128 x %g dynamic (GradScaler rule on the device; %d clean steps, %d skipped so far)" % (sc, clean, tr.skipped_steps()[0])


def api_path_rate(loop, n_steps=120):
    """The same step driven through the reference-shaped surface: render() -> NeRFLoss -> torch autograd -> FusedAdam
    (Trainer.step_autograd), i.e. what train.py would exercise; the loop hands render() its next batch (`next_rays`), as a
    dataloader that is one batch ahead can.  Secondary number, not `value`."""
    tr = loop.trainer
    cur = loop.draw(on_side=False)
    for _ in range(24):                # (the first calls build the render stepper's buffers; the allocator and the clocks settle)
        nxt = loop.draw(on_side=False); tr.step_autograd(cur[0], cur[1], cur[2], next_batch=nxt); cur = nxt
    torch.cuda.synchronize()
    gc.collect(); gc.disable()         # as in the timed windows: a generation-2 pass is 35 ms, i.e. 70 steps' worth
    try:
        t = time.perf_counter()
        for _ in range(n_steps):
            nxt = loop.draw(on_side=False); tr.step_autograd(cur[0], cur[1], cur[2], next_batch=nxt); cur = nxt
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n_steps
    finally:
        gc.enable()
    return {"rays_per_s": loop.rays / dt, "ms_per_step": dt * 1e3,
            "what": "render(next_rays=...) + NeRFLoss + autograd + FusedAdam: the native stepper's forward / backward halves around torch's loss"}


def api_path_plain_rate(loop, n_steps=120):
    """What an UNCHANGED training_step gets (train.py:159-185): `render(model, rays_o, rays_d)` with no `next_rays` -- the reference's
    render() has no such argument -- then NeRFLoss, autograd, FusedAdam.  The batch's march runs synchronously inside render()
    (the API hands it the rays only then, and the result shapes depend on the sample count: one host round trip per step)."""
    tr = loop.trainer
    for _ in range(24):
        cur = loop.draw(on_side=False); tr.step_autograd(cur[0], cur[1], cur[2])
    torch.cuda.synchronize()
    gc.collect(); gc.disable()
    try:
        t = time.perf_counter()
        for _ in range(n_steps):
            cur = loop.draw(on_side=False); tr.step_autograd(cur[0], cur[1], cur[2])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n_steps
    finally:
        gc.enable()
    return {"rays_per_s": loop.rays / dt, "ms_per_step": dt * 1e3,
            "what": "render(model, rays_o, rays_d) exactly as train.py:159-185 calls it (no next_rays: synchronous march) + NeRFLoss + autograd + FusedAdam"}


FULL_RUN_STEPS = 30000          # BASELINE.json configs[1]: 30 epochs x 1000 steps (opt.py:40, datasets/base.py:17-19)
FULL_RUN_TEST_POSES = 200       # the Synthetic-NeRF test split (README.md:118-121 reports mean PSNR / FPS over it)
HARD_TEST_POSES = 50            # held-out poses of the lego_hard leg (its exact volumetric ground truth costs ~0.1 s a frame, outside the timed bracket)


def frame_bytes(n_rays, samples_per_ray):
    """Algorithmic bytes of one rendered frame by SURVEY.md 8(d)'s per-unit figures: AABB 32 B/ray + test-time march (24 B of ray in,
    32 B/sample out) + hash encode 588 + MLPs 210 + composite 28 B/sample and 52 B/ray."""
    return n_rays * (32.0 + 24.0 + 52.0) + n_rays * samples_per_ray * (32.0 + 588.0 + 210.0 + 28.0)


def render_profile():
    """The committed rocprofv3 kernel trace of the frame loop on the trained field (profiles/r*_render_trace.json, written by
    tools/profile_render.py): per-kernel share of a frame, dominant kernel."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_render_trace.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        rec = json.load(f)
    rec["source"] = os.path.relpath(files[-1], ROOT)
    return rec


def full_run(base_loop, args, dev, budget_s, workload="lego", n_poses=FULL_RUN_TEST_POSES, extras=True):
    """BASELINE.json configs[1] run literally: FULL_RUN_STEPS optimisation steps of 8192 rays from the random initialisation
    (cosine schedule over 30 epochs, occupancy warm-up, every step timed: one wall-clock bracket around the whole run), then the
    reference's evaluation protocol on the TRAINED field (train.py:193-237, test.ipynb cell 2): PSNR and render time of
    `render(test_time=True)` incl. ray generation over FULL_RUN_TEST_POSES held-out poses, with both chunkings of the frame loop.
    `workload` = "lego_hard_big" (extras off): the same run and protocol on the scene that does not flatter early termination
    (finite density, thin structures, object filling the frame) -- the `render_fps_800x800_hard` leg."""
    from ngp_pl_amd import synthetic as syn
    from ngp_pl_amd.bench_support import render_eval
    t_leg = time.perf_counter()
    loop = Loop(workload, args, dev, 0, 1, None, data=base_loop.data if workload == "lego" else None)       # fresh model (seed 1337); lego: the same HBM-resident dataset
    tr = loop.trainer
    steps = int(os.environ.get("NGP_FULL_RUN_STEPS", FULL_RUN_STEPS))
    tr.steps_per_epoch = max(steps // tr.num_epochs, 1)
    log = []
    torch.cuda.synchronize()
    gc.collect(); gc.disable()
    try:
        t0 = time.perf_counter()
        done = 0
        while done < steps:
            n = min(2500, steps - done)
            loop.steps(n); done += n
            if done % 5000 == 0 or done == steps:          # (reads two scalars back: ~12 syncs in the whole run, inside the bracket)
                m = tr.metrics()
                log.append({"step": done, "elapsed_s": round(time.perf_counter() - t0, 3), "train_psnr": round(m["psnr"], 2),
                            "samples_per_ray_marched": round(m["rm_s"], 2), "samples_per_ray_composited": round(m["vr_s"], 2)})
            if time.perf_counter() - t_leg > 0.6 * budget_s:
                break
        torch.cuda.synchronize()
        train_s = time.perf_counter() - t0
    finally:
        gc.enable()
    out = {"workload": ("BASELINE configs[1] literal: %d steps x %d rays from the random initialisation, 800x800 Lego-like, scale 0.5" % (done, loop.rays))
           if workload == "lego" else "%d steps x %d rays from the random initialisation; %s" % (done, loop.rays, loop.description),
           "steps": done, "train_s": train_s, "rays_per_s": done * loop.rays / train_s, "ms_per_step_mean": train_s / done * 1e3,
           "log": log, "complete": done == steps}
    if hasattr(tr, "skipped_steps"):        # steps the overflow guard did not apply (non-finite weight gradient: GradScaler's skip)
        out["skipped_steps"] = tr.skipped_steps()[0]
        out["loss_scale"] = _loss_scale_note(tr)
    progress("full_run: %d steps in %.2f s" % (done, train_s))
    poses = syn.hemisphere_poses(n_poses, seed=999).to(dev)       # held-out: the training set is seed 0
    # the reference's protocol AND chunking first (PSNR and FPS of the line are these); then the regrouped loop as an extra
    ref = render_eval(loop.model, loop.data, poses, psnr=True)
    if not extras:
        out["psnr"], out["psnr_min_max"] = ref.pop("psnr"), ref.pop("psnr_min_max")
        b = frame_bytes(loop.data.W * loop.data.H, ref["samples_per_ray"])
        ref["roofline_frac"] = b / (ref["ms_per_frame"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        ref["loop"] = "ngp_render_test_frame chunk_scale=1 probe_cap=0 (the reference's chunking, bit-identical to its host loop)"
        out.update(ref)
        del loop
        torch.cuda.empty_cache()
        return out
    fast = render_eval(loop.model, loop.data, poses, psnr=True, chunk_scale=2, probe_cap=64)       # (swept on the trained field: profiles/archive_r01_r04/r04_render_sweep_trained.txt)
    out["psnr"] = ref.pop("psnr")
    out["psnr_min_max"] = ref.pop("psnr_min_max")
    out["psnr_regrouped"] = fast.pop("psnr"); fast.pop("psnr_min_max")
    out["fps_200"], out["fps_200_regrouped"] = ref["fps"], fast["fps"]
    fast["loop"] = "ngp_render_test_frame chunk_scale=2 probe_cap=64 (the same composited samples per ray, regrouped: <= 1e-5 from the reference chunking)"
    ref["loop"] = "ngp_render_test_frame chunk_scale=1 probe_cap=0 (the reference's chunking, bit-identical to its host loop)"
    out["render"], out["render_reference_chunking"] = fast, ref
    try:
        out["render_reference_files"] = render_fps_reference_files(loop, poses[:REFERENCE_FILES_POSES])
    except Exception as e:                                        # noqa: BLE001 -- an extra leg: say so, keep the run
        out["render_reference_files"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    n_rays = loop.data.W * loop.data.H
    rr = {}
    for key, rec in (("regrouped", fast), ("reference_chunking", ref)):
        b = frame_bytes(n_rays, rec["samples_per_ray"])
        gbs = b / (rec["ms_per_frame"] * 1e-3) / 1e9
        rr[key] = {"bytes_per_frame": b, "ms_per_frame": rec["ms_per_frame"], "achieved": round(gbs, 1), "unit": "GB/s", "peak": HBM_PEAK_GBS,
                   "frac": gbs / HBM_PEAK_GBS}
    prof = render_profile()
    out["roofline_render"] = {"bound": "hbm", "kernel": (prof or {}).get("dominant_kernel"), **rr["reference_chunking"], "regrouped": rr["regrouped"],
                              "samples_per_ray": ref["samples_per_ray"], "builder_profile": prof,
                              "what": "algorithmic bytes of a frame (SURVEY.md 8(d) per-unit figures x the rays and samples of the frame) / mean frame time over the held-out poses"}
    del loop
    torch.cuda.empty_cache()
    return out


REFERENCE_FILES_POSES = 10


@torch.no_grad()
def render_fps_reference_files(loop, poses):
    """test.ipynb cell 2 around the reference's OWN models/rendering.py + networks.py + custom_functions.py (unmodified; oracle/ref_on_binding)
    on this package's `vren` / `tinycudann` bindings: `get_rays` + `render(model, rays_o, rays_d, test_time=True)` per held-out pose of the
    trained field, one untimed frame first.  What a user of the unchanged files gets at test time: the host loop of rendering.py:46-118
    (a `.item()`-style sync and a handful of torch kernels per iteration) over this library's kernels -- next to `render_fps_800x800`, the
    same frames through `ngp_render_test_frame`.  The last frame is compared with that renderer's."""
    from oracle import ref_on_binding as R
    from ngp_pl_amd import synthetic as syn
    from ngp_pl_amd.rendering import render
    if not R.available():
        return {"error": "the reference's models/*.py are neither at /root/reference nor staged under oracle/_ref/py"}
    mods = R.load()
    theirs = R.make_model(loop.model.scale, loop.dev)
    theirs.load_state_dict(loop.model.state_dict(), strict=False)
    theirs.eval()
    ro, rd = syn.get_rays(loop.data.directions, poses[0])
    mods.rendering.render(theirs, ro, rd, test_time=True)
    times = []
    for i in range(poses.shape[0]):
        torch.cuda.synchronize()
        t = time.perf_counter()
        ro, rd = syn.get_rays(loop.data.directions, poses[i])
        res = mods.rendering.render(theirs, ro, rd, test_time=True)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t)
    ours = render(loop.model, ro, rd, test_time=True)
    mean = sum(times) / len(times)
    out = {"fps": 1.0 / mean, "ms_per_frame": mean * 1e3, "n_frames": len(times), "source": mods.source,
           "max_abs_rgb_difference_to_ngp_render_test_frame": float((res["rgb"].float() - ours["rgb"]).abs().max()),
           "total_samples": int(res["total_samples"]), "total_samples_ngp_render_test_frame": int(ours["total_samples"]),
           "what": "get_rays + the reference's own render(test_time=True) (models/rendering.py:46-118, unmodified) on ngp_pl_amd.vren / ngp_pl_amd.tcnn"}
    del theirs
    torch.cuda.empty_cache()
    return out


def usable_cpus():
    """Cores this process may actually use: affinity mask and cgroup quota, not the machine's core count (32 OpenMP threads
    spinning on a 4-core quota turn the 20 s sample into many minutes)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(model, data, budget_s=20.0, timeout_s=110.0):
    """The CPU oracle timed on the host cores on BASELINE.json configs[0] (256 rays/batch): full
    step = AABB + march + composite fwd/bwd with the reference's own kernels compiled for the CPU
    (oracle/_ref, falls back to our C restatement) + hash grid / MLPs / SH / Adam as fp32 torch-CPU
    (the tiny-cuda-nn restatement, which is why kind = "port").  Checker code only: this is the one
    place outside tests/ and smoke() that touches oracle/.
    Runs in a child process in its OWN session, result through a file (no pipe a grandchild could hold open); the parent
    polls with a deadline and kills the child's process group when it is exceeded: this leg cannot outlive timeout_s."""
    import tempfile
    cores = min(usable_cpus(), 32)     # tiny tensors: more threads only add synchronisation cost
    enc = model.xyz_encoder
    gen = torch.Generator(device=data.device); gen.manual_seed(7)
    R, n_batches = 256, 101
    batches = [tuple(t.cpu() for t in data.sample(R, gen)) for _ in range(n_batches)]
    blob = {"density_w": enc.params.detach()[:enc.n_mlp].cpu().clone(), "table": enc.params.detach()[enc.n_mlp:].cpu().view(-1, 2).clone(),
            "rgb_w": model.rgb_net.params.detach().cpu().clone(), "bitfield": model.density_bitfield.cpu(),
            "ro": torch.stack([b[0] for b in batches]), "rd": torch.stack([b[1] for b in batches]), "gt": torch.stack([b[2] for b in batches]),
            "cores": cores, "budget_s": budget_s}
    tmp = tempfile.mkdtemp(prefix="ngp_cpu_baseline_")
    path, out_path, err_path = os.path.join(tmp, "in.pt"), os.path.join(tmp, "out.json"), os.path.join(tmp, "err.txt")
    failed = None
    try:
        torch.save(blob, path)
        env = dict(os.environ, OMP_NUM_THREADS=str(cores), MKL_NUM_THREADS=str(cores), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        t0 = time.perf_counter()
        with open(out_path, "w") as fo, open(err_path, "w") as fe:
            child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", path], stdout=fo, stderr=fe,
                                     stdin=subprocess.DEVNULL, env=env, start_new_session=True)
        while child.poll() is None and time.perf_counter() - t0 < timeout_s:
            time.sleep(0.2)
        if child.poll() is None:
            try:
                os.killpg(child.pid, signal.SIGKILL)          # the child's own process group, nothing else
            except OSError:
                pass
            try:
                child.wait(timeout=5)
            except subprocess.TimeoutExpired:
                pass
            failed = "worker did not finish %.0f s of CPU work within %.0f s on this host (%d usable cores)" % (budget_s, timeout_s, cores)
        else:
            lines = [ln for ln in open(out_path).read().splitlines() if ln.startswith("{")]
            if child.returncode == 0 and lines:
                return json.loads(lines[-1])
            failed = "worker exited with code %s: %s" % (child.returncode, open(err_path).read().strip()[-300:])
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    return {"value": None, "unit": "rays/s", "cores": cores, "kind": "port", "sample": "not measured in this run: " + failed}


def cpu_baseline_worker(path):
    """Child process of cpu_baseline(): no GPU, inputs from the file, one JSON line on stdout."""
    import numpy as np
    from oracle import tcnn_oracle as T
    from oracle.vren_oracle import Oracle, Reference
    blob = torch.load(path)
    cores, budget_s = int(blob["cores"]), float(blob["budget_s"])
    torch.set_num_threads(cores)
    vr = Reference(True) if Reference.available(True) else Oracle(True)
    field = T.Field(scale=0.5)
    field.density_w = blob["density_w"].requires_grad_(True)
    field.table = blob["table"].requires_grad_(True)
    field.rgb_w = blob["rgb_w"].requires_grad_(True)
    opt = torch.optim.Adam([field.density_w, field.table, field.rgb_w], lr=1e-2, eps=1e-15)
    bitfield = blob["bitfield"].numpy()
    R = blob["ro"].shape[1]
    c = np.zeros((1, 3), np.float32); hs = np.full((1, 3), 0.5, np.float32)
    n_done, S_tot, t0 = -1, 0, time.perf_counter()      # step -1 is an untimed warm-up (lazy inits)
    while n_done < 0 or (time.perf_counter() - t0 < budget_s and n_done < blob["ro"].shape[0] - 1):
        ro, rd, gt = blob["ro"][n_done + 1], blob["rd"][n_done + 1], blob["gt"][n_done + 1]
        _, hits_t, _ = vr.ray_aabb_intersect(ro.numpy(), rd.numpy(), c, hs, 1)
        ht = hits_t[:, 0].copy(); m = (ht[:, 0] >= 0) & (ht[:, 0] < 0.01); ht[m, 0] = 0.01
        noise = np.random.rand(R).astype(np.float32)
        rays_a, xyzs, dirs, deltas, ts, _ = vr.raymarching_train(ro.numpy(), rd.numpy(), ht, bitfield, 1, 0.5, 0.0, noise, 128, 1024)
        if ts.shape[0] > 0:                              # (an empty occupancy grid gives no samples: nothing to evaluate or to update)
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
    print(json.dumps({"value": R * n_done / dt, "unit": "rays/s", "cores": cores, "kind": "port",
                      "sample": "%d full steps of 256 rays (configs[0]), %.1f samples/ray, %.1f s; vren = %s; tcnn parts = fp32 torch-CPU" % (
                                    n_done, S_tot / max(n_done, 1) / R, dt,
                                    "reference .cu built for CPU (oracle/_ref)" if isinstance(vr, Reference) else "oracle/ngp_oracle.c")}), flush=True)


def march_guard_record():
    """Tripped termination guards of the marching kernels so far (all zero for sane rays) and, if any, the first tripping probe."""
    from ngp_pl_amd import _lib as native
    try:
        rec = {"march_guards": native.march_guard_counts()}
        if rec["march_guards"][0]:
            rec["march_guard_first_probe"] = native.march_guard_first()
        return rec
    except native.NgpError as e:
        return {"march_guards": None, "march_guards_error": str(e)[:200]}


def stage_roofline(loop, ms_per_step):
    """`kernel_roofline` of a secondary workload reduced to what identifies its dominant stage (no builder profiles: those were
    recorded on the headline workload)."""
    r = kernel_roofline(loop, ms_per_step, n_steps=10)
    keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_ms", "samples_marched_per_launch", "samples_active_per_launch",
            "units_priced", "main_stream_stage_sum_ms")
    out = {k: r[k] for k in keep}
    out["traffic"] = None
    out["stages"] = [{"stage": d["stage"], "ms": d["ms"]} for d in r["stages"][:5]]
    return out


def secondary_line(name, args, dev, late_steps=0, roofline=True):
    """A short run of one of the other recipes (single GPU, rank 0): 320 setup steps, 100 timed, the dominant stage's roofline;
    `late_steps` > 0: a second 100-step window after that many steps in total (where the field has sharpened)."""
    progress("secondary %s: building" % name)
    t_build = time.perf_counter()
    loop = Loop(name, args, dev, 0, 1, None)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    progress("secondary %s: running" % name)
    r = loop.run(setup_steps=args.setup_steps, warmup=10, steps=100, min_timed=100)
    progress("secondary %s: done" % name)
    met = r["metrics"]
    out = {"workload": loop.description, "rays_per_s": r["rays_per_s"], "ms_per_step": r["ms_per_step"], "rays_per_batch": loop.rays,
           "samples_per_ray_marched": met["rm_s"], "samples_per_ray_composited": met["vr_s"], "train_psnr": met["psnr"],
           "cascades": loop.model.cascades, "timed_steps_total": r["timed_steps_total"], "setup_steps": args.setup_steps,
           "global_step_at_end": r["global_step_at_end"], "dataset_build_s": round(t_build, 2)}
    if roofline:
        try:
            out["roofline"] = stage_roofline(loop, r["ms_per_step"])
        except Exception as e:                               # noqa: BLE001
            out["roofline"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    if late_steps > loop.trainer.global_step + 100:
        loop.steps(late_steps - loop.trainer.global_step - 100)
        dt, _ = loop.timed(100)
        m2 = loop.trainer.metrics()
        out["late"] = {"global_step_at_end": loop.trainer.global_step, "rays_per_s": loop.rays * 100 / dt, "ms_per_step": dt / 100 * 1e3,
                       "samples_per_ray_marched": m2["rm_s"], "samples_per_ray_composited": m2["vr_s"], "train_psnr": m2["psnr"]}
    out.update(march_guard_record())
    del loop
    torch.cuda.empty_cache()
    return out


def sensitivity(headline, args, dev, late_steps=3000):
    """rays/s against LIVE (composited) samples per ray: the headline scene (opaque surfaces) and the lego_hard scene at two
    densities, each at the bench operating point (step ~430) and `late_steps` steps in.  What the procedural Lego-like scene
    flatters is early termination; these points bracket a scene that does not."""
    pts = [{"scene": "lego (opaque surfaces; the headline)", "global_step": headline["global_step"], "live_samples_per_ray": headline["vr_s"],
            "marched_samples_per_ray": headline["rm_s"], "rays_per_s": headline["rays_per_s"], "ms_per_step": headline["ms_per_step"]}]
    detail = {}
    for name in ("lego_hard", "lego_hard_soft", "lego_hard_big"):
        try:
            r = secondary_line(name, args, dev, late_steps=late_steps, roofline=False)
        except Exception as e:                               # noqa: BLE001
            detail[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            continue
        detail[name] = r
        pts.append({"scene": r["workload"], "global_step": r["global_step_at_end"], "live_samples_per_ray": r["samples_per_ray_composited"],
                    "marched_samples_per_ray": r["samples_per_ray_marched"], "rays_per_s": r["rays_per_s"], "ms_per_step": r["ms_per_step"],
                    "train_psnr": r["train_psnr"]})
        if "late" in r:
            lt = r["late"]
            pts.append({"scene": r["workload"], "global_step": lt["global_step_at_end"], "live_samples_per_ray": lt["samples_per_ray_composited"],
                        "marched_samples_per_ray": lt["samples_per_ray_marched"], "rays_per_s": lt["rays_per_s"], "ms_per_step": lt["ms_per_step"],
                        "train_psnr": lt["train_psnr"]})
    pts.sort(key=lambda d: d["live_samples_per_ray"])
    return {"points": pts, "what": "train rays/s (Trainer.step, 8192 rays) against composited samples per ray; same recipe, same code, scenes of "
                                   "increasing translucency / thin structure", "detail": detail}


def api_path_reference_files_rate(loop, n_steps=60):
    """train.py:159-185 + :131 around the reference's OWN files -- models/rendering.py, models/networks.py, models/custom_functions.py,
    losses.py, unmodified (oracle/ref_on_binding.py: loaded from /root/reference or the staged copies under oracle/_ref/py) -- with
    `vren` / `tinycudann` aliased to this package's bindings and apex's FusedAdam replaced by this package's: INTEGRATION.md's
    "Option A", timed.  The reference's NGP is given the state of the bench's model (parameters, occupancy grid) so that it runs at
    the same operating point as `api_path_plain`.  Every kernel is libngp_hip.so's; the Python around them is the reference's."""
    from oracle import ref_on_binding as R
    from ngp_pl_amd.optim import FusedAdam
    if not R.available():
        return {"error": "the reference's models/*.py are neither at /root/reference nor staged under oracle/_ref/py"}
    dev = loop.dev
    theirs = R.make_model(loop.model.scale, dev)
    missing = theirs.load_state_dict(loop.model.state_dict(), strict=False)
    step = R.TrainingStep(theirs, FusedAdam, lr=loop.trainer.opt.param_groups[0]["lr"])
    step.global_step = loop.trainer.global_step
    for _ in range(20):
        cur = loop.draw(on_side=False); _, loss = step(cur[0], cur[1], cur[2])
    torch.cuda.synchronize()
    gc.collect(); gc.disable()
    try:
        t = time.perf_counter()
        S = 0
        for _ in range(n_steps):
            cur = loop.draw(on_side=False); res, loss = step(cur[0], cur[1], cur[2])
            S += int(res["rm_samples"])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n_steps
    finally:
        gc.enable()
    out = {"rays_per_s": loop.rays / dt, "ms_per_step": dt * 1e3, "samples_per_ray_marched": S / n_steps / loop.rays, "loss": float(loss),
           "source": R.load().source, "state_dict_missing": list(missing.missing_keys), "state_dict_unexpected": list(missing.unexpected_keys),
           "what": "the reference's own models/{rendering,networks,custom_functions}.py + losses.py, unmodified, on ngp_pl_amd.vren / "
                   "ngp_pl_amd.tcnn (module aliases only) + ngp_pl_amd.optim.FusedAdam; train.py:159-185 statement for statement incl. the "
                   "occupancy update every 16 steps through the reference's Python"}
    del theirs, step
    torch.cuda.empty_cache()
    return out


DP_EVAL_POSES = 40            # held-out poses of the evaluation sharded over the ranks (world > 1 or a 1-rank process group)


def dp_eval(loop, args, dev, rank, world, dist):
    """Under a process group: the evaluation the reference runs across ranks (train.py:193-237), with the exchange STILL INSTALLED:
    `NGP_DP_EVAL_STEPS` (default 2000) more data-parallel steps, then DP_EVAL_POSES held-out poses dealt round-robin over the ranks,
    per-pose PSNR and frame times all_gather'ed; plus what RCCL itself says about the communicator (ngp_comm_info)."""
    from ngp_pl_amd import synthetic as syn
    from ngp_pl_amd.bench_support import sharded_eval
    from ngp_pl_amd.rendering import render
    more = int(os.environ.get("NGP_DP_EVAL_STEPS", "2000"))
    t0 = time.perf_counter()
    loop.steps(more)
    loop.fence()
    train_s = time.perf_counter() - t0
    ex = loop.exchange
    poses = syn.hemisphere_poses(DP_EVAL_POSES, seed=999).to(dev)
    ro, rd = syn.get_rays(loop.data.directions, poses[rank % DP_EVAL_POSES])
    with torch.no_grad():
        render(loop.model, ro, rd, test_time=True)           # untimed: workspaces
    torch.cuda.synchronize()

    def render_pose(i):
        torch.cuda.synchronize()
        t = time.perf_counter()
        ro, rd = syn.get_rays(loop.data.directions, poses[i])
        with torch.no_grad():
            out = render(loop.model, ro, rd, test_time=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) * 1e3
        gt = loop.data.ground_truth(ro, rd)
        mse = float(((out["rgb"] - gt) ** 2).mean())
        import math
        return ms, -10.0 * math.log10(max(mse, 1e-12))
    rec = sharded_eval(render_pose, DP_EVAL_POSES, rank, world, dist, dev)
    rec["field_state"] = "after %d data-parallel steps of %d rays per rank (exchange installed)" % (loop.trainer.global_step, loop.rays)
    rec["extra_train_steps"] = more
    rec["extra_train_rays_per_s"] = more * loop.rays * world / train_s
    if hasattr(ex, "info"):
        rec.update(ex.info())
    return rec


LINE_LIMIT = 3072             # bytes: the driver keeps only a tail of stdout (BENCH_r05: a 22 KB line came back unparsed)
DETAIL_PATH = os.environ.get("NGP_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json"))


def _num(x, digits=5):
    """Numbers of the line at `digits` significant figures (a float's repr is 17-18 characters)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    return float("%.*g" % (digits, x)) if x == x and abs(x) != float("inf") else None


def _get(rec, *path):
    for k in path:
        if isinstance(rec, dict) and k in rec:
            rec = rec[k]
        elif isinstance(rec, list) and isinstance(k, int) and k < len(rec):
            rec = rec[k]
        else:
            return None
    return rec


def compact_line(rec):
    """The ONE line of stdout: the driver's contract fields, `roofline` and `cpu_baseline` as objects, one scalar per leg, the legs
    that failed by name -- at most LINE_LIMIT bytes.  Everything else (per-stage times, logs, every leg's full record) is the detail
    record: DETAIL_PATH next to this script, and stderr."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "dry_run", "note", "error", "exchange", "exchange_impl", "exchange_ms", "exposed_exchange_ms", "timed_windows", "timed_steps_total",
            "march_guards")
    out = {k: _num(rec[k]) if not isinstance(rec[k], str) else rec[k][:200] for k in keep if k in rec}
    cfg = rec.get("config")
    if isinstance(cfg, dict):
        out["config"] = {k: (_num(v) if not isinstance(v, str) else v[:160]) for k, v in cfg.items()}
    roof = rec.get("roofline")
    if isinstance(roof, dict):
        r = {k: _num(roof[k]) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_ms", "avg_ms_is", "error") if k in roof}
        prof = roof.get("builder_profile") or {}
        if prof.get("frac") is not None:          # the same entry from the committed rocprofv3 kernel trace (its own operating point)
            r["profile_avg_ms"], r["profile_frac"] = _num(prof.get("kernel_sum_ms")), _num(prof["frac"])
        for k in ("samples_marched_per_launch", "samples_active_per_launch"):
            if k in roof:
                r[k] = _num(roof[k], 7)
        if _get(roof, "whole_step", "frac") is not None:
            r["whole_step_frac"] = _num(roof["whole_step"]["frac"])
        out["roofline"] = r
    cpu = rec.get("cpu_baseline")
    if isinstance(cpu, dict):
        out["cpu_baseline"] = {k: (_num(v) if not isinstance(v, str) else v[:150]) for k, v in cpu.items() if k in ("value", "unit", "cores", "kind", "sample", "error")}
    scalars = {
        "render_fps_800x800": ("render_fps_800x800", "fps"), "render_fps_800x800_regrouped": ("render_fps_800x800_regrouped", "fps"),
        "render_fps_800x800_hard": ("render_fps_800x800_hard", "fps"), "render_fps_800x800_reference_files": ("render_fps_800x800_reference_files", "fps"),
        "render_samples_per_ray": ("render_fps_800x800", "samples_per_ray"), "render_hard_samples_per_ray": ("render_fps_800x800_hard", "samples_per_ray"),
        "render_roofline_frac": ("full_run", "roofline_render", "frac"),
        "psnr": ("full_run", "psnr"), "psnr_hard": ("render_fps_800x800_hard", "psnr"), "full_run_steps": ("full_run", "steps"), "full_run_train_s": ("full_run", "train_s"),
        "api_path_rays_per_s": ("api_path", "rays_per_s"), "api_path_plain_rays_per_s": ("api_path_plain", "rays_per_s"),
        "api_path_reference_files_rays_per_s": ("api_path_reference_files", "rays_per_s"),
        "configs3_unbounded_rays_per_s": ("secondary", 0, "rays_per_s"), "configs2_16k_rays_per_s": ("secondary", 1, "rays_per_s"),
        "cold_start_rays_per_s": ("cold_start", "rays_per_s"),
        "dp_eval_psnr": ("dp_eval", "psnr"), "dp_eval_fps_aggregate": ("dp_eval", "render_fps_aggregate"), "rccl_ranks": ("dp_eval", "rccl_ranks"),
        "exchange_mode": ("dp_eval", "exchange_mode"),
    }
    for name, path in scalars.items():
        v = _get(rec, *path)
        if v is not None:
            out[name] = _num(v)
    pts = _get(rec, "sensitivity", "points")
    if pts:
        lo, hi = min(pts, key=lambda d: d["live_samples_per_ray"]), max(pts, key=lambda d: d["live_samples_per_ray"])
        out["sensitivity_rays_per_s"] = {"at_live_samples_per_ray": [_num(lo["live_samples_per_ray"]), _num(hi["live_samples_per_ray"])],
                                         "rays_per_s": [_num(lo["rays_per_s"]), _num(hi["rays_per_s"])]}
    modes = rec.get("exchange_modes")
    if isinstance(modes, dict):
        out["exchange_modes_ms_per_step"] = {k: _num(v.get("ms_per_step")) if isinstance(v, dict) and "error" not in v else "error" for k, v in modes.items()
                                             if isinstance(v, dict)}
    failed = sorted(k for k, v in rec.items() if isinstance(v, dict) and "error" in v)
    if failed:
        out["legs_failed"] = failed
    out["detail"] = os.path.basename(DETAIL_PATH) + " (next to bench.py) and stderr: every leg's full record"
    line = json.dumps(out, separators=(",", ":"))
    if len(line) > LINE_LIMIT:                 # never expected; the contract fields win
        for k in list(out):
            if k not in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                         "data", "config", "roofline", "cpu_baseline", "error"):
                out.pop(k)
        line = json.dumps(out, separators=(",", ":"))
    return line


def write_detail(rec):
    """The full record: DETAIL_PATH (best effort: a read-only tree must not cost the line) and stderr."""
    text = json.dumps(rec)
    try:
        with open(DETAIL_PATH, "w") as f:
            f.write(text + "\n")
    except OSError as e:
        progress("could not write %s: %s" % (DETAIL_PATH, e))
    print("[bench detail] " + text, file=sys.stderr, flush=True)
    # (the driver keeps the tail of stderr: what it should end with is the headline, not the middle of the detail record)
    progress("line: value %s %s, %s ms/step; roofline %s; cpu_baseline %s" % (
        rec.get("value"), rec.get("unit"), rec.get("ms_per_step"),
        {k: _num(v) for k, v in (rec.get("roofline") or {}).items() if k in ("kernel", "achieved", "frac", "avg_ms", "traffic")},
        {k: _num(v) for k, v in (rec.get("cpu_baseline") or {}).items() if k in ("value", "cores", "kind")}))


class OnlyTheJsonLineOnStdout:
    """Everything a library prints to file descriptor 1 while the bench runs (RCCL writes a five-line version banner there when the
    first communicator is created) goes to stderr; `emit` writes the one JSON line to the real stdout."""

    def __init__(self):
        sys.stdout.flush()
        self.real = os.dup(1)
        os.dup2(2, 1)

    def emit(self, line):
        os.write(self.real, (line + "\n").encode())        # straight to the descriptor: no Python-level buffer or lock in the way


def progress(what):
    """Leg-by-leg progress on stderr, always (a run that is killed from outside then says where it was)."""
    print("[bench %8.2f s] %s" % (time.perf_counter() - _T0, what), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


class LineKeeper:
    """Owns the one JSON line.  `headline(record)` hands over the record the moment the timed windows are done;
    `leg(name, fn, budget_s)` runs one optional leg on the calling thread under a wall-clock budget; `finish()` prints.
    A daemon thread watches the running leg's budget and the global deadline: on expiry it writes the stacks of all threads
    to stderr, records the timeout in the line, prints the line (once) and ends the process with os._exit -- a hung kernel or
    a host spin cannot be interrupted from Python, it can only be left behind.  Exit status 0 when the headline was complete."""

    def __init__(self, out_stream, deadline_s, rank=0, required=("roofline", "cpu_baseline")):
        self.out_stream, self.rank, self.required = out_stream, rank, required
        self.deadline = _T0 + deadline_s
        self.deadline_s = deadline_s
        self.lock = threading.Lock()
        self.record = None
        self.partial = {}                 # what to print if the headline itself never completes
        self.emitted = False
        self.current = ("startup", _T0, self.deadline)       # (leg, started, leg deadline)
        self.pending = []                 # legs announced but not run yet (named in the line if the process has to leave early)
        self._stop = threading.Event()
        self.thread = threading.Thread(target=self._watch, name="bench-watchdog", daemon=True)
        self.thread.start()

    # -- main-thread side ------------------------------------------------------------------------
    def phase(self, name, budget_s):
        now = time.perf_counter()
        self.current = (name, now, min(now + budget_s, self.deadline))
        progress("%s ..." % name)

    def headline(self, record):
        with self.lock:
            self.record = record
        progress("headline complete: %.3g %s, %.4f ms/step" % (record.get("value") or float("nan"), record.get("unit"), record.get("ms_per_step") or float("nan")))

    def announce(self, names):
        self.pending = list(names)

    def remaining(self):
        return self.deadline - time.perf_counter()

    def leg(self, name, fn, budget_s, reserve_s=3.0):
        """Runs fn() and stores its result under `name`; exceptions are recorded, not raised.  Skipped (with a note) when less
        than the leg's budget is left before the global deadline."""
        if name in self.pending:
            self.pending.remove(name)
        left = self.remaining() - reserve_s
        if left < min(budget_s, 5.0):
            self.record[name] = {"error": "skipped: %.0f s left before the %.0f s deadline" % (max(left, 0.0), self.deadline_s)}
            progress("%s skipped (deadline)" % name)
            return
        self.phase(name, min(budget_s, left))
        t = time.perf_counter()
        try:
            res = fn()
        except Exception as e:                       # noqa: BLE001 -- recorded in the line, the run goes on
            res = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        with self.lock:
            if not self.emitted:
                self.record[name] = res
        progress("%s done in %.2f s" % (name, time.perf_counter() - t))

    def finish(self):
        """Prints the line (once).  The watchdog stays: what follows (process-group teardown) is still under the deadline."""
        self._emit()

    # -- watchdog side ---------------------------------------------------------------------------
    def _emit(self):
        with self.lock:
            if self.emitted:
                return
            self.emitted = True
            rec = self.record if self.record is not None else dict(self.partial)
            if self.rank == 0:
                write_detail(rec)
                self.out_stream.emit(compact_line(rec))

    def _watch(self):
        while not self._stop.wait(0.25):
            name, started, leg_deadline = self.current
            now = time.perf_counter()
            if now < leg_deadline and now < self.deadline:
                continue
            why = "timeout in %s: %.0f s (budget %.0f s, global deadline %.0f s)" % (name, now - started, leg_deadline - started, self.deadline_s)
            print("[bench %8.2f s] %s -- stacks of all threads follow; printing the line and leaving" % (now - _T0, why), file=sys.stderr, flush=True)
            try:
                faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
            except Exception:                        # noqa: BLE001
                pass
            with self.lock:
                if self.record is not None:
                    self.record[name] = {"error": why}
                    for later in self.pending:
                        if later != name:
                            self.record.setdefault(later, {"error": "not run: " + why})
                    self.record["error"] = why
                else:
                    self.partial.update({"value": None, "error": why})
            ok = self.record is not None
            self._emit()
            sys.stderr.flush()
            os._exit(0 if ok else 1)

    def stop(self):
        self._stop.set()


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-baseline-worker":
        return cpu_baseline_worker(sys.argv[2])
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(respawn_under_torchrun(args))
    out_stream = OnlyTheJsonLineOnStdout()
    rank = int(os.environ.get("RANK", 0)); local = int(os.environ.get("LOCAL_RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1))
    if args.dry_run or not torch.cuda.is_available():
        return dry_run(args, rank, world, out_stream)
    keeper = LineKeeper(out_stream, args.deadline, rank)
    keeper.partial = {"metric": "train rays/sec (800x800 Lego-like, 8192 rays/batch/GPU, full step incl. optimizer)", "value": None, "unit": "rays/s",
                      "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True}
    keeper.phase("device + process group", 90.0)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or "RANK" in os.environ:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
        world = dist.get_world_size()
    from ngp_pl_amd.bench_support import render_fps

    keeper.phase("dataset + model (%s)" % args.workload, 90.0)
    loop = Loop(args.workload, args, dev, rank, world, dist)
    keeper.phase("setup %d steps + warm-up %d + timed windows of %d" % (args.setup_steps, args.warmup, args.steps), 120.0)
    r = loop.run(args.setup_steps, args.warmup, args.steps)
    met = r["metrics"]
    out = {
        "metric": "train rays/sec (800x800 Lego-like, 8192 rays/batch/GPU, full step incl. optimizer)",
        "value": r["rays_per_s"], "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16/f32", "dtype_detail": "hash tables, features, MLP operands f16 with f32 MFMA/blend accumulation; march, composite, Adam f32",
        "data": "synthetic: procedural Lego-like scene, %d x %dx%d images in HBM; tcnn random init (seed 1337) TRAINED inside this run "
                "(%d setup + %d warm-up steps, untimed)" % (args.images, args.res, args.res, args.setup_steps, args.warmup),
        "config": {"workload": loop.description + "; steady state",
                   "rays_per_gpu": loop.rays, "image_res": args.res, "n_images": args.images, "setup_steps_untimed": args.setup_steps,
                   "samples_per_ray_marched": met["rm_s"], "samples_per_ray_composited": met["vr_s"], "train_psnr": met["psnr"],
                   "parallelism": "dp%d (per-ray data parallel, one native-gradient exchange per step)" % world,
                   "loss_scale": _loss_scale_note(loop.trainer)},
        "value_is": "Trainer.step (native step: direct C-ABI calls, no autograd graph); api_path = the same step through render()+autograd",
        "parity_note": "every PSNR in this line is on a procedural scene (no dataset exists on the box) and every one rests on the hash-grid / MLP / SH "
                       "arithmetic, which is held to OUR fp32 restatement of tiny-cuda-nn only (parity unpinned: tiny-cuda-nn is not in the reference tree); "
                       "marching and compositing are pinned to the reference's own kernels",
        "timed_windows": r["timed_windows"], "timed_steps_total": r["timed_steps_total"], "window_ms_per_step_min_max": r["window_ms_per_step_min_max"],
        "ms_per_step_hip_events": r["ms_per_step_hip_events"], "cold_start": r["cold_start"],
    }
    for k in ("exchange", "exchange_impl", "exchange_ms", "exposed_exchange_ms"):
        if k in r:
            out[k] = r[k]
    if "host_ms_per_step" in r:
        out["host_ms_per_step"] = r["host_ms_per_step"]
    out.update(march_guard_record())
    keeper.headline(out)
    exchange = loop.exchange
    if dist is not None and not args.timed_only:
        # collective legs, every rank, right behind the timed windows and with the exchange still installed: (1) more data-parallel
        # steps, then the evaluation sharded over the ranks; (2) the other exchange modes.  The headline is already with the
        # watchdog: a leg that hangs (a mode that has never run on real links) costs its budget, not the line.
        primary = {k: r[k] for k in ("exchange_ms", "exposed_exchange_ms") if k in r}
        want_modes = hasattr(loop.exchange, "switch_mode") and (world > 1 or os.environ.get("NGP_BENCH_BOTH_MODES") == "1")
        coll = ([("dp_eval", lambda: dp_eval(loop, args, dev, rank, world, dist), 60.0)] if not args.no_dp_eval else []) + \
               ([("exchange_modes", lambda: loop.other_exchange_modes(primary), 45.0)] if want_modes else [])
        if rank == 0:
            keeper.announce([name for name, _, _ in coll])
        for name, fn, budget in coll:
            if rank == 0:
                keeper.leg(name, fn, budget)
            else:
                keeper.phase(name, budget)
                try:
                    fn()
                except Exception as e:                      # noqa: BLE001 -- rank 0 reports; this rank must still reach the teardown
                    progress("%s failed on rank %d: %s: %s" % (name, rank, type(e).__name__, str(e)[:200]))
    if dist is not None:      # what follows runs on rank 0 only: no collectives from here on
        loop.exchange.uninstall(loop.trainer)
    if rank == 0 and not args.timed_only:
        legs = ["roofline"]
        if not args.no_cpu_baseline and world == 1:
            legs.append("cpu_baseline")
        do_full = not args.no_full_run and world == 1 and args.workload == "lego"
        if do_full:
            legs.append("full_run")
        if not args.no_render:
            legs += ["render_fps_800x800", "render_fps_800x800_regrouped"]
        do_hard = do_full and not args.no_render and not args.no_secondary
        if do_hard:
            legs.append("render_fps_800x800_hard")
        if not args.no_api:
            legs += ["api_path", "api_path_plain", "api_path_reference_files"]
        secondary = not args.no_secondary and world == 1 and args.workload == "lego"
        if secondary:
            legs += ["secondary", "sensitivity"]
        keeper.announce(legs)
        keeper.leg("roofline", lambda: kernel_roofline(loop, r["ms_per_step"]), 30.0)
        if "cpu_baseline" in legs:
            # (the driver's contract asks for this object: second in line, in a child process the parent can kill)
            keeper.leg("cpu_baseline", lambda: cpu_baseline(loop.model, loop.data), 125.0)
        if do_full:
            # the literal configs[1] run + the reference's evaluation protocol on the trained field; when it completes, the FPS
            # legs below are taken from it (200 held-out poses on the trained field) instead of 5 frames on the young one
            keeper.leg("full_run", lambda: full_run(loop, args, dev, 60.0), 60.0)
        fr = keeper.record.get("full_run") if do_full else None
        # `render_fps_800x800` = the REFERENCE'S protocol and chunking (test.ipynb cell 2: bit-identical to its host loop);
        # `render_fps_800x800_regrouped` = the same composited samples per ray regrouped into fewer iterations (an extra)
        if isinstance(fr, dict) and fr.get("complete") and "render" in fr and not args.no_render:
            for name, key in (("render_fps_800x800", "render_reference_chunking"), ("render_fps_800x800_regrouped", "render")):
                if name in keeper.pending:
                    keeper.pending.remove(name)
                keeper.record[name] = dict(fr[key], field_state="trained: after %d steps of %d rays (full_run)" % (fr["steps"], loop.rays))
            if "render_reference_files" in fr:
                keeper.record["render_fps_800x800_reference_files"] = fr["render_reference_files"]
        elif not args.no_render:
            # device-driven frame loop; chunk_scale/probe_cap only regroup the SAME per-ray samples into fewer
            # iterations (tests/test_train_gpu.py::test_device_frame_loop_matches_host_loop)
            state = "after %d training steps of %d rays" % (loop.trainer.global_step, loop.rays)

            def fast_frames():
                fast = render_fps(loop.model, loop.data, n_frames=5, chunk_scale=4, probe_cap=64)
                fast["loop"] = "ngp_render_test_frame chunk_scale=4 probe_cap=64 (the same composited samples per ray, regrouped: <= 1e-5 from the reference chunking)"
                fast["field_state"] = state
                return fast

            def reference_frames():
                ref = render_fps(loop.model, loop.data, n_frames=3)
                ref["loop"] = "ngp_render_test_frame chunk_scale=1 probe_cap=0 (the reference's chunking, bit-identical to its host loop)"
                ref["field_state"] = state
                return ref
            keeper.leg("render_fps_800x800", reference_frames, 30.0)
            keeper.leg("render_fps_800x800_regrouped", fast_frames, 30.0)
        if do_hard:
            # the FPS figure on the scene that does not flatter early termination, beside the opaque one
            keeper.leg("render_fps_800x800_hard", lambda: full_run(loop, args, dev, 45.0, workload="lego_hard_big", n_poses=HARD_TEST_POSES, extras=False), 45.0)
        if not args.no_api:
            keeper.leg("api_path", lambda: api_path_rate(loop), 30.0)
            keeper.leg("api_path_plain", lambda: api_path_plain_rate(loop), 30.0)
            keeper.leg("api_path_reference_files", lambda: api_path_reference_files_rate(loop), 40.0)
        if secondary:
            head_pt = {"global_step": r.get("global_step_at_end"), "vr_s": met["vr_s"], "rm_s": met["rm_s"], "rays_per_s": r["rays_per_s"],
                       "ms_per_step": r["ms_per_step"]}
            del loop
            torch.cuda.empty_cache()

            def both():
                res = []
                for name in ("unbounded", "lego16k"):
                    try:
                        res.append(secondary_line(name, args, dev))
                    except Exception as e:                   # noqa: BLE001
                        res.append({"workload": name, "error": "%s: %s" % (type(e).__name__, str(e)[:300])})
                return res
            keeper.leg("secondary", both, 60.0)
            keeper.leg("sensitivity", lambda: sensitivity(head_pt, args, dev), 60.0)
    if rank == 0:
        keeper.finish()
    keeper.phase("leaving" if rank == 0 else "waiting for rank 0's legs", 30.0 if rank == 0 else max(keeper.remaining(), 1.0))
    if dist is not None:
        if hasattr(exchange, "close"):
            exchange.close()                   # the library's own RCCL communicator
        dist.barrier()
        dist.destroy_process_group()
    keeper.finish()
    keeper.stop()


if __name__ == "__main__":
    main()
