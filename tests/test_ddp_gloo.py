"""CPU, world_size 2 over gloo: the multi-GPU path's host logic, REAL code on CPU tensors --
`Trainer._exchange_and_update` (the tail of Trainer.step: hook ordering, zero-sample rank) driving
`ngp_pl_amd.ddp.GradientExchange` (what replaces Lightning's DDP all-reduce, train.py:268-272).

Each rank computes real gradients of its own ray batch with the fp32 CPU oracle field, packs them the way
the fused backward leaves them (packed-f16 grid gradient and per-workgroup MLP partial rows, both at loss
scale 128 / world), runs the step tail, and the unscaled gradient the optimizer receives must equal the
single-process gradient of the concatenated batch (mean over ranks), including when one rank's batch
produced no samples at all."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

N_PTS = 96          # samples per rank
N_TABLE = 4096      # only the first rows of the table are given gradients to exchange (the collective does not care which)


def _field():
    from oracle import tcnn_oracle as T
    f = T.Field(scale=0.5)
    with torch.no_grad():
        f.table.uniform_(-0.3, 0.3, generator=torch.Generator().manual_seed(3))
    for p in f.parameters():
        p.requires_grad_(True)
    return f


def _rank_loss(field, rank, empty):
    """Sum over this rank's samples / N_PTS (a rank without samples contributes 0): the mean over ranks of these
    is the mean over the concatenated batch."""
    if empty:
        return None
    g = torch.Generator().manual_seed(50 + rank)
    x = (torch.rand(N_PTS, 3, generator=g) - 0.5) * 0.98
    d = torch.randn(N_PTS, 3, generator=g)
    tgt = torch.rand(N_PTS, 3, generator=g)
    sigma, rgb, _ = field.forward(x, d)
    return (((rgb - tgt) ** 2).sum() + 1e-2 * sigma.sum()) / N_PTS


def _grads(loss, field):
    if loss is None:
        return [torch.zeros_like(p) for p in (field.density_w, field.rgb_w, field.table)]
    return list(torch.autograd.grad(loss, [field.density_w, field.rgb_w, field.table]))


class _Params:
    def __init__(self, n):
        self.params = torch.zeros(n)

    def numel(self):
        return self.params.numel()


class _Model:
    """What Trainer._exchange_and_update / GradientExchange touch of NGP."""

    def __init__(self, n_grid):
        self.xyz_encoder = _Params(0); self.xyz_encoder.n_mlp = 3072; self.xyz_encoder.n_grid = n_grid
        self.rgb_net = _Params(7168)
        self._native = None
        self._g16 = torch.zeros(n_grid, dtype=torch.float16)

    def _grid_grad16(self, dev):
        return self._g16


class _CapturingAdam:
    """Stands where optim.FusedAdam stands: receives the native record after the exchange."""

    def __init__(self, model):
        self.model, self.param_groups, self.seen = model, [{"lr": 0.0}], None

    def step(self, grad_scale=1.0, found_inf=None, stream_handle=None):
        nat = self.model._native
        s = nat["scale"] * grad_scale
        n_part = nat["n_partials"]
        self.seen = dict(grid=nat["grid16"].float() / s, density=nat["density_partials"].view(n_part, -1).sum(0) / s,
                         rgb=nat["rgb_partials"].view(n_part, -1).sum(0) / s, found_inf=found_inf, lr=self.param_groups[0]["lr"])
        self.model._native = None


RANGES = [(0, 1000), (1000, 2600), (2600, N_TABLE)]     # stand-in for the library's launch-group plan (ngp_hashgrid_bwd_binned_group_entries)


def _make_trainer(model, world):
    from ngp_pl_amd.ddp import GradientExchange
    from ngp_pl_amd.trainer import Trainer
    tr = Trainer.__new__(Trainer)            # no GPU here: only the fields the step tail reads
    tr.model, tr.opt = model, _CapturingAdam(model)
    tr.global_step, tr.steps_per_epoch, tr.base_lr, tr.num_epochs = 0, 1000, 1e-2, 30
    tr.grad_scale, tr.loss_scale = 1.0, 128.0
    tr.grad_hook = tr.mlp_grad_hook = tr.group_hook = None
    tr.bwd_groups = 1
    ex = GradientExchange(model, dist, world, ranges=RANGES).install(tr)
    assert tr.loss_scale == 128.0 / world and tr.grad_hook is not None and tr.mlp_grad_hook is not None
    assert tr.bwd_groups == 3 and tr.group_hook is not None
    return tr, ex


def _worker(rank, world, port, q, empty_rank):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    field = _field()
    n_grid = 2 * N_TABLE
    # what a single process would compute on the concatenated batch
    losses = [_rank_loss(field, r, r == empty_rank) for r in range(world)]
    want = _grads(sum(lo for lo in losses if lo is not None) / world, field)
    mine = _grads(_rank_loss(field, rank, rank == empty_rank), field)
    model = _Model(n_grid)
    tr, ex = _make_trainer(model, world)
    order = []
    mlp_hook, grid_hook = tr.mlp_grad_hook, tr.grad_hook
    tr.mlp_grad_hook = lambda: (order.append("mlp"), mlp_hook())[1]
    tr.grad_hook = lambda: (order.append("grid"), grid_hook())[1]
    if rank == empty_rank:
        # Trainer.step's S == 0 branch: the rank joins both collectives with zeros
        tr._exchange_and_update(tr.zero_native(torch.device("cpu")), None, None)
    else:
        n_part = 3 + rank                                   # ranks may have different partial counts
        split = torch.rand(n_part, 1, generator=torch.Generator().manual_seed(rank))
        split = split / split.sum(0, keepdim=True)
        ls = tr.loss_scale
        native = dict(grid16=model._g16, density_partials=(split * mine[0][None] * ls).reshape(-1).contiguous(),
                      rgb_partials=(split * mine[1][None] * ls).reshape(-1).contiguous(), n_partials=n_part, scale=ls)

        def table_backward():                               # the table backward overwrites the packed-f16 gradient, group by group
            full = (mine[2].reshape(-1)[:n_grid] * ls).half()
            if rank == 0:                                   # rank 0: piecewise, as Trainer.step's binned backward hands it over
                for g, (a, b) in enumerate(RANGES):
                    order.append("table_bwd[%d]" % g)
                    model._g16[2 * a:2 * b] = full[2 * a:2 * b]
                    tr.group_hook(g, len(RANGES), a, b)
            else:                                           # rank 1: in one piece (the one-pass fallback for oversized batches) --
                order.append("table_bwd")                   # the ranks' collective sequences must still match
                model._g16.copy_(full)
        tr._exchange_and_update(native, table_backward, None)
    seen = tr.opt.seen
    want_order = ["mlp", "grid"] if rank == empty_rank else (["mlp", "table_bwd[0]", "table_bwd[1]", "table_bwd[2]", "grid"] if rank == 0 else ["mlp", "table_bwd", "grid"])
    ok = order == want_order
    ok &= seen["found_inf"] is None and model._native is None and seen["lr"] == pytest.approx(1e-2)
    errs = {}
    for key, w in (("density", want[0]), ("rgb", want[1]), ("grid", want[2].reshape(-1)[:n_grid])):
        scale = float(w.abs().max())
        errs[key] = float((seen[key] - w).abs().max()) / scale
    ok &= errs["density"] < 1e-5 and errs["rgb"] < 1e-5          # f32 all the way
    ok &= errs["grid"] < 2e-3                                     # one f16 rounding per rank + one per ring add
    # every rank ends with the same reduced gradient (lock step without a second collective)
    gathered = [torch.zeros_like(seen["grid"]) for _ in range(world)]
    dist.all_gather(gathered, seen["grid"])
    ok &= all(torch.equal(gathered[0], g) for g in gathered)
    # an overflow in the f16 sum is seen by every rank: both ranks hold 40000 -> the sum is inf in f16
    model._g16.fill_(40000.0)
    tr._exchange_and_update(dict(grid16=model._g16, density_partials=torch.zeros(3072), rgb_partials=torch.zeros(7168),
                                 n_partials=1, scale=tr.loss_scale), None, None)
    flag = tr.opt.seen["found_inf"]
    ok &= flag is not None and int(flag[0]) != 0
    q.put((rank, bool(ok), errs))
    dist.barrier()
    dist.destroy_process_group()


def _run(empty_rank):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, empty_rank)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
    assert [(r, ok) for r, ok, _ in res] == [(0, True), (1, True)], res


def test_two_rank_exchange_equals_single_process_gradient():
    _run(empty_rank=-1)


def test_two_rank_exchange_with_a_rank_without_samples():
    _run(empty_rank=1)


def _sampler_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ngp_pl_amd.bench_support import GpuDataset
    data = GpuDataset(16, 3, "cpu", seed=0)                  # same dataset on every rank ...
    gen = torch.Generator(); gen.manual_seed(1234 + rank)    # ... per-rank independent batches (datasets/base.py:25-29)
    ro, rd, gt = data.sample(64, gen)
    gathered = [torch.zeros_like(rd) for _ in range(world)]
    dist.all_gather(gathered, rd)
    ok = not torch.equal(gathered[0], gathered[1])
    pose_sum = data.poses.sum().reshape(1).clone()
    dist.all_reduce(pose_sum)
    ok &= bool(torch.isclose(pose_sum, data.poses.sum() * world).all())
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_batches_differ_dataset_agrees():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sampler_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def _eval_worker(rank, world, port, q, n_poses):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ngp_pl_amd.bench_support import sharded_eval
    rendered = []

    def render_pose(i):                      # stand-in renderer: frame time and PSNR are functions of (pose, rank)
        rendered.append(i)
        return 2.0 + rank, 30.0 + 0.1 * i
    rec = sharded_eval(render_pose, n_poses, rank, world, dist, "cpu")
    q.put((rank, rendered, rec))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_poses", [(2, 7), (3, 5)])
def test_evaluation_is_sharded_round_robin_and_gathered(world, n_poses):
    """train.py:193-237 across ranks: pose i is rendered by rank i % world only, every rank ends with the same gathered record
    (mean PSNR over ALL poses, one FPS figure per GPU from that rank's own frames), also when the poses do not divide evenly."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_eval_worker, args=(r, world, port, q, n_poses)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
    for rank, rendered, rec in res:
        assert rendered == list(range(rank, n_poses, world))
        assert rec == res[0][2]                                                   # the same record on every rank
    rec = res[0][2]
    assert rec["ranks"] == world and rec["poses_per_rank"] == [len(range(r, n_poses, world)) for r in range(world)]
    assert abs(rec["psnr"] - (30.0 + 0.1 * (n_poses - 1) / 2)) < 1e-9
    assert rec["render_fps_per_gpu"] == pytest.approx([1e3 / (2.0 + r) for r in range(world)])
    assert rec["render_fps_aggregate"] == pytest.approx(sum(1e3 / (2.0 + r) for r in range(world)))


def test_bench_launcher_spawns_n_ranks():
    """`python bench.py --gpus 2` with no rank environment starts 2 ranks (torch.distributed.run, 127.0.0.1) and
    reports n_gpus from the process group; without a GPU it stops there (dry run)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True


# ---------------------------------------------------------------------------------------------------------------------------
# ShardedExchange: reduce-scatter -> Adam on the rank's shard -> all-gather of the f16 table
# ---------------------------------------------------------------------------------------------------------------------------
N_ENTRIES_SH = 4093                 # table entries (x 2 features): NOT divisible by world x 8, so the last shard is short


def _adam_reference(p, m, v, g, lr, b1, b2, eps, step):
    """apex FusedAdam / optim.hip adam_dense, in the kernels' operation order (f32)."""
    bc1, bc2 = 1.0 - b1 ** step, 1.0 - b2 ** step
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).add_(g * g, alpha=1 - b2)
    p.sub_(lr * ((m / bc1) / ((v / bc2).sqrt() + eps)))


class _Opt:
    def __init__(self):
        self.param_groups, self.t, self.betas, self.eps, self.weight_decay = [{"lr": 0.0}], 0, (0.9, 0.999), 1e-15, 0.0


def _sharded_worker(rank, world, port, q, n_chunks=1, kind="sharded"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from ngp_pl_amd.ddp import DirectExchange, ShardedExchange
    from ngp_pl_amd.trainer import Trainer
    n_grid, n_d, n_r = 2 * N_ENTRIES_SH, 3072, 7168
    g0 = torch.Generator().manual_seed(1)
    master0 = (torch.rand(n_d + n_grid, generator=g0) - 0.5) * 0.2           # identical on every rank (DDP's broadcast)
    rgb0 = (torch.rand(n_r, generator=g0) - 0.5) * 0.2

    class Enc:
        pass
    model = _Model(n_grid)
    model.xyz_encoder.params = master0.clone()
    model.rgb_net.params = rgb0.clone()
    state = {"enc": (torch.zeros(n_d + n_grid), torch.zeros(n_d + n_grid)), "rgb": (torch.zeros(n_r), torch.zeros(n_r))}
    rgb_half = torch.zeros(n_r, dtype=torch.float16)

    tr = Trainer.__new__(Trainer)
    tr.model, tr.opt = model, _Opt()
    tr.global_step, tr.steps_per_epoch, tr.base_lr, tr.num_epochs = 0, 1000, 1e-2, 30
    tr.grad_scale, tr.loss_scale = 1.0, 128.0
    tr.grad_hook = tr.mlp_grad_hook = tr.group_hook = tr.update_hook = None
    tr.bwd_groups = 1

    def adam_cpu(lr, step, grad_scale, nat, flag_mlp, flag_shard, stream_handle, ex=None):
        ex = exchange
        em, ev = state["enc"]
        p = model.xyz_encoder.params
        for c, (lo, hi) in enumerate(ex.pieces):            # this rank's piece of every chunk; the reduce-scatters' outputs sit back to back
            if flag_shard is None and hi > lo:
                g = ex._shard16[c * ex.piece:c * ex.piece + hi - lo].float() / grad_scale
                _adam_reference(p[n_d + lo:n_d + hi], em[n_d + lo:n_d + hi], ev[n_d + lo:n_d + hi], g, lr, 0.9, 0.999, 1e-15, step)
                ex._h_big[n_d + lo:n_d + hi] = p[n_d + lo:n_d + hi].half()
        if flag_mlp is None:
            _adam_reference(p[:n_d], em[:n_d], ev[:n_d], nat["density_partials"].view(nat["n_partials"], -1).sum(0) / grad_scale, lr, 0.9, 0.999, 1e-15, step)
            ex._h_big[:n_d] = p[:n_d].half()
            rm, rv = state["rgb"]
            _adam_reference(model.rgb_net.params, rm, rv, nat["rgb_partials"].view(nat["n_partials"], -1).sum(0) / grad_scale, lr, 0.9, 0.999, 1e-15, step)
            rgb_half.copy_(model.rgb_net.params.half())

    if kind == "direct":             # point-to-point transfers + the N slices added in rank order in f32 (the mirror of ngp_stepper_tail's mode 2)
        exchange = DirectExchange(model, dist, world, rank, adam=adam_cpu).install(tr)
    else:
        exchange = ShardedExchange(model, dist, world, rank, adam=adam_cpu, n_chunks=n_chunks).install(tr)
    ok = tr.loss_scale == 128.0 / world and tr.update_hook is not None and exchange.piece % 8 == 0
    ok &= exchange.padded >= n_grid > exchange.padded - exchange.chunk and len(exchange.pieces) == n_chunks
    if n_chunks == 1:
        ok &= exchange.shard_len * world >= n_grid > exchange.shard_len * (world - 1)
    exchange._h_big[:n_d + n_grid] = master0.half()
    # three steps of per-rank gradients; values are multiples of 1/64 below 4 in magnitude, so that f16 sums over <= 8 ranks are exact (below 32 in magnitude: f16 spacing 1/64)
    # whatever the order of the ring's adds, and the expected update can be formed from the f32 sum
    ref_p, ref_rgb = master0.clone(), rgb0.clone()
    ref_state = {"enc": (torch.zeros(n_d + n_grid), torch.zeros(n_d + n_grid)), "rgb": (torch.zeros(n_r), torch.zeros(n_r))}
    for step in range(1, 4):
        grads = []
        for r in range(world):
            gr = torch.Generator().manual_seed(1000 * step + r)
            grid = torch.randint(-255, 256, (n_grid,), generator=gr).float() / 64
            grid[torch.rand(n_grid, generator=gr) < 0.5] = 0                 # untouched entries, as in a real backward
            dens = torch.randn(n_d, generator=gr) * 1e-2
            rgbw = torch.randn(n_r, generator=gr) * 1e-2
            grads.append((grid, dens, rgbw))
        mine = grads[rank]
        ls = tr.loss_scale                                                   # 128 / world
        # the table backward writes into the seated (padded) gradient buffer; values are stored scaled: mine * ls / ls ... use ls-free units
        native = dict(grid16=model._grid_grad16(torch.device("cpu")), density_partials=(mine[1] * ls).clone(), rgb_partials=(mine[2] * ls).clone(),
                      n_partials=1, scale=ls)

        def table_backward(mine=mine):
            model._grid_grad16(torch.device("cpu")).copy_(mine[0].half())    # (already "scaled": the reference below divides the same way)
        tr._exchange_and_update(native, table_backward, None)
        # single-process reference: grid gradient = sum over ranks / (ls * world); MLP = sum of (g * ls) / (ls * world)
        total_scale = ls * world
        gsum = sum(g[0] for g in grads).half().float() / total_scale
        _adam_reference(ref_p[n_d:], ref_state["enc"][0][n_d:], ref_state["enc"][1][n_d:], gsum, 1e-2, 0.9, 0.999, 1e-15, step)
        dsum = sum((g[1] * ls) for g in grads) / total_scale
        rsum = sum((g[2] * ls) for g in grads) / total_scale
        _adam_reference(ref_p[:n_d], ref_state["enc"][0][:n_d], ref_state["enc"][1][:n_d], dsum, 1e-2, 0.9, 0.999, 1e-15, step)
        _adam_reference(ref_rgb, ref_state["rgb"][0], ref_state["rgb"][1], rsum, 1e-2, 0.9, 0.999, 1e-15, step)
    table16 = exchange._h_big[:n_d + n_grid].clone()
    # (1) every rank holds the same f16 table ...
    gathered = [torch.zeros_like(table16) for _ in range(world)]
    dist.all_gather(gathered, table16)
    ok &= all(torch.equal(gathered[0], t) for t in gathered)
    # (2) ... equal to the single-process update (grid part bit for bit: exact f16 sums; MLP part: the all-reduce sums f32 in ring order)
    ok &= torch.equal(table16[n_d:], ref_p[n_d:].half())
    ok &= float((table16[:n_d].float() - ref_p[:n_d].half().float()).abs().max()) <= 2e-3 * float(ref_p[:n_d].abs().max())
    # (3) the f32 master is whole again after gather_master(), on every rank
    exchange.gather_master()
    ok &= torch.equal(model.xyz_encoder.params[n_d:], ref_p[n_d:])
    # (4) the padding of the gathered table stayed zero, the optimizer step count advanced once per step
    ok &= bool((exchange._h_big[n_d + n_grid:] == 0).all()) and tr.opt.t == 3
    # (5) a non-finite shard skips only that shard, on every rank alike: rank 0's first entries get an overflowing sum
    before = exchange._h_big[:n_d + n_grid].clone()
    big = torch.zeros(n_grid); big[:8] = 40000.0
    native = dict(grid16=model._grid_grad16(torch.device("cpu")), density_partials=torch.zeros(n_d), rgb_partials=torch.zeros(n_r), n_partials=1, scale=tr.loss_scale)
    tr._exchange_and_update(native, lambda: model._grid_grad16(torch.device("cpu")).copy_(big.half()), None)
    after = exchange._h_big[:n_d + n_grid]
    sl = exchange.shard_len
    ok &= torch.equal(after[n_d:n_d + sl], before[n_d:n_d + sl]) if world > 1 else True        # shard 0 skipped (inf in its sum) ...
    gathered = [torch.zeros_like(after) for _ in range(world)]
    dist.all_gather(gathered, after.clone())
    ok &= all(torch.equal(gathered[0], t) for t in gathered)                                   # ... and the ranks still agree
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_chunks", [(2, 1), (3, 1), (4, 1), (8, 1), (3, 2), (8, 4)])
def test_sharded_exchange_matches_the_single_process_update(world, n_chunks):
    """n_chunks > 1 is the layout of the OVERLAPPED schedule (a chunk of the gradient leaves as soon as the table backward has
    completed it; rank r owns one piece of every chunk): every configuration must reproduce the single-process Adam update bit for
    bit -- hence the chunked schedule at world 8 is bit-identical to the serial one."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q, n_chunks)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
    assert res == [(r, True) for r in range(world)], res


@pytest.mark.parametrize("world", [2, 3, 8])
def test_direct_exchange_matches_the_single_process_update(world):
    """`ddp.DirectExchange` -- every rank's slices sent straight to their owners (isend / irecv pairs: an RCCL group of sends and
    receives over all xGMI links on the device), the N slices added in rank order in f32, the updated share sent to every peer --
    under the same harness as the ring-shaped ShardedExchange: every rank ends with the same f16 table, equal to the single-process
    Adam update bit for bit, f32 master whole after gather_master(), a non-finite share skipped on every rank alike."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q, 1, "direct")) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
    assert res == [(r, True) for r in range(world)], res


def test_f16_ring_accumulation_at_world_8_is_bounded_against_the_f32_sum():
    """The default exchange (`sharded`) reduce-scatters the packed-f16 table gradient through RCCL's ring: a chunk travels rank to
    rank and every hop adds one rank's addend IN f16 (world - 1 roundings on top of each addend's own), where the reference's DDP
    all-reduces f32 gradients (train.py:268-272: one rounding).  Simulated here hop by hop for world 8 on gradients with the
    spread of real table gradients (|g| log-uniform over 2^-20 .. 2^-3 at loss scale 128 / world, random signs, a third of the
    entries touched by one rank only): the ring result stays within (world - 1) half-ulps of the running sums of the exact (f32)
    sum rounded once -- the a-priori bound; relative to the sum of the addends' magnitudes that is ~1e-4 on average and below
    (world - 1) 2^-11 = 3.4e-3 always (where the addends cancel, the error is large against the RESULT, as with any finite
    accumulator: the f32 sum has the same property 2^13 lower).  `direct` (f32,
    rank order, one rounding: ddp.DirectExchange / ngp_sum_slices_f16) has no such term; it stays the second mode until it has run
    on real links."""
    world, n = 8, 1 << 18
    g = np.random.RandomState(5)
    mag = np.exp2(g.uniform(-20, -3, size=(world, n)))
    x = (mag * g.choice([-1.0, 1.0], size=(world, n))).astype(np.float16)          # every rank's addend: one f16 rounding of its own
    lonely = g.rand(n) < 0.33
    owner = g.randint(0, world, n)
    x[:, lonely] = np.where(np.arange(world)[:, None] == owner[None, lonely], x[:, lonely], np.float16(0))
    exact = x.astype(np.float64).sum(0)
    once = exact.astype(np.float32).astype(np.float16).astype(np.float64)          # DDP's f32 sum, cast to the f16 the optimizer reads
    # ring reduce-scatter: the chunk that ends on rank r starts on rank r + 1 and visits r + 2, ..., r (each adds its slice in f16)
    ring = np.empty(n, np.float64)
    bound = np.zeros(n, np.float64)
    chunk = n // world
    for r in range(world):
        sl = slice(r * chunk, (r + 1) * chunk)
        order = [(r + 1 + k) % world for k in range(world)]
        acc = x[order[0], sl].copy()
        for q in order[1:]:
            acc = (acc.astype(np.float16) + x[q, sl]).astype(np.float16)           # numpy adds f16 + f16 in f16 (one rounding per hop)
            bound[sl] += np.spacing(np.abs(acc).astype(np.float16)).astype(np.float64) / 2
        ring[sl] = acc.astype(np.float64)
    err = np.abs(ring - exact)
    assert (err <= bound + 1e-30).all()                                             # the a-priori bound: half an ulp of every running sum
    same_sign = (np.sign(x.astype(np.float64)).min(0) >= 0) | (np.sign(x.astype(np.float64)).max(0) <= 0)
    ulp = np.maximum(np.spacing(np.abs(once).astype(np.float16)).astype(np.float64), 2.0 ** -24)
    dev = np.abs(ring - once)[same_sign] / ulp[same_sign]
    assert dev.max() <= world - 1, float(dev.max())                                 # no cancellation: at most one ulp of the result per hop
    assert (ring[lonely] == once[lonely]).all()                                     # an entry only one rank touched arrives unchanged (x + 0 is exact)
    rel = np.abs(ring - once)[~lonely] / np.abs(x.astype(np.float64)).sum(0)[~lonely]
    assert rel.mean() < 2e-4 and rel.max() < (world - 1) * 2.0 ** -11               # relative to the sum of magnitudes: ~1e-4 on average
