"""CPU, world_size 2 over gloo: the multi-GPU path's host logic -- per-rank independent ray
batches (the reference's sharding axis, datasets/base.py:25-29 under DistributedSampler) and the
native-gradient all-reduce that replaces DDP's (train.py:270-272)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _Enc:
    n_mlp = 3072


class _Net:
    params = torch.zeros(7168)


class _Model:
    xyz_encoder = _Enc()
    rgb_net = _Net()
    _native = None


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ngp_pl_amd.bench_support import GpuDataset, all_reduce_native, all_reduce_native_mlp
    torch.manual_seed(100 + rank)
    n_part = 3 + rank                                   # ranks may have different partial counts
    m = _Model()
    m._native = dict(grid16=torch.full((1000,), float(rank + 1)), density_partials=torch.ones(n_part * 3072),
                     rgb_partials=torch.full((n_part * 7168,), 2.0), n_partials=n_part, scale=128.0)
    all_reduce_native_mlp(m, dist)                      # early, asynchronous half (trainer's mlp_grad_hook)
    all_reduce_native(m, dist, world)
    nat = m._native
    assert "_mlp_work" not in nat and "_mlp_small" not in nat
    ok = bool((nat["grid16"] == 3.0).all())                                         # 1 + 2
    ok &= bool((nat["density_partials"] == 3 + 4).all()) and nat["n_partials"] == 1 and nat["density_partials"].numel() == 3072
    ok &= bool((nat["rgb_partials"] == 2.0 * 7).all()) and nat["scale"] == 256.0   # sum over ranks, mean folded into the unscale
    # per-rank batches differ, same dataset
    data = GpuDataset(16, 3, "cpu", seed=0)
    gen = torch.Generator(); gen.manual_seed(1234 + rank)
    ro, rd, gt = data.sample(64, gen)
    gathered = [torch.zeros_like(rd) for _ in range(world)]
    dist.all_gather(gathered, rd)
    ok &= not torch.equal(gathered[0], gathered[1])
    pose_sum = data.poses.sum().reshape(1).clone()
    dist.all_reduce(pose_sum)
    ok &= bool(torch.isclose(pose_sum, data.poses.sum() * world).all())
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_native_gradient_allreduce():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]
