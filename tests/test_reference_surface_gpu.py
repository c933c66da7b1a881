"""GPU: the reference's OWN training surface around the native path -- what an unchanged train.py does with the model:
`FusedAdam(net_params, lr, eps=1e-15)` + `CosineAnnealingLR(net_opt, ...)` (train.py:123-137), `NeRFSystem.forward` ->
`render(self.model, rays_o, rays_d)` (train.py:78-91), `NeRFLoss`, `loss.backward()`, optionally under torch's GradScaler
(Lightning precision=16, train.py:274) and wrapped in `DistributedDataParallel` (Lightning's DDPPlugin, train.py:268-272)."""
import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

pytestmark = pytest.mark.gpu


def _batch(n, seed, dev="cuda"):
    from ngp_pl_amd import synthetic as syn
    g = np.random.RandomState(seed)
    W = 200
    dirs = syn.get_ray_directions(W, W, syn.intrinsics(W))
    poses = syn.hemisphere_poses(16, seed=1)
    ro, rd = syn.get_rays(dirs[torch.from_numpy(g.randint(0, W * W, n))], poses[torch.from_numpy(g.randint(0, 16, n))])
    ro, rd = ro.to(dev).contiguous(), rd.to(dev).contiguous()
    gt, _ = syn.render_ground_truth(ro, rd, n_steps=96)
    return ro, rd, gt.contiguous()


class _System(nn.Module):
    """NeRFSystem as far as the optimizer and DDP see it (train.py:60-91): owns the model, forward = render()."""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, rays_o, rays_d):
        from ngp_pl_amd.rendering import render
        return render(self.model, rays_o, rays_d, test_time=False)


def _make(seed=11):
    from ngp_pl_amd.networks import NGP
    torch.manual_seed(seed)
    m = NGP(scale=0.5).cuda()
    m.register_training_buffers()
    m.update_density_grid(0.01 * 1024 / 3 ** 0.5, warmup=True)
    return m


def _loss(results, gt):
    from ngp_pl_amd.losses import NeRFLoss
    d = NeRFLoss()(results, {"rgb": gt})
    return sum(lo.mean() for lo in d.values())            # train.py:173


def _net_params(system):
    return [p for n, p in system.named_parameters() if n not in ("dR", "dT")]           # train.py:123-127


def test_fused_adam_is_built_and_scheduled_the_way_train_py_does_it():
    """train.py:131-137 verbatim against this package's FusedAdam: a torch Optimizer built from the parameter list, driven by
    CosineAnnealingLR; with f32 `.grad` tensors (model.native_grads False, the default) one step equals torch.optim.Adam on the same
    gradients, the f16 working copies follow, and the scheduler's learning rate is the one the next step applies."""
    from torch.optim.lr_scheduler import CosineAnnealingLR
    from ngp_pl_amd.optim import FusedAdam
    model = _make()
    system = _System(model)
    net_params = _net_params(system)
    assert len(net_params) == 3 and model.native_grads is False
    opt = FusedAdam(net_params, 1e-2, eps=1e-15)
    sch = CosineAnnealingLR(opt, 30, 1e-2 / 30)
    assert isinstance(opt, torch.optim.Optimizer) and opt.model is model and model.native_grads is False
    live = [p for p in net_params if p.numel()]
    ref_p = [p.detach().clone().requires_grad_(True) for p in live]
    ref = torch.optim.Adam(ref_p, 1e-2, eps=1e-15)
    ref_sch = CosineAnnealingLR(ref, 30, 1e-2 / 30)
    for it in range(3):
        ro, rd, gt = _batch(1024, 300 + it)
        loss = _loss(system(ro, rd), gt)
        opt.zero_grad()
        loss.backward()
        assert all(p.grad is not None and p.grad.dtype == torch.float32 for p in live) and model._native is None
        for q, p in zip(ref_p, live):
            q.grad = p.grad.detach().clone()
        opt.step(); ref.step()
        sch.step(); ref_sch.step()                         # (an epoch per iteration: the schedule moves visibly)
        assert opt.param_groups[0]["lr"] == ref.param_groups[0]["lr"] < 1e-2
        for q, p in zip(ref_p, live):
            scale = float(q.detach().abs().max())
            assert float((q.detach() - p.detach()).abs().max()) <= 2e-6 * scale + 1e-9, it
    enc, net = model.xyz_encoder, model.rgb_net
    assert torch.equal(enc._half.get(enc.params), enc.params.detach().half()) and torch.equal(net._half.get(net.params), net.params.detach().half())
    sd = opt.state_dict()                                  # the torch Optimizer protocol (Lightning checkpoints call it)
    assert len(sd["state"]) == 2 and sd["param_groups"][0]["eps"] == 1e-15
    opt.load_state_dict(sd)


def test_native_gradients_and_grad_tensors_drive_the_same_update():
    """`FusedAdam(net_params, ..., native_grads=True)`: the backward leaves packed-f16 / partial-row buffers instead of `.grad` and
    one fused launch applies them.  The SAME gradients, materialised as f32 `.grad` tensors the way the autograd node does it for
    native_grads=False (partials summed, f16 table gradient cast, both divided by the loss scale), drive a second model through the
    per-tensor route: parameters, moments and f16 working copies agree to float rounding after every one of three steps."""
    from ngp_pl_amd import tcnn
    from ngp_pl_amd._lib import call, ptr, stream
    from ngp_pl_amd.optim import FusedAdam
    model_a, model_b = _make(seed=12), _make(seed=12)
    model_a.native_loss_scaler = None                      # the fixed loss scale: the native buffers are re-scaled by hand below
    sys_a, sys_b = _System(model_a), _System(model_b)
    opt_a = FusedAdam(_net_params(sys_a), 1e-2, eps=1e-15, native_grads=True)
    opt_b = FusedAdam(_net_params(sys_b), 1e-2, eps=1e-15)
    assert model_a.native_grads is True and model_b.native_grads is False
    enc_a, net_a, enc_b, net_b = model_a.xyz_encoder, model_a.rgb_net, model_b.xyz_encoder, model_b.rgb_net
    for it in range(3):
        ro, rd, gt = _batch(1024, 400 + it)
        loss = _loss(sys_a(ro, rd), gt)
        opt_a.zero_grad()
        loss.backward()
        nat = model_a._native
        assert nat is not None and enc_a.params.grad is None and net_a.params.grad is None
        scale, n_part = nat["scale"], nat["n_partials"]
        g_enc = torch.empty_like(enc_b.params)
        g_enc[:enc_b.n_mlp] = tcnn.reduce_partials(nat["density_partials"], n_part, enc_b.n_mlp) / scale
        call("ngp_cast_f16_to_f32", ptr(nat["grid16"]), enc_b.n_grid, 1.0 / scale, ptr(g_enc[enc_b.n_mlp:]), stream())
        enc_b.params.grad = g_enc
        net_b.params.grad = tcnn.reduce_partials(nat["rgb_partials"], n_part, net_b.params.numel()) / scale
        opt_a.step(); opt_b.step()
        assert model_a._native is None
        for pa, pb in ((enc_a.params, enc_b.params), (net_a.params, net_b.params)):
            tol = 2e-6 * float(pa.detach().abs().max()) + 1e-9
            assert float((pa.detach() - pb.detach()).abs().max()) <= tol, it
            for k in ("exp_avg", "exp_avg_sq"):
                ma, mb = opt_a.state[pa][k], opt_b.state[pb][k]
                assert float((ma - mb).abs().max()) <= 2e-6 * float(ma.abs().max()) + 1e-20, (it, k)
        assert float((enc_a._half.t.float() - enc_b._half.t.float()).abs().max()) <= 1e-3 * float(enc_a._half.t.float().abs().max())
    assert float((enc_a.params.detach() - _make(seed=12).xyz_encoder.params.detach()).abs().max()) > 1e-3


def test_unchanged_training_step_under_torch_gradscaler():
    """Lightning precision=16 wraps the step in torch.amp.GradScaler: scale(loss).backward(), unscale_ over the optimizer's `.grad`
    tensors, inf check, step (train.py:274).  That protocol needs f32 `.grad`s -- the default gradient route."""
    from ngp_pl_amd.optim import FusedAdam
    model = _make(seed=13)
    system = _System(model)
    opt = FusedAdam(_net_params(system), 1e-2, eps=1e-15)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
    before = [p.detach().clone() for p in _net_params(system)]
    for it in range(3):
        ro, rd, gt = _batch(1024, 500 + it)
        with torch.autocast("cuda", dtype=torch.float16):
            loss = _loss(system(ro, rd), gt)
        opt.zero_grad()
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
    after = _net_params(system)
    assert all(bool(torch.isfinite(p).all()) for p in after)
    assert all(float((a - b).abs().max()) > 1e-3 for a, b in zip(after, before) if a.numel())
    assert scaler.get_scale() == 1024.0                    # no overflow was seen


def test_optimizer_checkpoint_round_trip_continues_bit_for_bit():
    """`optimizer.state_dict()` -> `load_state_dict()` into a freshly built model + optimizer (what resuming a Lightning checkpoint
    does): the NEXT update equals the uninterrupted run's bit for bit -- parameters, f16 working copies and moments -- which needs
    the bias-correction count of the native gradient route to travel with the state (it restarted at 1 before round 5)."""
    import copy
    from ngp_pl_amd.optim import FusedAdam
    from ngp_pl_amd.rendering import render

    def one_step(model, opt, it):
        ro, rd, gt = _batch(1024, 700 + it)
        noise = torch.rand(1024, generator=torch.Generator().manual_seed(it)).cuda()       # the march's jitter, the same on both sides
        res = render(model, ro, rd, test_time=False, noise=noise)
        opt.zero_grad()
        _loss(res, gt).backward()
        assert model._native is not None                    # the native gradient route
        opt.step()

    a = _make(seed=31)
    opt_a = FusedAdam(_net_params(_System(a)), 1e-2, eps=1e-15, native_grads=True)
    for it in range(3):
        one_step(a, opt_a, it)
    assert opt_a.t == 3 and opt_a.state[a.rgb_net.params]["step"] == 3
    sd_model, sd_opt = copy.deepcopy(a.state_dict()), copy.deepcopy(opt_a.state_dict())
    b = _make(seed=77)
    b.load_state_dict(sd_model)
    opt_b = FusedAdam(_net_params(_System(b)), 1e-2, eps=1e-15, native_grads=True)
    opt_b.load_state_dict(sd_opt)
    assert opt_b.t == 3
    one_step(a, opt_a, 3); one_step(b, opt_b, 3)
    for pa, pb in ((a.xyz_encoder.params, b.xyz_encoder.params), (a.rgb_net.params, b.rgb_net.params)):
        assert torch.equal(pa.detach(), pb.detach())
        for k in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(opt_a.state[pa][k], opt_b.state[pb][k]), k
    assert torch.equal(a.xyz_encoder._half.get(a.xyz_encoder.params), b.xyz_encoder._half.get(b.xyz_encoder.params))
    assert opt_b.t == 4 == opt_b.state[b.rgb_net.params]["step"]


# -------------------------------------------------------------------------------------------------------------------------
# DistributedDataParallel around the system (train.py:268-272)
# -------------------------------------------------------------------------------------------------------------------------
def _ddp_worker(rank, world, port, backend, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        torch.cuda.set_device(0)
        dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": torch.device("cuda", 0)} if backend == "nccl" else {}))
        from torch.nn.parallel import DistributedDataParallel as DDP
        from ngp_pl_amd.optim import FusedAdam
        model = _make(seed=20 + rank)                      # different initial parameters per rank: DDP's constructor broadcast fixes that
        system = _System(model)
        ddp = DDP(system, device_ids=[0] if backend == "nccl" else None)
        opt = FusedAdam(_net_params(system), 1e-2, eps=1e-15)
        assert model.native_grads is False
        start = [p.detach().clone() for p in _net_params(system)]
        ok, notes = True, []
        for it in range(4):
            ro, rd, gt = _batch(1024, 600 + 10 * it + rank)                              # per-rank batches
            loss = _loss(ddp(ro, rd), gt)                  # forward THROUGH the wrapper: DDP arms its reducer there
            opt.zero_grad()
            loss.backward()                                # ... and its hooks all-reduce the f32 .grad tensors here
            g = model.xyz_encoder.params.grad
            ok &= g is not None and bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0
            opt.step()
            if it == 1:
                model.update_density_grid(0.01 * 1024 / 3 ** 0.5, warmup=True)
        torch.cuda.synchronize()
        params = [p.detach() for p in _net_params(system) if p.numel()]
        ok &= all(float((p - s).abs().max()) > 1e-3 for p, s in zip(params, [s for s in start if s.numel()]))
        flat = torch.cat([p.reshape(-1) for p in params]).cpu()
        if world > 1:
            parts = [torch.zeros_like(flat) for _ in range(world)]
            dist.all_gather(parts, flat)
            same = all(torch.equal(parts[0], t) for t in parts)
            ok &= same
            notes.append("ranks identical: %s" % same)
        q.put((rank, bool(ok), notes))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:                                 # noqa: BLE001
        import traceback
        q.put((rank, False, [traceback.format_exc()[-1500:]]))


def _spawn(world, backend):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(60)
    assert all(r[1] for r in res), res
    return res


def test_ngp_under_distributed_data_parallel_one_rank_rccl():
    """The reference's multi-GPU route, unchanged: the module wrapped in DistributedDataParallel over RCCL (one rank: the only
    world the test box has).  Four iterations -- a parameter that never receives a gradient would fail the SECOND one
    ("Expected to have finished reduction in the prior iteration"): the SH encoding's empty `params` does not require grad."""
    _spawn(1, "nccl")


def test_ngp_under_distributed_data_parallel_two_ranks_stay_identical():
    """Two ranks on the one GPU (gloo moves the CUDA gradients through the host; RCCL refuses two ranks per device): DDP's hooks
    fire on the f32 gradients of the fused render node, the ranks start from rank 0's parameters and stay bit-identical although
    their batches differ."""
    _spawn(2, "gloo")


def test_a_deep_copied_model_still_trains_correctly():
    """`copy.deepcopy(model)` creates new Parameter objects without the tags FusedAdam uses to find the model and its f16 working
    copies.  The optimizer then treats the tensors as plain ones (per-tensor kernel, f32 `.grad`) -- and bumps their version counters,
    so the modules re-cast their working copies at the next forward instead of going on with stale ones."""
    import copy
    from ngp_pl_amd.optim import FusedAdam
    model = copy.deepcopy(_make(seed=14))
    system = _System(model)
    opt = FusedAdam(_net_params(system), 1e-2, eps=1e-15)
    assert opt.model is None
    enc = model.xyz_encoder
    for it in range(3):
        ro, rd, gt = _batch(1024, 700 + it)
        loss = _loss(system(ro, rd), gt)
        opt.zero_grad()
        loss.backward()
        before = enc.params.detach().clone()
        opt.step()
        assert float((enc.params.detach() - before).abs().max()) > 1e-4
        assert torch.equal(enc._half.get(enc.params), enc.params.detach().half())          # what the next forward will read


def test_native_render_node_trains_under_the_dynamic_loss_scale():
    """render() + NeRFLoss + backward + FusedAdam(native_grads=True) exactly as train.py:159-185 strings them together, with the
    GradScaler of Lightning's precision=16 (train.py:274) living on the device: the native node's field backward multiplies its seeds
    by the dynamic scale, FusedAdam's one launch divides by it, skips on overflow and updates it.  From GradScaler's default 65536 on
    a fresh model: no parameter goes non-finite, the scale stays a power of two, skipped steps leave the applied count behind, and
    the loss falls."""
    from ngp_pl_amd.optim import FusedAdam
    model = _make(seed=14)
    system = _System(model)
    opt = FusedAdam(_net_params(system), 1e-2, eps=1e-15, native_grads=True)
    assert model.native_loss_scaler["init_scale"] == 65536.0
    losses = []
    for it in range(40):
        ro, rd, gt = _batch(2048, 700 + it % 4)
        loss = _loss(system(ro, rd), gt)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    rs = model._render_stepper
    scale, clean = rs.loss_scale_state()
    applied = opt.applied_steps()[0]
    assert scale > 1.0 and math.log2(scale) == int(math.log2(scale)) and 0 <= clean <= 40
    assert applied + int(round(math.log2(65536.0 / scale))) == 40 or applied == 40          # every skip halved the scale (no growth within 40 steps)
    for p in _net_params(system):
        assert bool(torch.isfinite(p).all())
    assert sum(losses[-5:]) < 0.5 * sum(losses[:5])
