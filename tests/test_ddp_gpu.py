"""GPU: the multi-rank training step END TO END with real kernels -- two processes share the one GPU of the test box (RCCL refuses
two ranks on one device, so the collectives go through gloo, which moves CUDA tensors through the host; everything else is the
product path: Trainer.step, the native gradient buffers, ngp_pl_amd.ddp.GradientExchange, ngp_found_inf2, the fused Adam).
Checks, per step: every rank's reduced gradient is the SUM of the ranks' local gradients (packed f16 grid, f32 MLP blocks), the
parameters of the ranks stay bit-identical although their batches differ, and a rank whose batch produces no samples keeps up."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, n_groups, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import numpy as np
        from ngp_pl_amd import synthetic as syn
        from ngp_pl_amd.ddp import GradientExchange
        from ngp_pl_amd.networks import NGP
        from ngp_pl_amd.trainer import Trainer
        torch.manual_seed(100 + rank)                     # different initial parameters per rank on purpose: the broadcast must fix that
        m = NGP(scale=0.5).cuda()
        m.register_training_buffers()
        tr = Trainer(m)
        ex = GradientExchange(m, dist, world, n_groups=n_groups).install(tr)
        ex.broadcast_parameters()
        assert tr.loss_scale == 128.0 / world

        def batch(n, seed):
            g = np.random.RandomState(seed)
            W = 200
            dirs = syn.get_ray_directions(W, W, syn.intrinsics(W))
            poses = syn.hemisphere_poses(16, seed=1)
            ro, rd = syn.get_rays(dirs[torch.from_numpy(g.randint(0, W * W, n))], poses[torch.from_numpy(g.randint(0, 16, n))])
            ro, rd = ro.cuda(), rd.cuda()
            gt, _ = syn.render_ground_truth(ro, rd, n_steps=96)
            return ro, rd, gt.contiguous()

        ok, notes = True, []
        pre = {}
        reduce_grid = tr.grad_hook

        def spy():                                         # local gradient right before the grid collective
            nat = m._native
            parts = pre.pop("parts", None)
            if parts:                                      # n_groups > 1: the ranges were cloned group by group, before their piece went out
                g = nat["grid16"].clone()
                for (a, b), t in parts:
                    g[2 * a:2 * b] = t
                pre["grid"] = g
            else:
                pre["grid"] = nat["grid16"].clone()
            pre["scale"] = nat["scale"]
            return reduce_grid()
        tr.grad_hook = spy
        reduce_piece = tr.group_hook
        if reduce_piece is not None:
            # (an asynchronous piece may have been reduced IN PLACE by the time the grad hook runs: a clone taken there raced with it --
            # round 6, when the step's timing moved)
            def spy_piece(group, n_groups_, a, b):
                nat = m._native
                if nat is not None:
                    torch.cuda.synchronize()
                    pre.setdefault("parts", []).append(((a, b), nat["grid16"][2 * a:2 * b].clone()))
                return reduce_piece(group, n_groups_, a, b)
            tr.group_hook = spy_piece
        adam = tr.opt.step

        def spy_adam(grad_scale=1.0, found_inf=None, stream_handle=None):
            nat = m._native
            pre["grid_sum"] = nat["grid16"].clone(); pre["mlp_sum"] = torch.cat([nat["density_partials"], nat["rgb_partials"]]).clone()
            pre["scale_after"] = nat["scale"]
            return adam(grad_scale=grad_scale, found_inf=found_inf, stream_handle=stream_handle)
        tr.opt.step = spy_adam
        for step in range(4):
            ro, rd, gt = batch(2048, seed=1000 + 10 * step + rank)           # per-rank batches
            if step == 2 and rank == 1:
                ro = ro + 10.0; rd = rd.abs() + 0.1                          # this rank's rays all miss the box: S = 0
            out = tr.step(ro, rd, gt)
            torch.cuda.synchronize()
            if step == 2 and rank == 1:
                ok &= out["rm_samples"] == 0
            # (1) reduced grid gradient == f16 sum of the ranks' local gradients
            mine = pre["grid"].float().cpu() if out["rm_samples"] > 0 else torch.zeros(m.xyz_encoder.n_grid)
            parts = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            want = parts[0].half()
            for p in parts[1:]:
                want = (want.float() + p).half()
            got = pre["grid_sum"].cpu()
            err = (got.float() - want.float()).abs().max().item() / max(want.float().abs().max().item(), 1e-12)
            ok &= err < 2e-3 and pre["scale_after"] == 128.0 and bool(torch.isfinite(got.float()).all())
            notes.append("step %d grid err %.2e" % (step, err))
            # (2) parameters bit-identical across ranks
            for p in (m.xyz_encoder.params, m.rgb_net.params):
                mine_p = p.detach().cpu()
                all_p = [torch.zeros_like(mine_p) for _ in range(world)]
                dist.all_gather(all_p, mine_p)
                ok &= all(torch.equal(all_p[0], a) for a in all_p)
            # (3) every rank holds the same reduced MLP sums
            small = pre["mlp_sum"].cpu()
            alls = [torch.zeros_like(small) for _ in range(world)]
            dist.all_gather(alls, small)
            ok &= all(torch.equal(alls[0], a) for a in alls) and float(small.abs().sum()) > 0
        ok &= bool(torch.isfinite(m.xyz_encoder.params).all())
        q.put((rank, bool(ok), notes))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:                                                  # surface the traceback in the parent
        import traceback
        q.put((rank, False, [traceback.format_exc()]))
        raise


@pytest.mark.parametrize("n_groups", [1, 2])
def test_two_ranks_train_in_lock_step_on_one_gpu(n_groups):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_groups, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(120)
    assert [(r, ok) for r, ok, _ in res] == [(0, True), (1, True)], res


def _sharded_worker(rank, world, port, q, n_chunks=1):
    """Trains 6 steps twice from the same initialisation on the same per-rank batches: with the all-reduce exchange and with the
    sharded one (reduce-scatter -> shard Adam -> all-gather).  Both must leave every rank with the same f16 working table, and
    the f32 master (after gather_master) must agree between the two exchanges: the grid gradient sums are the same f16 ring sums
    of exact per-rank addends whichever collective forms them (2 ranks: one add), the MLP blocks see the same all-reduce."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import numpy as np
        from ngp_pl_amd import synthetic as syn
        from ngp_pl_amd.ddp import GradientExchange, ShardedExchange
        from ngp_pl_amd.networks import NGP
        from ngp_pl_amd.trainer import Trainer

        def batch(n, seed):
            g = np.random.RandomState(seed)
            W = 200
            dirs = syn.get_ray_directions(W, W, syn.intrinsics(W))
            poses = syn.hemisphere_poses(16, seed=1)
            ro, rd = syn.get_rays(dirs[torch.from_numpy(g.randint(0, W * W, n))], poses[torch.from_numpy(g.randint(0, 16, n))])
            ro, rd = ro.cuda(), rd.cuda()
            gt, _ = syn.render_ground_truth(ro, rd, n_steps=96)
            return ro, rd, gt.contiguous()
        batches = [batch(1024, seed=2000 + 10 * s + rank) for s in range(6)]
        empty = (batches[3][0] + 10.0, batches[3][1].abs() + 0.1, batches[3][2])      # rank 1's 4th batch misses the box
        results = {}
        for kind in ("allreduce", "sharded"):
            torch.manual_seed(7)
            m = NGP(scale=0.5).cuda()
            m.register_training_buffers()
            tr = Trainer(m)
            ex = (GradientExchange(m, dist, world) if kind == "allreduce" else ShardedExchange(m, dist, world, rank, n_chunks=n_chunks)).install(tr)
            ex.broadcast_parameters()
            for s in range(6):
                b = empty if (s == 3 and rank == 1) else batches[s]
                nb = batches[s + 1] if s + 1 < 6 and not (s + 1 == 3 and rank == 1) else None
                tr.step(*b, next_batch=None if nb is None else (nb[0], nb[1]))
            torch.cuda.synchronize()
            enc = m.xyz_encoder
            half = enc._half.get(enc.params).clone()
            if kind == "sharded":
                assert tr.update_hook is not None and ex.padded >= enc.n_grid and len(ex.pieces) == n_chunks
                ex.gather_master()
            results[kind] = (half.cpu(), enc.params.detach().clone().cpu(), m.rgb_net.params.detach().clone().cpu())
            ex.uninstall(tr)
            del tr, m
        ok, notes = True, []
        for kind in results:
            half = results[kind][0]
            alls = [torch.zeros_like(half) for _ in range(world)]
            dist.all_gather(alls, half)
            same = all(torch.equal(alls[0], a) for a in alls)
            ok &= same; notes.append("%s: ranks' f16 tables identical: %s" % (kind, same))
        a, b = results["allreduce"], results["sharded"]
        for name, x, y in (("f16 table", a[0], b[0]), ("f32 master", a[1], b[1]), ("rgb net", a[2], b[2])):
            same = torch.equal(x, y)
            ok &= same; notes.append("%s equal between the exchanges: %s (max diff %.3g)" % (name, same, float((x.float() - y.float()).abs().max())))
        q.put((rank, bool(ok), notes))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, False, [traceback.format_exc()]))
        raise


@pytest.mark.parametrize("n_chunks", [1, 2])
def test_sharded_exchange_trains_like_the_allreduce_exchange_on_one_gpu(n_chunks):
    """n_chunks = 2: the chunked layout of the native tail (rank r owns one piece of every chunk; `ngp_adam_step_field_pieces` at
    world 2, both ranks) with real kernels and real gradients."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q, n_chunks)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(120)
    assert [(r, ok) for r, ok, _ in res] == [(0, True), (1, True)], res


# ---------------------------------------------------------------------------------------------------------------------------
# the exchange enqueued by the library (csrc/comm.hip + ngp_stepper_tail) on a real RCCL communicator of ONE rank
# ---------------------------------------------------------------------------------------------------------------------------
def _native_worker(port, q):
    """Four runs of 20 steps from the same initialisation on the same batches (two occupancy updates, one batch that misses the box):
      A  ddp.ShardedExchange    -- torch.distributed collectives issued from Python (the host-side mirror the gloo tests drive)
      B  ddp.NativeExchange     sharded, one chunk
      C  ddp.NativeExchange     sharded, two chunks behind two launch groups of the table backward
      D  ddp.NativeExchange     allreduce (gradient all-reduce only, whole-table Adam)
      E  ddp.NativeExchange     direct (point-to-point transfers, the N slices added in rank order in f32: round 5)
      F  ddp.DirectExchange     -- its torch.distributed mirror
    At world 1 every collective is the identity, so all six must leave the SAME parameters bit for bit: same kernels for the MLP
    sums, exact fixed-point table sums whatever the launch groups, the same Adam arithmetic whether it walks a shard, pieces or the
    whole table, the same device-side bias-correction count."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        import numpy as np
        from ngp_pl_amd import synthetic as syn
        from ngp_pl_amd.ddp import DirectExchange, NativeExchange, ShardedExchange
        from ngp_pl_amd.networks import NGP
        from ngp_pl_amd.trainer import Trainer

        def batch(n, seed):
            g = np.random.RandomState(seed)
            W = 200
            dirs = syn.get_ray_directions(W, W, syn.intrinsics(W))
            poses = syn.hemisphere_poses(16, seed=1)
            ro, rd = syn.get_rays(dirs[torch.from_numpy(g.randint(0, W * W, n))], poses[torch.from_numpy(g.randint(0, 16, n))])
            ro, rd = ro.cuda().contiguous(), rd.cuda().contiguous()
            gt, _ = syn.render_ground_truth(ro, rd, n_steps=96)
            return ro, rd, gt.contiguous()
        batches = [batch(2048, seed=3000 + s) for s in range(20)]
        batches[7] = (batches[7][0] + 10.0, batches[7][1].abs() + 0.1, batches[7][2])          # no samples: joins the collectives with zeros

        def run(kind):
            torch.manual_seed(9)
            m = NGP(scale=0.5).cuda()
            m.register_training_buffers()
            # A-F at the fixed loss scale (the torch.distributed mirrors have no scaler); G: the device-side loss scaler under the library's
            # exchange, S: the same factor applied statically through grad_scale
            tr = Trainer(m, loss_scaler=kind == "G", grad_scale=65536.0 if kind == "S" else 1.0)
            if kind == "A":
                ex = ShardedExchange(m, dist, 1, 0)
            elif kind == "F":
                ex = DirectExchange(m, dist, 1, 0)
            else:
                ex = NativeExchange(m, dist, 1, 0, mode={"D": "allreduce", "E": "direct"}.get(kind, "sharded"), n_chunks=2 if kind == "C" else 1,
                                    n_groups=2 if kind == "C" else 1)
            ex.install(tr)
            ex.broadcast_parameters()
            log = []
            for s in range(20):
                nb = batches[s + 1] if s + 1 < 20 else None
                out = tr.step(*batches[s], next_batch=None if nb is None else (nb[0], nb[1]))
                log.append((out["rm_samples"], tr.last["stats"].tolist() if out["rm_samples"] > 0 else None))
            torch.cuda.synchronize()
            enc = m.xyz_encoder
            snap = dict(half=enc._half.get(enc.params).clone().cpu(), master=enc.params.detach().cpu().clone(), rgb=m.rgb_net.params.detach().cpu().clone(),
                        log=log, applied=tr.opt.applied_steps(), m=[t.cpu().clone() for t in tr.opt.moments("enc")], times=None)
            if kind == "C":                                  # one more step with the stage timing on: the exchange's device times
                tr.events = []
                tr.step(*batches[0]); ex.sample_times()
                tr.events = None
                snap["times"] = (ex.exchange_ms(), ex.exposed_ms())
            snap["scale"] = tr.loss_scale_state()
            ex.uninstall(tr)
            if hasattr(ex, "close"):
                ex.close()
            return snap
        res = {k: run(k) for k in "ABCDEFGS"}
        ok, notes = True, []
        a = res["A"]
        ok &= a["log"][7][0] == 0 and a["log"][8][0] > 0
        for k in "BCDEF":
            r = res[k]
            same_log = r["log"] == a["log"]
            same = all(torch.equal(r[key], a[key]) for key in ("half", "master", "rgb")) and all(torch.equal(x, y) for x, y in zip(r["m"], a["m"]))
            ok &= same_log and same
            notes.append("%s vs A: losses identical %s, parameters + moments identical %s, applied steps %s" % (k, same_log, same, r["applied"]))
        # the dynamic loss scale under the library's exchange (G: sharded, 1 rank; GradScaler's defaults: 65536, no growth within 20
        # steps, no overflow on these batches) is the same arithmetic as the factor applied statically (S) -- and not the fixed scale's
        g, st = res["G"], res["S"]
        same_gs = g["log"] == st["log"] and all(torch.equal(g[key], st[key]) for key in ("half", "master", "rgb")) and all(torch.equal(x, y) for x, y in zip(g["m"], st["m"]))
        ok &= same_gs and g["scale"] == (65536.0, 20) and st["scale"] == (1.0, 0) and g["applied"] == (20, 20) and not torch.equal(g["master"], a["master"])
        notes.append("G vs S (loss scaler under the exchange vs the same static factor): identical %s, scale %s / %s, applied %s / %s" % (
            same_gs, g["scale"], st["scale"], g["applied"], st["applied"]))
        t = res["C"]["times"]
        ok &= t is not None and t[0] is not None and t[0] > 0 and t[1] is not None
        notes.append("C exchange_ms %.4f exposed %.4f" % (t[0] or -1, t[1] or -1))
        ok &= res["B"]["applied"] == (20, 20)
        q.put((bool(ok), notes))
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((False, [traceback.format_exc()[-3000:]]))
        raise


def _scaler_worker(port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=0, world_size=1)
        import numpy as np
        from ngp_pl_amd import synthetic as syn
        from ngp_pl_amd.ddp import NativeExchange
        from ngp_pl_amd.networks import NGP
        from ngp_pl_amd.trainer import Trainer

        def batch(n, seed):
            g = np.random.RandomState(seed)
            W = 200
            dirs = syn.get_ray_directions(W, W, syn.intrinsics(W))
            poses = syn.hemisphere_poses(16, seed=1)
            ro, rd = syn.get_rays(dirs[torch.from_numpy(g.randint(0, W * W, n))], poses[torch.from_numpy(g.randint(0, 16, n))])
            ro, rd = ro.cuda().contiguous(), rd.cuda().contiguous()
            gt, _ = syn.render_ground_truth(ro, rd, n_steps=96)
            return ro, rd, gt.contiguous()
        batches = [batch(2048, seed=4000 + s) for s in range(24)]

        def run(exchange):
            torch.manual_seed(9)
            m = NGP(scale=0.5).cuda()
            m.register_training_buffers()
            tr = Trainer(m, loss_scaler=dict(init_scale=2.0 ** 40, growth_interval=3))
            ex = None
            if exchange:
                ex = NativeExchange(m, dist, 1, 0, mode=exchange)
                ex.install(tr); ex.broadcast_parameters()
            seq = []
            for s in range(24):
                tr.step(*batches[s])
                seq.append((tr.loss_scale_state(), tr.opt.applied_steps()[0]))
            finite = bool(torch.isfinite(m.xyz_encoder.params).all()) and bool(torch.isfinite(m.rgb_net.params).all())
            if ex is not None:
                ex.uninstall(tr); ex.close()
            return seq, finite
        plain = run(None)
        notes, ok = [], plain[1] and plain[0][0][0][0] < 2.0 ** 40
        for mode in ("sharded", "allreduce", "direct"):
            seq, finite = run(mode)
            same = seq == plain[0]
            ok &= same and finite
            notes.append("%s: scale / skip sequence equal to the plain native step %s, finite %s, final %s" % (mode, same, finite, seq[-1]))
        q.put((bool(ok), notes))
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((False, [traceback.format_exc()[-3000:]]))
        raise


def test_loss_scaler_under_the_native_exchange_follows_the_same_skip_sequence():
    """From a scale that must overflow (2^40), the library's exchange (every mode, 1 rank) and the plain native step walk the SAME
    sequence of skips, halvings and doublings: under the exchange the decision is keyed on the reduced MLP sums, into which a rank whose
    own field backward raised the overflow flag writes one inf -- so every rank would see it."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_scaler_worker, args=(port, q))
    p.start()
    ok, notes = q.get(timeout=600)
    p.join(120)
    assert ok, notes


def test_native_exchange_equals_the_torch_distributed_exchange_on_one_rank():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_native_worker, args=(port, q))
    p.start()
    ok, notes = q.get(timeout=600)
    p.join(120)
    assert ok, notes


def test_communicator_collectives_on_one_rank():
    """ngp_comm_* straight through the C ABI: create from a unique id, the four collectives at world 1 (identity / copy), destroy."""
    import ctypes as C
    from ngp_pl_amd._lib import call, ptr
    torch.cuda.set_device(0)
    buf = (C.c_ubyte * 128)()
    call("ngp_comm_unique_id", C.cast(buf, C.c_void_p))
    h = C.c_void_p()
    call("ngp_comm_create", C.cast(buf, C.c_void_p), 1, 0, C.byref(h))
    w, r, v, st = C.c_int32(), C.c_int32(), C.c_int32(), C.c_void_p()
    call("ngp_comm_info", h, C.byref(w), C.byref(r), C.byref(v), C.byref(st))
    assert (w.value, r.value) == (1, 0) and v.value >= 20000 and st.value
    x = torch.randn(4096, device="cuda"); x0 = x.clone()
    call("ngp_comm_all_reduce", h, ptr(x), x.numel(), 0, None)
    g = torch.randn(1024, device="cuda").half(); out = torch.zeros_like(g)
    call("ngp_comm_reduce_scatter", h, ptr(g), ptr(out), out.numel(), 1, None)
    t = torch.randn(2048, device="cuda").half(); t0 = t.clone()
    call("ngp_comm_all_gather", h, ptr(t), ptr(t), t.numel(), 1, None)
    # the point-to-point forms: no peers at world 1 (no-ops that must not touch the buffers) ...
    stage = torch.full((2048,), 7.0, device="cuda").half()
    call("ngp_comm_exchange_slices", h, ptr(t), ptr(stage), t.numel(), 1, None)
    call("ngp_comm_all_gather_direct", h, ptr(t), t.numel(), 1, None)
    # ... and the slice sum for three "ranks" laid out as the landing area has them: own slice at rank 1, the others from the stage
    own = (torch.randint(-64, 65, (4096,), device="cuda").float() / 16).half()
    st3 = (torch.randint(-64, 65, (3, 4096), device="cuda").float() / 16).half()
    summed = torch.zeros(4096, device="cuda").half()
    call("ngp_sum_slices_f16", ptr(own), ptr(st3), 3, 1, 4096, ptr(summed), None)
    b = torch.arange(100, device="cuda", dtype=torch.uint8); b0 = b.clone()
    call("ngp_comm_broadcast", h, ptr(b), b.numel(), 0, None)
    torch.cuda.synchronize()
    assert torch.equal(x, x0) and torch.equal(out, g) and torch.equal(t, t0) and torch.equal(b, b0)
    assert bool((stage == 7.0).all())
    assert torch.equal(summed, (st3[0].float() + own.float() + st3[2].float()).half())
    from ngp_pl_amd import _lib
    _lib.lib().ngp_comm_destroy(h)
