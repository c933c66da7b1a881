"""Round 6: the lego_hard_big run of bench.py's `render_fps_800x800_hard` leg ended in NaN between steps 25 000 and 30 000
(gpurun_out/r06_c01_bench_detail_1.json).  Where, and what goes non-finite first?
  python tools/hard_divergence.py [workload] [steps] [check_every]
Trains the workload with bench.py's Loop; every `check_every` steps reads the metrics and the parameters' ranges; keeps a snapshot
(parameters, optimizer state, occupancy grid, draw counter) of the last healthy check and, at the first unhealthy one, replays from
the snapshot one step at a time with a census of every intermediate buffer."""
import importlib.util
import math
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)

workload = sys.argv[1] if len(sys.argv) > 1 else "lego_hard_big"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
every = int(sys.argv[3]) if len(sys.argv) > 3 else 500
args = types.SimpleNamespace(rays=0, res=800, images=100)
dev = torch.device("cuda", 0)
loop = bench.Loop(workload, args, dev, 0, 1, None)
tr, m = loop.trainer, loop.model
tr.steps_per_epoch = max(steps // tr.num_epochs, 1)
enc, net = m.xyz_encoder, m.rgb_net


def census(tag):
    p, q = enc.params.detach(), net.params.detach()
    tab = p[enc.n_mlp:]
    st = tr.opt.state
    rec = {"table_max": float(tab.abs().max()), "density_w_max": float(p[:enc.n_mlp].abs().max()), "rgb_w_max": float(q.abs().max()),
           "params_finite": bool(torch.isfinite(p).all() and torch.isfinite(q).all()),
           "grid_max": float(m.density_grid.max()), "grid_finite": bool(torch.isfinite(m.density_grid).all())}
    print("[%s] step %d %s" % (tag, tr.global_step, {k: (round(v, 5) if isinstance(v, float) else v) for k, v in rec.items()}), flush=True)
    return rec


def snapshot():
    return {"model": {k: v.clone() for k, v in m.state_dict().items()}, "opt": tr.opt.state_dict(), "step": tr.global_step, "draws": loop.draws,
            "occ_updates": getattr(m, "_occ_updates", 0), "marches": None}


good = None
bad_at = None
done = 0
while done < steps:
    loop.steps(every); done += every
    met = tr.metrics()
    rec = census("check")
    print("    metrics", {k: round(v, 4) for k, v in met.items()}, "lr %.3g" % tr.opt.param_groups[0]["lr"], flush=True)
    print("    skipped steps so far", tr.skipped_steps(), flush=True)
    healthy = rec["params_finite"]
    if not healthy:
        bad_at = done
        break
if bad_at is None:
    print("no divergence in %d steps" % steps)
    sys.exit(0)
print("UNHEALTHY at the check of step %d" % bad_at)
# Second pass from scratch (training is bit-reproducible run to run): per-step checks from `bad_at - every` on.
del loop, tr, m
torch.cuda.empty_cache()
loop = bench.Loop(workload, args, dev, 0, 1, None)
tr, m = loop.trainer, loop.model
tr.steps_per_epoch = max(steps // tr.num_epochs, 1)
enc, net = m.xyz_encoder, m.rgb_net
loop.steps(bad_at - every)
census("replay start")
B = None
for i in range(every):
    loop.steps(1)
    B = tr._buf
    S = tr.last["rm_samples"]
    st = tr.last["stats"].tolist()
    sig = B.view("sigmas", torch.float32, S); rgbs = B.view("rgbs", torch.float32, 3 * S)
    dls = B.view("dL_dsigmas", torch.float32, S)
    g16 = m._grid_grad16(dev).float()[2 * enc.meta.offset[6]:]          # (the dense levels' gradient stays in the partial tables: never written here)
    p, q = enc.params.detach(), net.params.detach()
    row = {"step": tr.global_step, "loss": st[0], "S": S, "sigma_max": float(sig.max()) if S else 0.0, "sigma_finite": bool(torch.isfinite(sig).all()),
           "rgb_finite": bool(torch.isfinite(rgbs).all()), "dLdsig_absmax": float(dls.abs().max()) if S else 0.0, "dLdsig_finite": bool(torch.isfinite(dls).all()),
           "grid_grad_absmax": float(g16.abs().max()), "grid_grad_finite": bool(torch.isfinite(g16).all()),
           "params_finite": bool(torch.isfinite(p).all() and torch.isfinite(q).all()), "table_max": float(p[enc.n_mlp:].abs().max()),
           "w_max": float(max(p[:enc.n_mlp].abs().max(), q.abs().max()))}
    bad = not (row["sigma_finite"] and row["rgb_finite"] and row["dLdsig_finite"] and row["grid_grad_finite"] and row["params_finite"] and math.isfinite(row["loss"]))
    row["skipped"] = tr.skipped_steps()[0] if i % 50 == 0 or bad else None
    if bad or i % 50 == 0:
        print(("BAD " if bad else "ok  ") + str({k: (float("%.5g" % v) if isinstance(v, float) else v) for k, v in row.items()}), flush=True)
    if bad:
        # which buffers: the forward's h (f16), features, dh, dfeats
        for name, dt, cnt in (("h", torch.float16, 16 * S), ("feats", torch.float16, 32 * S), ("dh", torch.float16, 16 * S), ("dfeats", torch.float16, 32 * S),
                              ("ws", torch.float32, S)):
            try:
                t = B.view(name, dt, cnt).float()
                print("    %-7s finite %s absmax %.5g" % (name, bool(torch.isfinite(t).all()), float(t[torch.isfinite(t)].abs().max()) if cnt else 0.0))
            except Exception as e:           # noqa: BLE001
                print("    %-7s (%s)" % (name, e))
        n_bad = int((~torch.isfinite(p)).sum()), int((~torch.isfinite(q)).sum())
        print("    non-finite params: xyz_encoder %d, rgb_net %d" % n_bad)
        break
