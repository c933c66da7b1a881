"""Scratch: what share of the live (sample, level) pairs of a training step has a feature gradient of exactly zero?  (The table
backward's list pass leaves those out; listing every live sample -- so that the pass would not depend on the field backward -- was
measured in round 6: slice owners 65 -> 128 us.)      python tools/zero_grad_census.py [workload] [steps]"""
import importlib.util, os, sys, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
workload = sys.argv[1] if len(sys.argv) > 1 else "lego"
steps = [int(a) for a in sys.argv[2:]] or [525, 3000]
args = types.SimpleNamespace(rays=0, res=800, images=100)
loop = bench.Loop(workload, args, torch.device("cuda", 0), 0, 1, None)
tr = loop.trainer
done = 0
for target in steps:
    loop.steps(target - done); done = target
    torch.cuda.synchronize()
    B = tr._buf
    S = int(tr.last["rm_samples"]); A = int(B.n_active.item())
    df = B.prefix("dfeats", torch.float16, 32 * S).view(16, S, 2)[:, :A]            # [level][compact position][2]
    zero = (df[..., 0] == 0) & (df[..., 1] == 0)
    dsig = B.prefix("dL_dsigmas", torch.float32, S); drgb = B.prefix("dL_drgbs", torch.float32, 3 * S).view(S, 3)
    act = B.prefix("active", torch.int32, S)[:A].long()
    seed0 = (dsig[act] == 0) & (drgb[act] == 0).all(1)
    print("%s step %d: %d marched, %d live; zero feature gradient: %.1f %% of (live sample, level) pairs; per level %s; whole sample zero %.1f %%; zero seeds %.1f %%" % (
        workload, done, S, A, 100 * float(zero.float().mean()), [round(100 * float(z), 1) for z in zero.float().mean(1)],
        100 * float(zero.all(0).float().mean()), 100 * float(seed0.float().mean())))
    sub = (df.float().abs() < 6.2e-5) & (df != 0)
    print("    subnormal f16 (|g| < 2^-14) among the non-zero: %.1f %%; |dfeats| median %.3e" % (100 * float(sub.float().sum() / (df != 0).float().sum()), float(df.float().abs().median())))
