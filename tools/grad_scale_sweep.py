"""Scratch (round 6): the native step runs tiny-cuda-nn's fixed loss scale 128 alone; the reference adds torch's GradScaler on top
(Lightning precision=16, train.py:274), i.e. 128 x a dynamic power of two.  At 128 alone half of the live (sample, level) feature
gradients flush to zero in f16 and nearly all others are subnormal (tools/zero_grad_census.py).  What does a larger static scale
(Trainer(grad_scale=...), the hook a GradScaler's scale goes through) do to quality, overflow and speed?
    python tools/grad_scale_sweep.py [workload] [steps] [scale ...]"""
import importlib.util, os, sys, time, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from ngp_pl_amd import synthetic as syn
from ngp_pl_amd.bench_support import render_eval
workload = sys.argv[1] if len(sys.argv) > 1 else "lego"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
scales = [a if a == "dyn" else float(a) for a in sys.argv[3:]] or [1.0, 16.0, 256.0, 4096.0, "dyn"]       # "dyn": the device-side loss scaler (the default since round 6)
args = types.SimpleNamespace(rays=0, res=800, images=100)
dev = torch.device("cuda", 0)
data = None
for gs in scales:
    loop = bench.Loop(workload, args, dev, 0, 1, None, data=data)
    data = loop.data
    tr = loop.trainer
    if gs == "dyn":
        assert tr.loss_scaler is not None
    else:
        tr.loss_scaler = None; tr.grad_scale = gs
    tr.steps_per_epoch = max(steps // tr.num_epochs, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loop.steps(steps)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    m = tr.metrics()
    B = tr._buf
    S = int(tr.last["rm_samples"]); A = int(B.n_active.item())
    df = B.prefix("dfeats", torch.float16, 32 * S).view(16, S, 2)[:, :A]
    zero = float(((df[..., 0] == 0) & (df[..., 1] == 0)).float().mean())
    poses = syn.hemisphere_poses(40, seed=999).to(dev)
    ev = render_eval(loop.model, loop.data, poses, psnr=True)
    label = "dynamic loss scale (final %g x 128, %d clean steps)" % tr.loss_scale_state() if gs == "dyn" else "grad_scale %g (x128 = %g)" % (gs, gs * 128)
    print("%s %s: %d steps in %.2f s (%.2f M rays/s), skipped %s, train psnr %.2f, live samples/ray %.1f, zero (sample, level) gradients %.1f %%, "
          "held-out PSNR %.3f dB (min %.2f max %.2f), %.0f FPS" % (workload, label, steps, dt, steps * loop.rays / dt / 1e6, tr.skipped_steps(), m["psnr"], m["vr_s"],
                                                                  100 * zero, ev["psnr"], ev["psnr_min_max"][0], ev["psnr_min_max"][1], ev["fps"]), flush=True)
    del loop, tr, B, df
    torch.cuda.empty_cache()
