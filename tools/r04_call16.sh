mkdir -p gpurun_out/r04p
(timeout 900 python -m pytest tests/test_train_gpu.py -q -p no:cacheprovider -k "render or frame_loop" 2>&1 | tail -6)
STEPS=30000 timeout 300 python - <<'PY' 2>&1 | tail -6
import argparse, json, os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from ngp_pl_amd import synthetic as syn
from ngp_pl_amd.bench_support import render_eval
args = argparse.Namespace(rays=0, res=800, images=100)
dev = torch.device("cuda", 0)
loop = bench.Loop("lego", args, dev, 0, 1, None)
loop.steps(30000); torch.cuda.synchronize()
poses = syn.hemisphere_poses(100, seed=999).to(dev)
for cs, cap in ((1, 0), (2, 64), (1, 0), (2, 64)):
    r = render_eval(loop.model, loop.data, poses, psnr=(cs == 2), chunk_scale=cs, probe_cap=cap)
    print(json.dumps({"chunk_scale": cs, "probe_cap": cap, "fps": round(r["fps"], 1), "ms": round(r["ms_per_frame"], 4), "psnr": r.get("psnr")}), flush=True)
PY
