mkdir -p gpurun_out/r04l
O=gpurun_out/r04l
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-render --no-full-run --no-api --workload lego16k"
run() { name=$1; shift; env "$@" timeout 200 $B > $O/$name.json 2> $O/$name.err; }
run default A=1
run tail0 NGP_FUSED_TAIL=0
run merge0 NGP_MERGE_IN_ADAM=0
run binned0 NGP_BINNED_BWD=0
run tworound_off NGP_TWO_ROUND=off
run native0 NGP_NATIVE_STEP=0
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04l/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-400:]); continue
    c = d["config"]
    print("%-14s value %.4g ms/step %.4f rm_s %.2f vr_s %.2f psnr %.2f cold rm_s %.1f" % (f.split("/")[-1][:-5], d["value"], d["ms_per_step"], c["samples_per_ray_marched"], c["samples_per_ray_composited"], c["train_psnr"], d["cold_start"]["samples_per_ray_marched"]))
PY
