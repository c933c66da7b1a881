#!/bin/bash
# Round 4, call 23: occupancy draws made ahead: parity + A/B of the step and of the 30 000-step run.
OUT=gpurun_out/r04w; mkdir -p $OUT; rm -f $OUT/*.json
timeout 900 python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "occupancy or off_the_main_stream or reproducible or erode" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
B="python bench.py --no-render --no-cpu-baseline --no-api"
for i in 1 2; do
  NGP_OCC_DRAW_AHEAD=0 $B > $OUT/one_$i.json 2> $OUT/one_$i.err
  NGP_OCC_DRAW_AHEAD=1 $B > $OUT/ahead_$i.json 2> $OUT/ahead_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04w/*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        fr = r.get("full_run") or {}
        print(f.split("/")[-1], "ms/step %.4f" % r["ms_per_step"], "value %.4e" % r["value"], "full_run", {k: fr.get(k) for k in ("train_s", "rays_per_s", "psnr")})
    except Exception as e:
        print(f, "unreadable", e)
PY
