#!/bin/bash
# Round 4, call 24: why the draws made ahead slow the long run: hardware-queue aliasing?
OUT=gpurun_out/r04x; mkdir -p $OUT; rm -f $OUT/*.json
timeout 600 python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "draws_made_ahead" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
B="python bench.py --no-render --no-cpu-baseline --no-api --steps 5 --warmup 2"
export NGP_FULL_RUN_STEPS=10000
NGP_OCC_DRAW_AHEAD=0 $B > $OUT/a_one.json 2> $OUT/a_one.err
NGP_OCC_DRAW_AHEAD=1 $B > $OUT/b_ahead.json 2> $OUT/b_ahead.err
GPU_MAX_HW_QUEUES=8 NGP_OCC_DRAW_AHEAD=1 $B > $OUT/c_ahead_q8.json 2> $OUT/c_ahead_q8.err
GPU_MAX_HW_QUEUES=8 NGP_OCC_DRAW_AHEAD=0 $B > $OUT/d_one_q8.json 2> $OUT/d_one_q8.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04x/*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        fr = r.get("full_run") or {}
        print(f.split("/")[-1], "ms/step %.4f" % r["ms_per_step"], "full_run", fr.get("train_s"), [(l["step"], l["elapsed_s"]) for l in fr.get("log", [])])
    except Exception as e:
        print(f, "unreadable", e)
PY
