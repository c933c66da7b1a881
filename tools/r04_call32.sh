#!/bin/bash
# Round 4, call 32: priority of the marching stream (-1 high, the default; 0 normal), alternating runs.
OUT=gpurun_out/r04af; mkdir -p $OUT; rm -f $OUT/*.json
B="python bench.py --no-render --no-cpu-baseline --no-api --no-full-run"
for i in 1 2; do
  NGP_MARCH_PRIORITY=-1 $B > $OUT/high_$i.json 2> $OUT/high_$i.err
  NGP_MARCH_PRIORITY=0 $B > $OUT/normal_$i.json 2> $OUT/normal_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04af/*.json")):
    r = json.loads(open(f).read().strip().splitlines()[-1])
    rf = r["roofline"]
    print(f.split("/")[-1], "ms/step %.4f" % r["ms_per_step"], "main sum", rf["main_stream_stage_sum_ms"], [(d["stage"][:12], d["ms"]) for d in rf["stages"]])
PY
