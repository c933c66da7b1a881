#!/bin/bash
# Round 4, call 22: placement of the next batch's march (count + scan + expansion now all on the marching stream), one run each.
OUT=gpurun_out/r04v; mkdir -p $OUT; rm -f $OUT/*.json
B="python bench.py --no-render --no-cpu-baseline --no-api --no-full-run"
for at in top hashgrid_fwd mlp_fwd composite_fw composite_bw mlp_bwd hashgrid_bwd mlp_fwd; do
  n=$(ls $OUT/*.json 2>/dev/null | wc -l)
  NGP_MARCH_AT=$at $B > $OUT/${n}_$at.json 2> $OUT/${n}_$at.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04v/*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        rf = r["roofline"]
        print(f.split("/")[-1], "ms/step %.4f" % r["ms_per_step"], "main sum", rf["main_stream_stage_sum_ms"], [(d["stage"][:12], d["ms"]) for d in rf["stages"]])
    except Exception as e:
        print(f, "unreadable", e)
PY
