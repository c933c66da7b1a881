#!/bin/bash
# Round 4, call 19: two packed-sample sets (expansion on the marching stream): tests + A/B of the driver's step.
OUT=gpurun_out/r04s; mkdir -p $OUT
timeout 900 python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "off_the_main_stream or set_sample_sets or reproducible or merge_folded or native_render_node or stale_values" > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
timeout 600 python -m pytest tests/test_field_gpu.py -x -q -m gpu -k "binned" > $OUT/pytest_field.txt 2>&1
tail -3 $OUT/pytest_field.txt
B="python bench.py --no-render --no-cpu-baseline --no-api --no-full-run"
rm -f $OUT/*.json
for i in 1 2; do
  NGP_LISTS_AHEAD=0 $B > $OUT/serial_$i.json 2> $OUT/serial_$i.err
  NGP_LISTS_AHEAD=1 $B > $OUT/ahead_$i.json 2> $OUT/ahead_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04s/*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        rf = r["roofline"]
        print(f.split("/")[-1], "ms/step %.4f" % r["ms_per_step"], "value %.4e" % r["value"], "main sum", rf["main_stream_stage_sum_ms"],
              [(d["stage"], d["ms"]) for d in rf["stages"]])
    except Exception as e:
        print(f, "unreadable", e)
PY
