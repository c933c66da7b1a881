#!/bin/bash
# Round 4, call 27: Adam streamed by four waves of the slice owners underneath the next task: parity + A/B of the step.
OUT=gpurun_out/r04aa; mkdir -p $OUT; rm -f $OUT/*.json
timeout 600 python -m pytest tests/test_field_gpu.py -x -q -m gpu -k "binned or adam" > $OUT/pytest_field.txt 2>&1
tail -3 $OUT/pytest_field.txt
timeout 900 python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "off_the_main_stream" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
B="python bench.py --no-render --no-cpu-baseline --no-api --no-full-run"
for i in 1 2; do
  NGP_ADAM_IN_APPLY=0 $B > $OUT/sep_$i.json 2> $OUT/sep_$i.err
  NGP_ADAM_IN_APPLY=1 $B > $OUT/fused_$i.json 2> $OUT/fused_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04aa/*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        rf = r["roofline"]
        print(f.split("/")[-1], "ms/step %.4f" % r["ms_per_step"], "value %.4e" % r["value"], "main sum", rf["main_stream_stage_sum_ms"],
              [(d["stage"][:12], d["ms"]) for d in rf["stages"]])
    except Exception as e:
        print(f, "unreadable", e)
PY
