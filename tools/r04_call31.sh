#!/bin/bash
# Round 4, call 31: slice owners take their first task without the queue round trip: parity + A/B.
OUT=gpurun_out/r04ae; mkdir -p $OUT; rm -f $OUT/*.json
timeout 600 python -m pytest tests/test_field_gpu.py -x -q -m gpu -k "binned or hashgrid_backward or adam" > $OUT/pytest.txt 2>&1
tail -2 $OUT/pytest.txt
timeout 600 python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "off_the_main_stream or reproducible" > $OUT/pytest2.txt 2>&1
tail -2 $OUT/pytest2.txt
V=$PWD/ngp_pl_amd/csrc/variants/libngp_hip_queue_old.so
B="python bench.py --no-render --no-cpu-baseline --no-api --no-full-run"
for i in 1 2; do
  NGP_HIP_LIB=$V $B > $OUT/old_$i.json 2> $OUT/old_$i.err
  $B > $OUT/new_$i.json 2> $OUT/new_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04ae/*.json")):
    r = json.loads(open(f).read().strip().splitlines()[-1])
    st = dict((d["stage"], d["ms"]) for d in r["roofline"]["stages"])
    print(f.split("/")[-1], "ms/step %.4f" % r["ms_per_step"], "hashgrid_bwd", st.get("hashgrid_bwd"))
PY
