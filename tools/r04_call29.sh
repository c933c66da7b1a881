#!/bin/bash
# Round 4, call 29: backward kernels stage their weights in one pass (one global round trip in the prologue): parity + A/B.
OUT=gpurun_out/r04ac; mkdir -p $OUT; rm -f $OUT/*.json
timeout 600 python -m pytest tests/test_field_gpu.py -x -q -m gpu -k "backward or generic or gradient" > $OUT/pytest.txt 2>&1
tail -2 $OUT/pytest.txt
V=$PWD/ngp_pl_amd/csrc/variants/libngp_hip_stage4.so
for a in 165000 25000; do
  NGP_HIP_LIB=$V python tools/bench_mlp.py 390000 $a 2>&1 | grep "^lib" | cut -c1-150
  python tools/bench_mlp.py 390000 $a 2>&1 | grep "^lib" | cut -c1-150
done
B="python bench.py --no-render --no-cpu-baseline --no-api --no-full-run"
for i in 1 2; do
  NGP_HIP_LIB=$V $B > $OUT/old_$i.json 2> $OUT/old_$i.err
  $B > $OUT/new_$i.json 2> $OUT/new_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04ac/*.json")):
    r = json.loads(open(f).read().strip().splitlines()[-1])
    st = dict((d["stage"], d["ms"]) for d in r["roofline"]["stages"])
    print(f.split("/")[-1], "ms/step %.4f" % r["ms_per_step"], "mlp_bwd", st.get("mlp_bwd"), "vr_s", r.get("vr_s"))
PY
