#!/bin/bash
# Round 4, call 30: forward kernels stage their weights in one pass: parity + A/B (field forward alone, the step, the frame loop).
OUT=gpurun_out/r04ad; mkdir -p $OUT; rm -f $OUT/*.json
timeout 600 python -m pytest tests/test_field_gpu.py tests/test_train_gpu.py -x -q -m gpu -k "field_forward or generic or sh_encoding or variants or render_matches or occupancy_update_matches or hdr" > $OUT/pytest.txt 2>&1
tail -2 $OUT/pytest.txt
V=$PWD/ngp_pl_amd/csrc/variants/libngp_hip_fwdstage_old.so
NGP_HIP_LIB=$V python tools/bench_mlp.py 390000 165000 2>&1 | grep "^lib" | cut -c1-110
python tools/bench_mlp.py 390000 165000 2>&1 | grep "^lib" | cut -c1-110
B="python bench.py --no-cpu-baseline --no-api --steps 20 --warmup 5"
for i in 1 2; do
  NGP_HIP_LIB=$V $B > $OUT/old_$i.json 2> $OUT/old_$i.err
  $B > $OUT/new_$i.json 2> $OUT/new_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04ad/*.json")):
    r = json.loads(open(f).read().strip().splitlines()[-1])
    st = dict((d["stage"], d["ms"]) for d in r["roofline"]["stages"])
    fr = r.get("full_run") or {}
    print(f.split("/")[-1], "ms/step %.4f" % r["ms_per_step"], "mlp_fwd", st.get("mlp_fwd"), "full_run", {k: (round(fr[k], 3) if isinstance(fr.get(k), float) else None) for k in ("train_s", "psnr", "fps_200", "fps_200_reference_chunking")})
PY
