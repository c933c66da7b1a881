#!/bin/bash
# Round 4, call 20: hash-grid forward with 8-byte corner pairs: parity + A/B against -DNGP_FWD_PAIR=0 (step and frame loop).
OUT=gpurun_out/r04t; mkdir -p $OUT; rm -f $OUT/*.json
timeout 900 python -m pytest tests/test_field_gpu.py tests/test_properties_gpu.py -x -q -m gpu -k "forward or outside or variants or lds_resident or hashgrid" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
B="python bench.py --no-cpu-baseline --no-api --no-full-run"
V=$PWD/ngp_pl_amd/csrc/variants/libngp_hip_nopair.so
for i in 1 2; do
  NGP_HIP_LIB=$V $B > $OUT/nopair_$i.json 2> $OUT/nopair_$i.err
  $B > $OUT/pair_$i.json 2> $OUT/pair_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04t/*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        rf = r["roofline"]
        st = dict((d["stage"], d["ms"]) for d in rf["stages"])
        rd = r.get("render") or {}
        print(f.split("/")[-1], "ms/step %.4f" % r["ms_per_step"], "hashgrid_fwd", st.get("hashgrid_fwd"), "main sum", rf["main_stream_stage_sum_ms"],
              "render", {k: rd[k] for k in rd if "fps" in k or "ms" in k} if isinstance(rd, dict) else rd)
    except Exception as e:
        print(f, "unreadable", e)
PY
