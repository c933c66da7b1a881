#!/bin/bash
# Round 4, call 28: Adam-streaming slice owners, variants (streaming waves 2 / 3 / 4, with and without a chunk share, loads in flight).
OUT=gpurun_out/r04ab; mkdir -p $OUT; rm -f $OUT/*.json
B="python bench.py --no-render --no-cpu-baseline --no-api --no-full-run"
V=$PWD/ngp_pl_amd/csrc/variants
NGP_ADAM_IN_APPLY=0 $B > $OUT/0_sep.json 2> $OUT/0_sep.err
for v in nowalk_s4 nowalk_s2 nowalk_s3 nowalk_s4_u8 walk_s2; do
  NGP_HIP_LIB=$V/libngp_hip_$v.so NGP_ADAM_IN_APPLY=1 $B > $OUT/1_$v.json 2> $OUT/1_$v.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04ab/*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        rf = r["roofline"]
        st = dict((d["stage"], d["ms"]) for d in rf["stages"])
        print(f.split("/")[-1], "ms/step %.4f" % r["ms_per_step"], "hashgrid_bwd", st.get("hashgrid_bwd"), "adam", st.get("adam"), "sum", round(st.get("hashgrid_bwd", 0) + st.get("adam", 0), 4))
    except Exception as e:
        print(f, "unreadable", e)
PY
