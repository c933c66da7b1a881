#!/bin/bash
# round 5, call 6: A/B of the two-launch march on ONE box (alternating), then the in-process step-gap accounting
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out
for rep in 1 2; do for f in 0 1; do
  NGP_MARCH_FUSED=$f timeout 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-full-run --no-render > $OUT/r05_c06_ab_${f}_$rep.json 2> $OUT/r05_c06_ab_${f}_$rep.err
done; done
python - <<'P'
import json
for f in (0,1):
  for rep in (1,2):
    d=json.load(open('gpurun_out/r05_c06_ab_%d_%d.json'%(f,rep)))
    st={s['stage'][:10]:s['ms'] for s in d['roofline']['stages']}
    print("fused=%d rep %d  step %.4f  api %.4f  plain %.4f  ref_files %.3f | %s" % (f,rep,d['ms_per_step'],d['api_path']['ms_per_step'],d['api_path_plain']['ms_per_step'],d['api_path_reference_files']['ms_per_step'], " ".join("%s=%.4f"%kv for kv in st.items())))
P
timeout 200 python tools/step_gaps.py > $OUT/r05_c06_step_gaps.txt 2> $OUT/r05_c06_step_gaps.err; cat $OUT/r05_c06_step_gaps.txt; tail -2 $OUT/r05_c06_step_gaps.err
