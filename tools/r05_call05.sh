#!/bin/bash
# round 5, call 5: two-launch march in the stepper + mark_invisible kernel: GPU suite (without the bench tests) + the driver's bench command
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_bench_gpu.py > $OUT/r05_c05_tests.log 2>&1
echo "pytest rc=$?" >> $OUT/r05_c05_tests.log; tail -25 $OUT/r05_c05_tests.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_c05_bench.json 2> $OUT/r05_c05_bench.err
echo "bench rc=$?"; tail -3 $OUT/r05_c05_bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r05_c05_bench.json'))
print("value", d['value'], d['ms_per_step'], [(s['stage'][:12], s['ms']) for s in d['roofline']['stages']])
for k in ('api_path','api_path_plain','api_path_reference_files'): print(k, d[k].get('rays_per_s'), d[k].get('ms_per_step'), d[k].get('error'))
print([(x['workload'][:11], round(x['rays_per_s']/1e6,2)) for x in d['secondary']], d['full_run']['train_s'], d['render_fps_800x800']['fps'])
P
