#!/bin/bash
# Scratch: step time + stage times of the default bench under settings of ONE environment variable, alternating, same box.
#   gpurun -- 'bash tools/gpu_call.sh <tag> "sh:ab_env.sh NGP_LISTS_BESIDE 0 1"'
cd "$(dirname "$0")/.."
var=$1; shift
for rep in 1 2 3; do
  for val in "$@"; do
    env $var=$val NGP_BENCH_DETAIL=/tmp/ab_detail.json timeout 120 python bench.py --steps 20 --warmup 5 --no-render --no-cpu-baseline --no-secondary --no-api --no-full-run >/dev/null 2>&1
    python -c "
import sys, json
d = json.load(open('/tmp/ab_detail.json'))
st = {s['stage']: s['ms'] for s in d['roofline']['stages']}
print('$var=$val  ms_per_step=%.4f  ' % d['ms_per_step'] + '  '.join('%s=%.4f' % (k.split('(')[0], st[k]) for k in ('hashgrid_bwd', 'hashgrid_bwd_lists(beside mlp_bwd)', 'hashgrid_fwd', 'mlp_bwd', 'mlp_fwd', 'adam', 'composite_bw', 'march_count(side stream)') if k in st), ' roofline avg_ms', d['roofline']['avg_ms'], 'frac %.4f' % d['roofline']['frac'])"
  done
done
