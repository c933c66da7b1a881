#!/bin/bash
# Scratch: step time + stage times of the default bench for every A/B library under ngp_pl_amd/csrc/variants/ (tools/build_variant.sh)
# and for the product build; optional correctness check of each with a pytest selection ($1).
#   gpurun --timeout 600 -- 'bash tools/ab_variants.sh "tests/test_field_gpu.py -k binned" > gpurun_out/ab.txt 2>&1'
cd "$(dirname "$0")/.."
for lib in ngp_pl_amd/csrc/libngp_hip.so ngp_pl_amd/csrc/variants/*.so; do
  [ -f "$lib" ] || continue
  echo "== $lib"
  if [ -n "$1" ]; then NGP_HIP_LIB=$PWD/$lib timeout 300 python -m pytest $1 -m gpu -q -x -p no:cacheprovider 2>&1 | tail -1; fi
  for rep in 1 2; do
    NGP_BENCH_DETAIL=/tmp/ab_detail.json NGP_HIP_LIB=$PWD/$lib timeout 120 python bench.py --steps 20 --warmup 5 --no-render --no-cpu-baseline --no-secondary --no-api --no-full-run >/dev/null 2>&1
    python -c "
import sys, json
d = json.load(open('/tmp/ab_detail.json'))
st = {s['stage']: s['ms'] for s in d['roofline']['stages']}
print('  ms_per_step=%.4f  ' % d['ms_per_step'] + '  '.join('%s=%.4f' % (k, st[k]) for k in ('hashgrid_bwd', 'hashgrid_fwd', 'mlp_bwd', 'mlp_fwd', 'adam', 'composite_bw', 'march_count(side stream)') if k in st))"
  done
done
