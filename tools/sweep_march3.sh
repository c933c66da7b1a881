#!/bin/bash
# Placement of the next batch's march under the native stepper (NGP_MARCH_AT), three runs each, ms per step of the timed windows.
mkdir -p gpurun_out
OUT=gpurun_out/march_sweep_r03.txt
: > $OUT
for rep in 1 2 3; do
for at in top hashgrid_fwd mlp_fwd composite_fw composite_bw mlp_bwd hashgrid_bwd adam; do
  NGP_MARCH_AT=$at timeout 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-render --no-api 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
st = {s['stage']: s['ms'] for s in r['stages']}
print('%-14s rep $rep  %.4f ms/step  %.2f M rays/s  stages: %s' % ('$at', d['ms_per_step'], d['value'] / 1e6, ' '.join('%s=%.3f' % (k, st[k]) for k in ('hashgrid_fwd', 'mlp_fwd', 'composite_fw+loss', 'composite_bw', 'mlp_bwd', 'hashgrid_bwd', 'adam', 'march_count(side stream)') if k in st)))" >> $OUT
done
done
sort $OUT
