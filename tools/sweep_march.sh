#!/bin/bash
# Scratch: step time for every placement of the next batch's march (NGP_MARCH_AT) with both pass-1 kernels
# (NGP_MARCH_WAVE).  Run on an MI355X, e.g.  gpurun --timeout 240 -- 'bash tools/sweep_march.sh | tee gpurun_out/sweep_march.txt'
cd "$(dirname "$0")/.."
for wave in 1 0; do
  for at in top hashgrid_fwd mlp_fwd mlp_bwd hashgrid_bwd; do
    for rep in 1 2; do
      NGP_MARCH_WAVE=$wave NGP_MARCH_AT=$at timeout 60 python bench.py --no-cpu-baseline --no-render --timed-only 2>/dev/null |
        python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wave=$wave at=$at rep=$rep ms_per_step=%.4f' % d['ms_per_step'])"
    done
  done
done
