#!/bin/bash
mkdir -p gpurun_out/b13
: > gpurun_out/b13/two_round.txt
for cfg in "off 8" "on 8" "on 16" "on 32" "on 64"; do set -- $cfg; echo "== NGP_TWO_ROUND=$1 K=$2, 8000 steps" >> gpurun_out/b13/two_round.txt; NGP_TWO_ROUND=$1 NGP_TWO_ROUND_K=$2 STEPS=8000 timeout 200 python tools/late_stage_times.py 2>&1 | tail -3 | grep -v "two-round counters" >> gpurun_out/b13/two_round.txt; done
for cfg in "off 8" "on 32"; do set -- $cfg; echo "== NGP_TWO_ROUND=$1 K=$2, 25000 steps" >> gpurun_out/b13/two_round.txt; NGP_TWO_ROUND=$1 NGP_TWO_ROUND_K=$2 STEPS=25000 timeout 200 python tools/late_stage_times.py 2>&1 | tail -3 | grep -v "two-round counters" >> gpurun_out/b13/two_round.txt; done
cat gpurun_out/b13/two_round.txt
