#!/bin/bash
# Scratch: round-3 late batch C -- scan-based compact first-round list: parity, late stage times, 30k-step run.
cd /root/repo; O=gpurun_out/r3c; mkdir -p $O
timeout 600 python -m pytest tests/test_vren_gpu.py tests/test_train_gpu.py -x -q -m gpu -k "first_k or count_k or two_round or native or raymarching_train" > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/tests.txt
for K in 32; do echo "== K=$K 8000"; NGP_TWO_ROUND_K=$K STEPS=8000 timeout 120 python tools/late_stage_times.py 2>&1 | grep -v amdgpu.ids; done > $O/late.txt 2>&1
timeout 300 python tools/train_eval.py > $O/train_eval.txt 2>&1
tail -n 4 $O/tests.txt; cat $O/late.txt; tail -c 1500 $O/train_eval.txt | head -c 600
