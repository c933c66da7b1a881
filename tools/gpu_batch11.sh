#!/bin/bash
mkdir -p gpurun_out/b11
export NGP_SPIN_TIMEOUT_S=20
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -q -x -k "two_round" > gpurun_out/b11/tests.txt 2>&1; echo "rc=$?" >> gpurun_out/b11/tests.txt
tail -n 12 gpurun_out/b11/tests.txt | grep -v "^$"
: > gpurun_out/b11/late.txt
for mode in off on; do echo "== NGP_TWO_ROUND=$mode" >> gpurun_out/b11/late.txt; NGP_TWO_ROUND=$mode STEPS=8000 timeout 200 python tools/late_stage_times.py 2>&1 | tail -2 >> gpurun_out/b11/late.txt; done
cat gpurun_out/b11/late.txt
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt2
NGP_TWO_ROUND=on STEPS=8000 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt2 -o late -- python $GRAFT_REPO_ROOT/tools/late_stage_times.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py "$(find /tmp/kt2 -name '*.db' | head -1)" 30 2>/dev/null | head -16
