#!/bin/bash
mkdir -p gpurun_out/b12
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt2
NGP_TWO_ROUND=on STEPS=3000 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt2 -o late -- python $GRAFT_REPO_ROOT/tools/late_stage_times.py > $GRAFT_REPO_ROOT/gpurun_out/b12/run.txt 2>&1
DB=$(find /tmp/kt2 -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py "$DB" 30 > $GRAFT_REPO_ROOT/gpurun_out/b12/kernels.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/b12/kernels.txt | head -30; tail -2 $GRAFT_REPO_ROOT/gpurun_out/b12/run.txt
