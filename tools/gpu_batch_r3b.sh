#!/bin/bash
# Scratch: round-3 late batch B -- kernel trace of late training steps, march placement late in training.
REPO=/root/repo; O=$REPO/gpurun_out/r3b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_late && STEPS=8000 NGP_TWO_ROUND_K=32 timeout 300 rocprofv3 --kernel-trace -d /tmp/kt_late -o late -- python $REPO/tools/late_stage_times.py > $O/late_under_trace.txt 2> $O/trace.err
DB=$(find /tmp/kt_late -name "*.db" | head -1)
python $REPO/tools/rocprof_summary.py "$DB" 200 > $O/late_kernel_trace_summary.txt 2>> $O/trace.err
cd $REPO
for at in top hashgrid_fwd mlp_fwd composite_bw mlp_bwd hashgrid_bwd; do echo "== NGP_MARCH_AT=$at 8000"; NGP_MARCH_AT=$at STEPS=8000 timeout 120 python tools/late_stage_times.py 2>&1 | grep -v amdgpu.ids | grep "ms/step\|stages"; done > $O/late_march_at.txt 2>&1
cat $O/late_kernel_trace_summary.txt $O/late_march_at.txt
