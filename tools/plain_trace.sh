#!/bin/bash
# Scratch: kernel timeline (last steps) + summary of the plain single-GPU bench.   gpurun -- 'bash tools/plain_trace.sh'
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt3; timeout 300 rocprofv3 --kernel-trace -d /tmp/kt3 -o bench -- python /root/repo/bench.py --steps 20 --warmup 5 --timed-only > /root/repo/gpurun_out/plain_trace_bench.json 2> /root/repo/gpurun_out/plain_trace.err
DB=$(find /tmp/kt3 -name "*.db" | head -1)
python /root/repo/tools/step_timeline.py $DB 3 4 > /root/repo/gpurun_out/plain_timeline.txt 2>&1
python /root/repo/tools/step_timeline.py $DB 1 0 occ_scan_kernel > /root/repo/gpurun_out/plain_timeline_update.txt 2>&1
python /root/repo/tools/rocprof_summary.py $DB 200 > /root/repo/gpurun_out/plain_trace_summary.txt 2>&1
