#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ka; timeout 300 rocprofv3 --kernel-trace -d /tmp/ka -o a -- python /root/repo/tools/api_steps.py > /root/repo/gpurun_out/api_trace.txt 2>/dev/null
DB=$(find /tmp/ka -name "*.db" | head -1)
python /root/repo/tools/step_timeline.py $DB 2 3 >> /root/repo/gpurun_out/api_trace.txt 2>&1
