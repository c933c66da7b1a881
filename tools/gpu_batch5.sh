#!/bin/bash
mkdir -p gpurun_out/b5
export NGP_SPIN_TIMEOUT_S=20
for cfg in "--prelude lego --sync 0" "--prelude lego --sync 1" "--prelude none --sync 0" "--prelude lego --sync 0 --prelude-steps 1"; do
  echo "== $cfg" >> gpurun_out/b5/trace2.txt
  timeout 200 python tools/guard_trace2.py $cfg 2>&1 | grep -v amdgpu >> gpurun_out/b5/trace2.txt
done
NGP_NATIVE_STEP=0 timeout 200 python tools/guard_trace2.py --prelude lego --sync 0 2>&1 | grep -v amdgpu > gpurun_out/b5/trace2_pystep.txt
cat gpurun_out/b5/trace2.txt; echo "== python step"; cat gpurun_out/b5/trace2_pystep.txt
