#!/bin/bash
mkdir -p gpurun_out/b6
export NGP_SPIN_TIMEOUT_S=20
for legs in "roofline,render,render_ref,api" "roofline" "render" "render_ref" "api" "none"; do
  echo "== legs $legs" >> gpurun_out/b6/trace3.txt
  timeout 200 python tools/guard_trace3.py --legs $legs 2>&1 | grep -v amdgpu >> gpurun_out/b6/trace3.txt
done
cat gpurun_out/b6/trace3.txt
