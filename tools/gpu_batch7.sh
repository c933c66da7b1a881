#!/bin/bash
mkdir -p gpurun_out/b7
export NGP_SPIN_TIMEOUT_S=20
for legs in "roofline,render,render_ref,api" "api"; do
  echo "== legs $legs" >> gpurun_out/b7/trace3.txt
  timeout 200 python tools/guard_trace3.py --legs $legs 2>&1 | grep -v amdgpu >> gpurun_out/b7/trace3.txt
done
NGP_BENCH_DEADLINE_S=200 timeout 260 python bench.py --gpus 1 --steps 20 --warmup 5 --secondary --no-cpu-baseline > gpurun_out/b7/bench_secondary.json 2> gpurun_out/b7/bench_secondary.err
cat gpurun_out/b7/trace3.txt; python -c "
import json; d=json.load(open('gpurun_out/b7/bench_secondary.json')); print(d['value'], d['march_guards'], [(s.get('march_guards'), s.get('rays_per_s')) for s in d['secondary']])"
