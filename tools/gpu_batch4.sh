#!/bin/bash
mkdir -p gpurun_out/b4
export NGP_SPIN_TIMEOUT_S=20
timeout 300 python tools/guard_trace.py > gpurun_out/b4/guard_trace.txt 2>&1; echo "rc=$?" >> gpurun_out/b4/guard_trace.txt
timeout 300 python -m pytest tests/test_train_gpu.py -m gpu -q -k "native_stepper" > gpurun_out/b4/tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/b4/tests.txt
grep -v amdgpu gpurun_out/b4/guard_trace.txt; tail -n 5 gpurun_out/b4/tests.txt
