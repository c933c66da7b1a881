#!/bin/bash
# round 3, GPU batch 2
mkdir -p gpurun_out/b2
export NGP_SPIN_TIMEOUT_S=20
NGP_BENCH_DEADLINE_S=200 timeout 260 python bench.py --gpus 1 --steps 20 --warmup 5 --secondary --no-cpu-baseline > gpurun_out/b2/bench_secondary.json 2> gpurun_out/b2/bench_secondary.err; echo "rc=$?" >> gpurun_out/b2/bench_secondary.err
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_field_gpu.py tests/test_ddp_gpu.py -m gpu -x -q > gpurun_out/b2/tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/b2/tests.txt
for lib in ngp_pl_amd/csrc/libngp_hip.so ngp_pl_amd/csrc/variants/libngp_hip_noruns.so ngp_pl_amd/csrc/variants/libngp_hip_dense_b2.so ngp_pl_amd/csrc/variants/libngp_hip_timing.so ngp_pl_amd/csrc/variants/libngp_hip_timing_noruns.so; do
  NGP_HIP_LIB=$PWD/$lib timeout 120 python tools/bench_bwd.py 155000 2>&1 | grep -v amdgpu.ids >> gpurun_out/b2/bwd_ab.txt
done
for lib in timing timing_noruns; do echo "== $lib" >> gpurun_out/b2/bin_tasks.txt; NGP_HIP_LIB=$PWD/ngp_pl_amd/csrc/variants/libngp_hip_$lib.so timeout 120 python tools/profile_bin_tasks.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/b2/bin_tasks.txt; done
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b2/bench.json 2> gpurun_out/b2/bench.err
NGP_HIP_LIB=$PWD/ngp_pl_amd/csrc/variants/libngp_hip_noruns.so timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-render --no-api > gpurun_out/b2/bench_noruns.json 2> gpurun_out/b2/bench_noruns.err
tail -n 4 gpurun_out/b2/tests.txt; cat gpurun_out/b2/bwd_ab.txt gpurun_out/b2/bin_tasks.txt | head -60; tail -n 12 gpurun_out/b2/bench_secondary.err
