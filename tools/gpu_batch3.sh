#!/bin/bash
# round 3, GPU batch 3
mkdir -p gpurun_out/b3
export NGP_SPIN_TIMEOUT_S=20
NGP_BENCH_DEADLINE_S=200 timeout 260 python bench.py --gpus 1 --steps 20 --warmup 5 --secondary --no-cpu-baseline > gpurun_out/b3/bench_secondary.json 2> gpurun_out/b3/bench_secondary.err; echo "rc=$?" >> gpurun_out/b3/bench_secondary.err
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_field_gpu.py tests/test_ddp_gpu.py -m gpu -q > gpurun_out/b3/tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/b3/tests.txt
for lib in ngp_pl_amd/csrc/libngp_hip.so ngp_pl_amd/csrc/variants/libngp_hip_notranspose.so ngp_pl_amd/csrc/variants/libngp_hip_timing.so ngp_pl_amd/csrc/variants/libngp_hip_timing_notranspose.so; do
  NGP_HIP_LIB=$PWD/$lib timeout 120 python tools/bench_bwd.py 155000 2>&1 | grep -v amdgpu.ids >> gpurun_out/b3/bwd_ab.txt
done
for lib in timing timing_notranspose; do echo "== $lib" >> gpurun_out/b3/bin_tasks.txt; NGP_HIP_LIB=$PWD/ngp_pl_amd/csrc/variants/libngp_hip_$lib.so timeout 120 python tools/profile_bin_tasks.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/b3/bin_tasks.txt; done
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b3/bench.json 2> gpurun_out/b3/bench.err
NGP_HIP_LIB=$PWD/ngp_pl_amd/csrc/variants/libngp_hip_notranspose.so timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-render --no-api > gpurun_out/b3/bench_notranspose.json 2> gpurun_out/b3/bench_notranspose.err
tail -n 6 gpurun_out/b3/tests.txt; cat gpurun_out/b3/bwd_ab.txt gpurun_out/b3/bin_tasks.txt | head -60
