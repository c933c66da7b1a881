#!/bin/bash
# round 3, GPU batch 1
mkdir -p gpurun_out/b1
export NGP_SPIN_TIMEOUT_S=20
timeout 300 python tools/repro_unbounded.py --steps 700 > gpurun_out/b1/unbounded.txt 2>&1; echo "unbounded rc=$?" >> gpurun_out/b1/unbounded.txt
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_field_gpu.py tests/test_ddp_gpu.py -m gpu -x -q > gpurun_out/b1/tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/b1/tests.txt
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/b1/bench.json 2> gpurun_out/b1/bench.err
NGP_NATIVE_STEP=0 timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b1/bench_pystep.json 2> gpurun_out/b1/bench_pystep.err
for lib in ngp_pl_amd/csrc/libngp_hip.so ngp_pl_amd/csrc/variants/libngp_hip_nopayload.so ngp_pl_amd/csrc/variants/libngp_hip_pb4.so ngp_pl_amd/csrc/variants/libngp_hip_pb12.so ngp_pl_amd/csrc/variants/libngp_hip_timing.so ngp_pl_amd/csrc/variants/libngp_hip_timing_nopayload.so; do
  NGP_HIP_LIB=$PWD/$lib timeout 120 python tools/bench_bwd.py 155000 2>&1 | grep -v amdgpu.ids >> gpurun_out/b1/bwd_ab.txt
done
NGP_HIP_LIB=$PWD/ngp_pl_amd/csrc/variants/libngp_hip_nopayload.so timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-render --no-api > gpurun_out/b1/bench_nopayload.json 2> gpurun_out/b1/bench_nopayload.err
tail -3 gpurun_out/b1/unbounded.txt gpurun_out/b1/tests.txt; cat gpurun_out/b1/bwd_ab.txt | head -40
