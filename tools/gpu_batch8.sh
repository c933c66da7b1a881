#!/bin/bash
mkdir -p gpurun_out/b8
export NGP_SPIN_TIMEOUT_S=20
for lib in ngp_pl_amd/csrc/libngp_hip.so ngp_pl_amd/csrc/variants/libngp_hip_spt4.so ngp_pl_amd/csrc/variants/libngp_hip_spt4_s3456x2.so ngp_pl_amd/csrc/variants/libngp_hip_spt4_s4608x2.so ngp_pl_amd/csrc/variants/libngp_hip_s3456x2.so ngp_pl_amd/csrc/variants/libngp_hip_spt4_s3456x2_t512.so; do
  NGP_HIP_LIB=$PWD/$lib timeout 120 python tools/bench_bwd.py 155000 2>&1 | grep -v amdgpu.ids >> gpurun_out/b8/bwd_ab.txt
done
cat gpurun_out/b8/bwd_ab.txt
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/b8/tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/b8/tests.txt
tail -n 5 gpurun_out/b8/tests.txt
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/b8/bench.json 2> gpurun_out/b8/bench.err
