#!/bin/bash
# Scratch: tools/bench_mlp.py over the built library and every A/B variant.   gpurun -- 'bash tools/ab_mlp.sh [n_samples n_active]...'
cd /root/repo
for lib in ngp_pl_amd/csrc/libngp_hip.so ngp_pl_amd/csrc/variants/*.so; do
  if [ $# -eq 0 ]; then
    NGP_HIP_LIB=$PWD/$lib python tools/bench_mlp.py 2>&1 | grep -v amdgpu.ids
  else
    for a in "$@"; do NGP_HIP_LIB=$PWD/$lib python tools/bench_mlp.py 290000 $a 2>&1 | grep -v amdgpu.ids; done
  fi
done
