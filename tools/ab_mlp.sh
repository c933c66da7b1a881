#!/bin/bash
# tools/bench_mlp.py (the fused field forward / backward alone) over the product library and every A/B build under variants/.
#   gpurun -- 'bash tools/gpu_call.sh <tag> "sh:ab_mlp.sh [n_active ...]"'
cd "$(dirname "$0")/.."
for lib in ngp_pl_amd/csrc/libngp_hip.so ngp_pl_amd/csrc/variants/*mlp*.so; do
  [ -f "$lib" ] || continue
  if [ $# -eq 0 ]; then
    NGP_HIP_LIB=$PWD/$lib python tools/bench_mlp.py 340000 166000 2>&1 | grep -v amdgpu.ids
  else
    for a in "$@"; do NGP_HIP_LIB=$PWD/$lib python tools/bench_mlp.py 340000 $a 2>&1 | grep -v amdgpu.ids; done
  fi
done
