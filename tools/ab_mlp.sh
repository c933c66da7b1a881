#!/bin/bash
# tools/bench_mlp.py (the fused field forward / backward alone) over the product library and every A/B build under variants/;
# the product's outputs are saved and every variant is compared with them bit for bit (dfeats, dW sums and partial rows, dh).
#   gpurun -- 'bash tools/gpu_call.sh <tag> "sh:ab_mlp.sh [n_active ...]"'
cd "$(dirname "$0")/.."
acts="$@"; [ -z "$acts" ] && acts=166000
for a in $acts; do
  NGP_MLP_DUMP=/tmp/mlp_ref_$a.pt python tools/bench_mlp.py 340000 $a 2>&1 | grep -v amdgpu.ids
  for lib in ngp_pl_amd/csrc/variants/*mlp*.so; do
    [ -f "$lib" ] || continue
    NGP_MLP_CMP=/tmp/mlp_ref_$a.pt NGP_HIP_LIB=$PWD/$lib python tools/bench_mlp.py 340000 $a 2>&1 | grep -v amdgpu.ids
  done
done
