#!/bin/bash
# The table backward alone for the product library and every A/B build under ngp_pl_amd/csrc/variants/ (tools/build_variant.sh);
# a build named *t2w<N>* was compiled with -DNGP_BIN_TIMING2=<N> (per-phase marks of wave N of every hashed task).
cd "$(dirname "$0")/.."
for lib in ngp_pl_amd/csrc/libngp_hip.so ngp_pl_amd/csrc/variants/*.so; do
  [ -f "$lib" ] || continue
  t2=""; case "$lib" in *t2w*) t2=$(echo "$lib" | sed -E 's/.*t2w([0-9]+).*/\1/');; esac
  NGP_BIN_T2=$t2 NGP_HIP_LIB=$PWD/$lib python tools/bench_bwd.py ${1:-165000} 2>&1 | grep -v amdgpu.ids
done
