#!/bin/bash
# The table backward alone (tools/bench_bwd.py) for the product library and every A/B build under ngp_pl_amd/csrc/variants/
# (tools/build_variant.sh <name> hashgrid_bwd_binned.hip -D...; a -DNGP_BIN_TIMING=1 build also prints per-task phases).
cd "$(dirname "$0")/.."
for lib in ngp_pl_amd/csrc/libngp_hip.so ngp_pl_amd/csrc/variants/*.so; do
  [ -f "$lib" ] || continue
  NGP_HIP_LIB=$PWD/$lib python tools/bench_bwd.py ${1:-165000} 2>&1 | grep -v amdgpu.ids
done
