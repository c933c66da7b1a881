#!/bin/bash
# Scratch: build an A/B variant of libngp_hip.so with extra -D flags for ONE translation unit.
#   tools/build_variant.sh <name> <source.hip> -DFOO=1 ...   ->  ngp_pl_amd/csrc/variants/libngp_hip_<name>.so
# Run the product against it with NGP_HIP_LIB=<path> (ngp_pl_amd/_lib.py).
set -e
cd "$(dirname "$0")/../ngp_pl_amd/csrc"
name=$1; src=$2; shift 2
mkdir -p variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-result"
if [ "$src" == "mlp.hip" ]; then FLAGS="$FLAGS -mllvm -amdgpu-mfma-vgpr-form"; fi
/opt/rocm/bin/hipcc $FLAGS "$@" -c $src -o variants/${src%.hip}_$name.o
objs=""
for f in march composite hashgrid mlp optim occupancy hashgrid_bwd_binned stepper comm; do
  if [ "$f.hip" == "$src" ]; then objs="$objs variants/${f}_$name.o"; else objs="$objs $f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o variants/libngp_hip_$name.so
echo variants/libngp_hip_$name.so
