#!/bin/bash
# tools/hard_divergence.py (30 000 steps on lego_hard_big: parameter ranges + skipped steps every 1000) over the product library and every
# A/B build under variants/.      gpurun -- 'bash tools/gpu_call.sh <tag> "sh:ab_hard.sh"'
cd "$(dirname "$0")/.."
for lib in ngp_pl_amd/csrc/libngp_hip.so ngp_pl_amd/csrc/variants/*.so; do
  [ -f "$lib" ] || continue
  echo "== $lib"
  NGP_HIP_LIB=$PWD/$lib python tools/hard_divergence.py lego_hard_big 30000 1000 2>&1 | grep -v amdgpu.ids | grep -E "skipped|check\]|first|replay" | awk 'NR%2==0 || /first|replay/' | tail -34
done
