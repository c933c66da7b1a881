#!/bin/bash
# round 5, call 23: 8^3 + 4^3 block hops from LDS bits -- frame parity tests, hops on / off on the trained fields, per-wave clocks
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd "$REPO"
OUT="$REPO/gpurun_out/call23"; mkdir -p "$OUT"
timeout 700 python -m pytest tests/test_train_gpu.py tests/test_reference_surface_gpu.py tests/test_reference_files_gpu.py -x -q -m gpu -k "frame_loop or native_test_renderer or render" > "$OUT/tests.log" 2>&1
tail -3 "$OUT/tests.log"
timeout 300 python tools/frame_hops_ab.py lego 20000 > "$OUT/ab_lego.log" 2>&1; grep -v amdgpu.ids "$OUT/ab_lego.log" | tail -13
timeout 300 python tools/frame_hops_ab.py lego_hard 20000 > "$OUT/ab_lego_hard.log" 2>&1; grep -v amdgpu.ids "$OUT/ab_lego_hard.log" | tail -13
NGP_HIP_LIB=$PWD/ngp_pl_amd/csrc/variants/libngp_hip_render_timing.so timeout 200 python tools/render_wave_times.py 20000 2>&1 | grep -v amdgpu.ids | tee "$OUT/wave_times.txt" | cut -c1-330
