#!/bin/bash
# round 5, call 22: per-wave clocks of the frame loop's thread-per-ray marcher (a -DNGP_RENDER_TIMING build)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
NGP_HIP_LIB=$PWD/ngp_pl_amd/csrc/variants/libngp_hip_render_timing.so timeout 200 python tools/render_wave_times.py 20000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_c22_wave_times.txt | cut -c1-420
