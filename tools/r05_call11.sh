#!/bin/bash
# round 5, call 11: box reciprocal by v_rcp_f32 (product) against the correctly rounded division (variant), same box, + the field tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
bash tools/ab_variants.sh "tests/test_field_gpu.py" > gpurun_out/r05_c11_ab.txt 2>&1
cat gpurun_out/r05_c11_ab.txt
