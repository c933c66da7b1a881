#!/bin/bash
# round 5, call 10: hash forward's workgroup-map lookup on the lanes (product) against the scalar walk (variant), same box, + the forward tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
bash tools/ab_variants.sh "tests/test_field_gpu.py -k forward" > gpurun_out/r05_c10_ab.txt 2>&1
cat gpurun_out/r05_c10_ab.txt
