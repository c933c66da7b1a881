#!/bin/bash
# round 5, call 1: the new reference-files tests, the tightened composite tests, the bench's new legs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_reference_files_gpu.py tests/test_vren_gpu.py tests/test_bench_gpu.py -q -m gpu -k "not driver_command" -p no:cacheprovider > gpurun_out/r05_c01_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05_c01_tests.log
tail -40 gpurun_out/r05_c01_tests.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_c01_bench.json 2> gpurun_out/r05_c01_bench.err
echo "bench rc=$?"
tail -30 gpurun_out/r05_c01_bench.err
