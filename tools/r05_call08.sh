#!/bin/bash
# round 5, call 8: the direct exchange on a 1-rank RCCL communicator (library mode 2 + torch.distributed mirror), bench under a 1-rank group in all three modes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out
timeout 900 python -m pytest tests/test_ddp_gpu.py tests/test_bench_gpu.py -q -m gpu -p no:cacheprovider -k "not driver_command and not cold_start" > $OUT/r05_c08_tests.log 2>&1
echo "pytest rc=$?" >> $OUT/r05_c08_tests.log; tail -25 $OUT/r05_c08_tests.log
