#!/bin/bash
mkdir -p gpurun_out/b9
export NGP_SPIN_TIMEOUT_S=20
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -q -x -s -k "bench_batch_match" > gpurun_out/b9/oracle_bwd.txt 2>&1; echo "rc=$?" >> gpurun_out/b9/oracle_bwd.txt
tail -n 30 gpurun_out/b9/oracle_bwd.txt | grep -v "^$"
