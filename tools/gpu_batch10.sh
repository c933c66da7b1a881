#!/bin/bash
mkdir -p gpurun_out/b10
export NGP_SPIN_TIMEOUT_S=20
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/b10/tests.txt 2>&1; echo "rc=$?" >> gpurun_out/b10/tests.txt
tail -n 25 gpurun_out/b10/tests.txt | grep -v "^$"
