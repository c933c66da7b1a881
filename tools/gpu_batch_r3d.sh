#!/bin/bash
# Scratch: round-3 batch D -- cell-run reuse in the hash forward: parity and per-level times.
cd /root/repo; O=gpurun_out/r3d; mkdir -p $O
NGP_FWD_REUSE_MAX_RES=4096 timeout 300 python -m pytest tests/test_hashgrid_gpu.py tests/test_field_gpu.py -x -q -m gpu -k "forward or fwd or encode or matches" > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/tests.txt
for r in 0 4096; do NGP_FWD_REUSE_MAX_RES=$r python tools/profile_fwd_levels.py 2>&1 | grep -v amdgpu.ids; done > $O/levels.txt
for r in 64 128 256 512; do NGP_FWD_REUSE_MAX_RES=$r python tools/profile_fwd_levels.py 2>&1 | grep -v amdgpu.ids | head -2; done >> $O/levels.txt
tail -n 3 $O/tests.txt; cat $O/levels.txt
