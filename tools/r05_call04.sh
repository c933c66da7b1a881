#!/bin/bash
# round 5, call 4: the whole GPU suite after the shrink (removed variants, internal header, Python-enqueued step gone) + the driver's bench command
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -x --durations=15 > $OUT/r05_c04_tests.log 2>&1
echo "pytest rc=$?" >> $OUT/r05_c04_tests.log; tail -30 $OUT/r05_c04_tests.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_c04_bench.json 2> $OUT/r05_c04_bench.err
echo "bench rc=$?"; grep -c . $OUT/r05_c04_bench.json; tail -5 $OUT/r05_c04_bench.err
