#!/bin/bash
# round 5, call 9: the whole GPU suite + the driver's command, on the final tree
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=8 > $OUT/r05_c09_tests.log 2>&1
echo "pytest rc=$?" >> $OUT/r05_c09_tests.log; tail -22 $OUT/r05_c09_tests.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_c09_bench.json 2> $OUT/r05_c09_bench.err
echo "bench rc=$?"; tail -3 $OUT/r05_c09_bench.err
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
