#!/bin/bash
# round 5, call 19: final tree -- rocprofv3 kernel trace of frames on the trained field (what roofline_render.builder_profile reads), the whole
# GPU suite, the driver's command, smoke()
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out
FRAMES=20 STEPS=30000 bash tools/render_trained_trace.sh r05 > $OUT/r05_c19_render_trace.log 2>&1; tail -3 $OUT/r05_c19_render_trace.log | cut -c1-200
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=8 > $OUT/r05_c19_tests.log 2>&1
echo "pytest rc=$?" >> $OUT/r05_c19_tests.log; tail -14 $OUT/r05_c19_tests.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_c19_bench.json 2> $OUT/r05_c19_bench.err
echo "bench rc=$?"; tail -3 $OUT/r05_c19_bench.err
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
