#!/bin/bash
# round 5, call 26: the whole GPU suite + smoke() on the final tree (after the reference-files FPS leg and the late tests)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=5 > gpurun_out/r05_c26_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05_c26_tests.log; tail -10 gpurun_out/r05_c26_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
