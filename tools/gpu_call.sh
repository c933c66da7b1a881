#!/bin/bash
# The one GPU-box runner (replaces the per-call scripts of earlier rounds):
#   gpurun --timeout 1800 -- 'bash tools/gpu_call.sh <tag> <what> [<what> ...]'
# <what>: suite   -- the whole GPU test suite + smoke()
#         bench   -- the driver's literal bench command; line -> gpurun_out/<tag>_bench_line.json, detail -> <tag>_bench_detail.json
#         bench3  -- the same three times back to back (run-to-run spread on one box)
#         profile -- tools/profile_bench.sh <tag> (kernel trace + PMC passes of the driver-shaped command)
#         py:<script> [args] -- python tools/<script> ..., log -> gpurun_out/<tag>_<script>.log   (quote the whole word)
#         sh:<script> [args] -- bash tools/<script> ..., log -> gpurun_out/<tag>_<script>.log
#         parity  -- the end-to-end parity tests with NGP_PARITY_LOG set: their measured error distributions -> <tag>_parity_distribution.txt
#         test:<expr> -- pytest -m gpu -k <expr>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=$1; shift
mkdir -p gpurun_out
for what in "$@"; do
  case "$what" in
    suite)
      timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=8 > gpurun_out/${TAG}_tests.log 2>&1
      echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log; tail -12 gpurun_out/${TAG}_tests.log
      python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/${TAG}_smoke.log ;;
    bench|bench3)
      n=1; [ "$what" = bench3 ] && n=3
      for i in $(seq 1 $n); do
        NGP_BENCH_DETAIL=gpurun_out/${TAG}_bench_detail_$i.json timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 \
          > gpurun_out/${TAG}_bench_line_$i.json 2> gpurun_out/${TAG}_bench_stderr_$i.log
        echo "bench rc=$? bytes=$(wc -c < gpurun_out/${TAG}_bench_line_$i.json)"; cat gpurun_out/${TAG}_bench_line_$i.json
        grep -v "bench detail" gpurun_out/${TAG}_bench_stderr_$i.log | tail -40 > gpurun_out/${TAG}_bench_progress_$i.log
        rm -f gpurun_out/${TAG}_bench_stderr_$i.log
      done ;;
    profile)
      bash tools/profile_bench.sh $TAG 2>&1 | tail -5 ;;
    py:*)
      cmd=${what#py:}; name=$(echo "$cmd" | cut -d' ' -f1 | sed 's/\.py$//')
      timeout 900 python tools/$cmd > gpurun_out/${TAG}_${name}.log 2>&1; echo "$name rc=$?"; tail -40 gpurun_out/${TAG}_${name}.log ;;
    sh:*)
      cmd=${what#sh:}; name=$(echo "$cmd" | cut -d' ' -f1 | sed 's/\.sh$//')
      timeout 1200 bash tools/$cmd > gpurun_out/${TAG}_${name}.log 2>&1; echo "$name rc=$?"; tail -60 gpurun_out/${TAG}_${name}.log ;;
    parity)
      rm -f gpurun_out/${TAG}_parity_distribution.txt
      NGP_PARITY_LOG=$PWD/gpurun_out/${TAG}_parity_distribution.txt timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider \
        -k "end_to_end or references_render_on_the_binding or references_train_py" 2>&1 | tail -5
      cat gpurun_out/${TAG}_parity_distribution.txt ;;
    test:*)
      timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "${what#test:}" 2>&1 | tail -45 | tee gpurun_out/${TAG}_test_k.log ;;
    *) echo "unknown: $what" ;;
  esac
done
