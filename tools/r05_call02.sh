#!/bin/bash
# round 5, call 2: reference-files tests again (autocast fix), kernel traces of the three workloads, step gaps
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_reference_files_gpu.py -q -m gpu -p no:cacheprovider > $OUT/r05_c02_tests.log 2>&1
echo "pytest rc=$?" >> $OUT/r05_c02_tests.log; tail -15 $OUT/r05_c02_tests.log
cd /tmp && export TMPDIR=/tmp
for W in lego unbounded lego16k; do
  rm -rf /tmp/kt_$W
  timeout 300 rocprofv3 --kernel-trace -d /tmp/kt_$W -o bench -- python $REPO/bench.py --workload $W --steps 20 --warmup 5 --timed-only > $OUT/r05_c02_trace_$W.json 2> $OUT/r05_c02_trace_$W.err
  DB=$(find /tmp/kt_$W -name "*.db" | head -1)
  python $REPO/tools/rocprof_summary.py $DB 200 > $OUT/r05_c02_kernel_trace_$W.txt 2>&1
  python $REPO/tools/step_gaps.py $DB 96 > $OUT/r05_c02_step_gaps_$W.txt 2>&1
  python $REPO/tools/step_timeline.py $DB 2 3 > $OUT/r05_c02_timeline_$W.txt 2>&1
  python $REPO/tools/step_timeline.py $DB 1 0 occ_scan_kernel > $OUT/r05_c02_timeline_update_$W.txt 2>&1
done
cd $REPO
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-full-run > $OUT/r05_c02_bench.json 2> $OUT/r05_c02_bench.err
echo "bench rc=$?"; tail -12 $OUT/r05_c02_bench.err
head -12 $OUT/r05_c02_step_gaps_lego.txt
