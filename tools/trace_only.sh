#!/bin/bash
# Kernel trace of the driver-shaped bench command only (step 1 of tools/profile_bench.sh).
TAG=${1:-r03}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 20 --warmup 5 --no-render --no-cpu-baseline --no-api"
rm -rf /tmp/kt && timeout 90 rocprofv3 --kernel-trace --stats -d /tmp/kt -o bench -- $BENCH --timed-only > "$OUT/bench_under_trace.json" 2> "$OUT/trace.err"
DB=$(find /tmp/kt -name "*.db" | head -1)
python "$REPO/tools/rocprof_summary.py" "$DB" 200 > "$OUT/kernel_trace_summary.txt" 2>> "$OUT/trace.err"
head -14 "$OUT/kernel_trace_summary.txt"
