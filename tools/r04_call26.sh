#!/bin/bash
# Round 4, call 26: kernel trace of the late tree (driver's command shape, timed windows last in the trace).
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/prof_r04late"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 20 --warmup 5 --no-render --no-cpu-baseline --no-api --no-full-run"
rm -rf /tmp/kt && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o bench -- $BENCH --timed-only > "$OUT/bench_under_trace.json" 2> "$OUT/trace.err"
DB=$(find /tmp/kt -name "*.db" | head -1)
python "$REPO/tools/rocprof_summary.py" "$DB" 200 > "$OUT/kernel_trace_summary.txt" 2>> "$OUT/trace.err"
head -30 "$OUT/kernel_trace_summary.txt" | cut -c1-150
