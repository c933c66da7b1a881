#!/bin/bash
# Round profile of the default bench on an MI355X box: kernel trace + the PMC passes the roofline block needs.
#   gpurun --timeout 900 -- 'bash tools/profile_bench.sh r02'
# Writes gpurun_out/prof_<tag>/: kernel_trace_summary.txt, bench_under_trace.json, pmc_*.txt (per-kernel counter means),
# pmc_traffic.json (what bench.py reads back from profiles/), counters_available.txt.  Copy what should be judged to profiles/.
# Counters are collected in their own runs with --kernel-trace only (no other trace domain next to --pmc).
TAG=${1:-r02}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 20 --warmup 5 --no-render --no-cpu-baseline --no-api --no-full-run --no-secondary"
rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|Counter)?\s*:?\s*(SQ_|TCC_|TCP_|GRBM_|FETCH|WRITE)" | head -400 > "$OUT/counters_available.txt" 2>&1
# 1. kernel trace of the driver's command shape (timed windows last in the trace)
rm -rf /tmp/kt && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o bench -- $BENCH --timed-only > "$OUT/bench_under_trace.json" 2> "$OUT/trace.err"
DB=$(find /tmp/kt -name "*.db" | head -1)
python "$REPO/tools/rocprof_summary.py" "$DB" 200 > "$OUT/kernel_trace_summary.txt" 2>> "$OUT/trace.err"
# 2. PMC passes (the run ends with bench.py's ROOFLINE_STEPS stage-timed steps: counters are averaged over those launches)
pass() {   # name, counters...
  local name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout 400 rocprofv3 --kernel-trace --pmc "$@" -f csv -d /tmp/pmc_$name -o bench -- $BENCH > "$OUT/bench_pmc_$name.json" 2> "$OUT/pmc_$name.err"
  python "$REPO/tools/pmc_summary.py" /tmp/pmc_$name 20 > "$OUT/pmc_$name.txt" 2>> "$OUT/pmc_$name.err"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES
pass sqwait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pass l2 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
python "$REPO/tools/pmc_traffic.py" "$OUT" > "$OUT/pmc_traffic.json" 2> "$OUT/pmc_traffic.err"
ls -la "$OUT"
