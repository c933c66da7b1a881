#!/bin/bash
# round 5, call 18: what bounds the frame marcher's early iterations -- SQ counters per dispatch (separate --pmc runs, --kernel-trace only)
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/call18"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; shift
  rm -rf /tmp/pmc_$name
  STEPS=20000 FRAMES=3 CONFIG=device_exact timeout 300 rocprofv3 --kernel-trace --pmc "$@" -f csv -d /tmp/pmc_$name -o r -- python $REPO/tools/render_trained.py > "$OUT/run_$name.out" 2> "$OUT/run_$name.err"
  python $REPO/tools/pmc_dispatches.py /tmp/pmc_$name render_march 11 > "$OUT/march_$name.txt" 2>> "$OUT/run_$name.err"
  cat "$OUT/march_$name.txt"
}
pass issue SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY
pass mem SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_ANY
