#!/bin/bash
# round 5, call 14: per-iteration kernel durations of one frame on the trained headline field, reference chunking and regrouped
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/call14"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for cfg in device_exact k2_cap64; do
  rm -rf /tmp/kr_$cfg
  STEPS=${STEPS:-20000} FRAMES=6 CONFIG=$cfg timeout 300 rocprofv3 --kernel-trace -d /tmp/kr_$cfg -o r -- python $REPO/tools/render_trained.py > "$OUT/render_$cfg.out" 2> "$OUT/render_$cfg.err"
  DB=$(find /tmp/kr_$cfg -name "*.db" | head -1)
  python $REPO/tools/render_trace.py "$DB" > "$OUT/frame_iterations_$cfg.txt" 2>> "$OUT/render_$cfg.err"
  tail -3 "$OUT/render_$cfg.out"; cat "$OUT/frame_iterations_$cfg.txt"
done
