#!/bin/bash
# round 5, call 16: the frame marcher's block hops -- parity tests, A/B on the trained fields, per-iteration trace
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd "$REPO"
OUT="$REPO/gpurun_out/call16"; mkdir -p "$OUT"
timeout 700 python -m pytest tests/test_train_gpu.py tests/test_reference_surface_gpu.py tests/test_reference_files_gpu.py -x -q -m gpu -k "frame_loop or native_test_renderer or render" > "$OUT/tests.log" 2>&1
tail -5 "$OUT/tests.log"
timeout 300 python tools/frame_hops_ab.py lego 20000 > "$OUT/ab_lego.log" 2>&1; grep -v amdgpu.ids "$OUT/ab_lego.log" | tail -14
timeout 300 python tools/frame_hops_ab.py lego_hard 20000 > "$OUT/ab_lego_hard.log" 2>&1; grep -v amdgpu.ids "$OUT/ab_lego_hard.log" | tail -14
cd /tmp && export TMPDIR=/tmp
for cfg in device_exact; do
  rm -rf /tmp/kr_$cfg
  STEPS=20000 FRAMES=6 CONFIG=$cfg timeout 300 rocprofv3 --kernel-trace -d /tmp/kr_$cfg -o r -- python $REPO/tools/render_trained.py > "$OUT/render_$cfg.out" 2> "$OUT/render_$cfg.err"
  DB=$(find /tmp/kr_$cfg -name "*.db" | head -1)
  python $REPO/tools/render_trace.py "$DB" > "$OUT/frame_iterations_$cfg.txt" 2>> "$OUT/render_$cfg.err"
  cat "$OUT/frame_iterations_$cfg.txt"
done
