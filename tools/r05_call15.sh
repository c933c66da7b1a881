#!/bin/bash
# round 5, call 15: frame loop with block bits in LDS + small first tile + vector begin: parity tests, then the per-iteration trace again
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd "$REPO"
mkdir -p gpurun_out/call15
timeout 600 python -m pytest tests/test_train_gpu.py tests/test_reference_surface_gpu.py tests/test_reference_files_gpu.py -x -q -m gpu -k "frame_loop or native_test_renderer or render" > gpurun_out/call15/tests.log 2>&1
tail -5 gpurun_out/call15/tests.log
OUT="$REPO/gpurun_out/call15"
cd /tmp && export TMPDIR=/tmp
for cfg in device_exact k2_cap64; do
  rm -rf /tmp/kr_$cfg
  STEPS=${STEPS:-20000} FRAMES=6 CONFIG=$cfg timeout 300 rocprofv3 --kernel-trace -d /tmp/kr_$cfg -o r -- python $REPO/tools/render_trained.py > "$OUT/render_$cfg.out" 2> "$OUT/render_$cfg.err"
  DB=$(find /tmp/kr_$cfg -name "*.db" | head -1)
  python $REPO/tools/render_trace.py "$DB" > "$OUT/frame_iterations_$cfg.txt" 2>> "$OUT/render_$cfg.err"
  tail -1 "$OUT/render_$cfg.out" | cut -c1-120; cat "$OUT/frame_iterations_$cfg.txt"
done
# untraced FPS, 40 poses
for cfg in device_exact k2_cap64; do STEPS=20000 FRAMES=40 CONFIG=$cfg timeout 200 python $REPO/tools/render_trained.py 2>/dev/null | tail -1 | cut -c1-140; done
