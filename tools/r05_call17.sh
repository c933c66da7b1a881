#!/bin/bash
# round 5, call 17: wave-per-ray late iterations of the frame loop -- parity tests, crossover sweep on the trained fields, per-iteration trace
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd "$REPO"
OUT="$REPO/gpurun_out/call17"; mkdir -p "$OUT"
timeout 700 python -m pytest tests/test_train_gpu.py tests/test_reference_surface_gpu.py tests/test_reference_files_gpu.py -x -q -m gpu -k "frame_loop or native_test_renderer or render" > "$OUT/tests.log" 2>&1
tail -5 "$OUT/tests.log"
timeout 300 python tools/frame_wave_ab.py lego 20000 > "$OUT/wave_lego.log" 2>&1; grep -v amdgpu.ids "$OUT/wave_lego.log" | tail -19
timeout 300 python tools/frame_wave_ab.py lego_hard 20000 > "$OUT/wave_lego_hard.log" 2>&1; grep -v amdgpu.ids "$OUT/wave_lego_hard.log" | tail -13
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kr_x
STEPS=20000 FRAMES=6 CONFIG=device_exact timeout 300 rocprofv3 --kernel-trace -d /tmp/kr_x -o r -- python $REPO/tools/render_trained.py > "$OUT/render.out" 2> "$OUT/render.err"
DB=$(find /tmp/kr_x -name "*.db" | head -1)
python $REPO/tools/render_trace.py "$DB" > "$OUT/frame_iterations.txt" 2>> "$OUT/render.err"
cat "$OUT/frame_iterations.txt"
