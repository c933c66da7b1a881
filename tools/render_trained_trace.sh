#!/bin/bash
# Kernel trace of rendered frames on the TRAINED field (30 000 steps), both chunkings:
#   gpurun --timeout 600 -- 'bash tools/render_trained_trace.sh r04'
# writes gpurun_out/<tag>_render_trace.json (what bench.py's roofline_render.profile reads from profiles/) and .txt
TAG=${1:-r04}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
FRAMES=${FRAMES:-20}
for cfg in k2_cap64 device_exact; do
  rm -rf /tmp/kr_$cfg
  STEPS=${STEPS:-30000} FRAMES=$FRAMES CONFIG=$cfg timeout 400 rocprofv3 --kernel-trace -d /tmp/kr_$cfg -o r -- python $REPO/tools/render_trained.py > "$OUT/${TAG}_render_$cfg.out" 2> "$OUT/${TAG}_render_$cfg.err"
  DB=$(find /tmp/kr_$cfg -name "*.db" | head -1)
  python $REPO/tools/render_trace_summary.py "$DB" $FRAMES "$OUT/${TAG}_render_$cfg.out" $cfg > "$OUT/${TAG}_render_trace_$cfg.json" 2>> "$OUT/${TAG}_render_$cfg.err"
done
python - "$OUT" "$TAG" <<'PY'
import json, sys
out, tag = sys.argv[1], sys.argv[2]
a = json.load(open("%s/%s_render_trace_k2_cap64.json" % (out, tag)))
b = json.load(open("%s/%s_render_trace_device_exact.json" % (out, tag)))
a["reference_chunking"] = b
json.dump(a, open("%s/%s_render_trace.json" % (out, tag), "w"), indent=1)
print(json.dumps(a, indent=1)[:3000])
PY
