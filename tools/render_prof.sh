#!/bin/bash
# Scratch: kernel-trace summary of rendered frames (after 540 training steps), k4_cap64 (bench setting) and reference chunking.
cd /tmp && export TMPDIR=/tmp
for cfg in k4_cap64 device_exact; do
  rm -rf /tmp/kr_$cfg
  STEPS=540 FRAMES=8 CONFIGS=$cfg timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kr_$cfg -o r -- python /root/repo/tools/profile_render.py > /root/repo/gpurun_out/render_$cfg.txt 2>/dev/null
  DB=$(find /tmp/kr_$cfg -name "*.db" | head -1)
  python - "$DB" >> /root/repo/gpurun_out/render_$cfg.txt <<'PY'
import sqlite3, sys, collections, re
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'render_begin' in r[0]]
sel = rows[idx[-8]:]           # the 8 timed frames + the warm-up one before are the last 9 render_begin's; take the last 8
agg = collections.OrderedDict()
for n, s, e in sel:
    k = re.sub(r"\(anonymous namespace\)::", "", n); k = re.sub(r"^void ", "", k); k = re.sub(r"[(<].*", "", k)[:48]
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([A-Za-z_0-9]+)", n)
    if m: k = m.group(1)[:48]
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += e - s
busy = sum(a[1] for a in agg.values()); span = sel[-1][2] - sel[0][1]
print("# 8 frames: GPU busy %.2f ms, span %.2f ms -> %.2f ms/frame" % (busy / 1e6, span / 1e6, span / 8e6))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%-48s calls/frame %6.1f  avg_us %8.1f  ms/frame %7.3f  %5.1f%%" % (k, a[0] / 8, a[1] / a[0] / 1e3, a[1] / 8e6, 100 * a[1] / busy))
PY
done
