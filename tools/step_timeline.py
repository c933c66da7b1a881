"""Scratch: kernel timeline of the last training steps from a rocprofv3 --kernel-trace run (rocpd .db): start (us, relative to
the first kernel shown), duration, queue, name.  Usage: step_timeline.py <results.db> [n_steps]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([A-Za-z_0-9]+?)(I[LbEi0-9]+E)?Ev?P", name)
    if m:
        return m.group(1) + (m.group(2) or "")
    name = re.sub(r"^void ", "", name)
    return re.sub(r"[(<].*", "", name)[:60]


db = sqlite3.connect(sys.argv[1])
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(db.execute("select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else "")))
marks = [i for i, r in enumerate(rows) if "adam_field_kernel" in r[0]]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if len(sys.argv) > 4:            # window around the last launch of a named kernel (e.g. occ_scan_kernel: an occupancy update)
    hit = [i for i, r in enumerate(rows) if sys.argv[4] in r[0]][-1]
    k = max(j for j in range(len(marks)) if marks[j] < hit)
    first, last = marks[k - 1] + 1, marks[min(k + 1, len(marks) - 1)]
else:
    last = marks[-1 - skip]
    first = marks[-1 - skip - n_steps] + 1
t0 = rows[first][1]
prev_end = t0
for r in rows[first:last + 1]:
    gap = (r[1] - prev_end) / 1e3
    print("%9.1f  +%6.1f us  q=%-4s %-50s%s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3] if qcol else "-", short(r[0]), ("   [idle %.1f us before]" % gap) if gap > 3 else ""))
    prev_end = max(prev_end, r[2])
