"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db) into a small text table:
per-kernel calls / average / total over the steady-state window (the last N training steps, found
by counting launches of the grid Adam kernel; `skip_last` steps after the window are left out, e.g. the
module-path steps bench.py runs after the timed region).
Usage: rocprof_summary.py <results.db> [n_steps] [skip_last] > profiles/xyz.txt"""
import collections
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([A-Za-z_0-9]+?)(I[LbEi0-9]+E)?Ev?P", name)
    if m:
        return m.group(1) + (m.group(2) or "")
    name = re.sub(r"^void ", "", name)
    return re.sub(r"[(<].*", "", name)[:70]


def main():
    db = sqlite3.connect(sys.argv[1])
    n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    rows = list(db.execute("select name, start, end, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, lds_size from kernels order by start"))
    marks = [i for i, r in enumerate(rows) if "adam_kernelILb0" in r[0] or "adam_field" in r[0]]          # (adam_field_kernel / adam_field_merge_kernel / adam_field_pieces_kernel: one per step)
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    if skip:
        rows = rows[:marks[-skip - 1] + 1]; marks = marks[:-skip]
    sel = rows[marks[-n_steps]:] if len(marks) >= n_steps else rows
    agg = collections.OrderedDict()
    for name, s, e, gx, wx, vg, ag, lds in sel:
        k = short(name)
        a = agg.setdefault(k, [0, 0.0, gx, wx, vg, ag, lds])
        a[0] += 1; a[1] += e - s
    busy = sum(a[1] for a in agg.values()); span = sel[-1][2] - sel[0][1]
    print("# steady-state window: last %d steps, %d dispatches, GPU busy %.2f ms, wall span %.2f ms" % (n_steps, len(sel), busy / 1e6, span / 1e6))
    print("%-44s %6s %10s %10s %6s %8s %5s %5s %7s" % ("kernel", "calls", "avg_us", "total_ms", "pct", "grid", "wg", "vgpr", "lds"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print("%-44s %6d %10.1f %10.3f %6.1f %8d %5d %5d %7d" % (k[:44], a[0], a[1] / a[0] / 1e3, a[1] / 1e6, 100 * a[1] / busy, a[2], a[3], a[4] + a[5], a[6]))


if __name__ == "__main__":
    main()
