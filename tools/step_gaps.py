"""Scratch: where a training step's wall time goes that no stage accounts for (VERDICT r04 "What's weak" 9).
From a rocprofv3 --kernel-trace run (rocpd .db) of `bench.py --timed-only`: over the last N steps (step = from the end of one
grid-Adam launch to the end of the next), per step
  wall            end-to-end
  main_busy       sum of kernel durations on the main stream's queue (the queue the Adam launch is on)
  main_idle       wall - main_busy, split into: idle while an occupancy-update kernel or the march of the NEXT batch is what the main
                  stream waits for, and launch gaps (< 12 us holes between consecutive main-queue kernels)
and the same averaged separately over steps with / without an occupancy update.
Usage: step_gaps.py <results.db> [n_steps=64] > profiles/r05_step_gaps.txt"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([A-Za-z_0-9]+?)(I[LbEi0-9]+E)?Ev?P", name)
    if m:
        return m.group(1)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"[(<].*", "", name)[:60]


db = sqlite3.connect(sys.argv[1])
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else "stream_id"
rows = [(short(n), s, e, q) for n, s, e, q in db.execute("select name, start, end, %s from kernels order by start" % qcol)]
marks = [i for i, r in enumerate(rows) if r[0].startswith("adam_field")]
marks = marks[-(n_steps + 1):]
mainq = rows[marks[-1]][3]
OCC = ("occ_", "density_grid_update", "packbits", "fillBuffer")
steps = []
for a, b in zip(marks[:-1], marks[1:]):
    t0, t1 = rows[a][2], rows[b][2]
    sel = [r for r in rows[a + 1:b + 1]]
    main = [r for r in sel if r[3] == mainq]
    side = [r for r in sel if r[3] != mainq]
    busy = sum(r[2] - r[1] for r in main)
    has_occ = any(r[0].startswith(OCC) for r in sel)
    occ_busy = 0
    if has_occ:          # everything on the main queue from the update's first kernel to its packbits (its hash / density forward included)
        names = [r[0] for r in main]
        first = next(k for k, n in enumerate(names) if n.startswith(OCC))
        last = max(k for k, n in enumerate(names) if n.startswith(OCC))
        occ_busy = sum(r[2] - r[1] for r in main[first:last + 1])
    # holes on the main queue
    holes, prev = [], t0
    for r in main:
        if r[1] > prev:
            holes.append((prev, r[1]))
        prev = max(prev, r[2])
    small = sum(h[1] - h[0] for h in holes if h[1] - h[0] < 12e3)
    big = [(h[0], h[1]) for h in holes if h[1] - h[0] >= 12e3]
    # what runs on the other queues during the big holes
    cover = {}
    for h0, h1 in big:
        for r in side:
            ov = min(h1, r[2]) - max(h0, r[1])
            if ov > 0:
                cover[r[0]] = cover.get(r[0], 0) + ov
    steps.append(dict(wall=t1 - t0, busy=busy, small=small, big=sum(h[1] - h[0] for h in big), occ=has_occ, occ_busy=occ_busy, cover=cover,
                      n_main=len(main), side_busy=sum(r[2] - r[1] for r in side)))


def avg(sel, key):
    return sum(s[key] for s in sel) / max(len(sel), 1) / 1e3


print("# %d steps; main queue = %s.  us per step (mean)" % (len(steps), mainq))
print("%-28s %6s %9s %10s %12s %12s %10s %8s" % ("steps", "count", "wall", "main_busy", "holes<12us", "holes>=12us", "occ_kernels", "launches"))
for label, sel in (("all", steps), ("without occupancy update", [s for s in steps if not s["occ"]]), ("with occupancy update", [s for s in steps if s["occ"]])):
    print("%-28s %6d %9.1f %10.1f %12.1f %12.1f %10.1f %8.1f" % (label, len(sel), avg(sel, "wall"), avg(sel, "busy"), avg(sel, "small"), avg(sel, "big"),
                                                              avg(sel, "occ_busy"), sum(s["n_main"] for s in sel) / max(len(sel), 1)))
tot = {}
for s in steps:
    for k, v in s["cover"].items():
        tot[k] = tot.get(k, 0) + v
print("# kernels on the other queues that run during the main queue's holes >= 12 us (us per step, mean over ALL steps):")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:8]:
    print("  %-44s %8.1f" % (k, v / len(steps) / 1e3))
wall_all = avg(steps, "wall")
print("# amortised: occupancy-update steps add %.1f us per step on average (wall of update steps - wall of plain steps) / %d" % (
    (avg([s for s in steps if s["occ"]], "wall") - avg([s for s in steps if not s["occ"]], "wall")) * len([s for s in steps if s["occ"]]) / max(len(steps), 1), 1))
