"""Summarise a rocprofv3 --pmc ... -f csv run: per kernel, mean counter values over the last N
dispatches of that kernel, plus mean duration from the kernel trace.  Usage:
  pmc_summary.py <dir-with-*_counter_collection.csv> [last_n]"""
import collections, csv, glob, os, re, sys

def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)[:60]

d = sys.argv[1]; last_n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
vals = collections.defaultdict(lambda: collections.defaultdict(dict))   # kernel -> dispatch -> counter -> value
with open(cc, newline="") as f:
    for row in csv.DictReader(f):
        k = short(row["Kernel_Name"]); disp = int(row["Dispatch_Id"])
        c = row["Counter_Name"]; v = float(row["Counter_Value"])
        vals[k][disp][c] = vals[k][disp].get(c, 0.0) + v
dur = collections.defaultdict(list)
if kt:
    with open(kt[0], newline="") as f:
        for row in csv.DictReader(f):
            dur[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
counters = sorted({c for k in vals for dd in vals[k].values() for c in dd})
print("%-58s %6s %9s " % ("kernel", "calls", "avg_us") + " ".join("%16s" % c for c in counters))
rows = []
for k, dd in vals.items():
    ids = sorted(dd)[-last_n:]
    mean = {c: sum(dd[i].get(c, 0.0) for i in ids) / len(ids) for c in counters}
    du = dur.get(k, [])
    avg = sum(du[-last_n:]) / max(1, len(du[-last_n:])) / 1e3 if du else 0.0
    rows.append((avg * len(dd), k, len(dd), avg, mean))
for _, k, n, avg, mean in sorted(rows, reverse=True)[:24]:
    print("%-58s %6d %9.1f " % (k, n, avg) + " ".join("%16.0f" % mean[c] for c in counters))
