"""Per-DISPATCH counter values of the kernels whose name contains <pattern>, last N dispatches, from a rocprofv3 --pmc ... -f csv run
(tools/pmc_summary.py gives per-kernel means).  Usage: pmc_dispatches.py <dir> <pattern> [last_n]"""
import collections, csv, glob, os, re, sys

d, pat = sys.argv[1], sys.argv[2]
last_n = int(sys.argv[3]) if len(sys.argv) > 3 else 12
cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
vals = collections.OrderedDict()
for row in csv.DictReader(open(cc, newline="")):
    if pat not in row["Kernel_Name"]:
        continue
    disp = int(row["Dispatch_Id"])
    name = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"]); name = re.sub(r"^void ", "", name); name = re.sub(r"\(.*", "", name)
    e = vals.setdefault(disp, {"name": name, "grid": row.get("Grid_Size", "?")})
    e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
counters = sorted({c for e in vals.values() for c in e if c not in ("name", "grid")})
print("%-8s %-40s %9s " % ("dispatch", "kernel", "grid") + " ".join("%18s" % c for c in counters))
for disp in sorted(vals)[-last_n:]:
    e = vals[disp]
    print("%-8d %-40s %9s " % (disp, e["name"][:40], e["grid"]) + " ".join("%18.0f" % e.get(c, 0.0) for c in counters))
