"""rocprofv3 kernel trace (rocpd .db) of tools/render_trained.py -> JSON: per-kernel share of a rendered frame on the trained field.
The frames are the launches from the (FRAMES)-th last `render_begin` kernel on (the warm-up frame before them is left out).
Usage: render_trace_summary.py <results.db> <frames> <stdout of render_trained.py> <config>"""
import collections, json, re, sqlite3, sys

db = sqlite3.connect(sys.argv[1]); frames = int(sys.argv[2]); cfg = sys.argv[4]
run = None
for ln in open(sys.argv[3]):
    if ln.startswith("{"):
        run = json.loads(ln)
rows = list(db.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "render_begin" in r[0]]
sel = rows[idx[-frames]:]


def short(n):
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([A-Za-z_0-9]+?)(I[LbEi0-9]+E)?Ev?P", n)
    if m:
        return m.group(1)
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return re.sub(r"[(<].*", "", n)[:48]


agg = collections.OrderedDict()
for n, s, e in sel:
    a = agg.setdefault(short(n), [0, 0.0]); a[0] += 1; a[1] += e - s
busy = sum(a[1] for a in agg.values()); span = sel[-1][2] - sel[0][1]
kern = [{"kernel": k, "launches_per_frame": a[0] / frames, "avg_us": a[1] / a[0] / 1e3, "ms_per_frame": a[1] / frames / 1e6, "share_of_busy": a[1] / busy}
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]]
print(json.dumps({"config": cfg, "frames": frames, "gpu_busy_ms_per_frame": busy / frames / 1e6, "span_ms_per_frame": span / frames / 1e6,
                  "dominant_kernel": kern[0]["kernel"], "kernels": kern, "run": run,
                  "what": "rocprofv3 --kernel-trace of tools/render_trained.py: every kernel between the first timed frame's render_begin and the end of the last frame (ray generation by torch included, as in the FPS protocol)"}))
