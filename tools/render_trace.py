"""Per-iteration kernel durations of the last rendered frame in a rocprofv3 kernel trace (rocpd .db)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,start,end,grid_x from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'render_begin' in r[0]]
st = idx[-1]; t0 = rows[st][1]; it = 0; line = []
keys = ('render_begin', 'render_march', 'hashgrid_fwd', 'mlp_fwd_kernelILi32ELi1', 'mlp_fwd_kernelILi32ELi2', 'render_composite', 'render_finish', 'field_fwd_kernel')
tot = {}
for r in rows[st:]:
    k = next((key for key in keys if key in r[0]), None)
    if k == 'render_march':
        if line: print(it, ' '.join(line))
        line = []; it += 1
        line.append("t=%.2fms" % ((r[1] - t0) / 1e6))
    if k:
        line.append("%s[%d]=%.0f" % (k[7:15], r[3], (r[2] - r[1]) / 1e3))
        tot[k] = tot.get(k, 0) + (r[2] - r[1]) / 1e6
print(it, ' '.join(line))
print("frame total %.3f ms" % ((rows[-1][2] - t0) / 1e6), {k: round(v, 3) for k, v in tot.items()})
