"""Kernel timeline of one bench step from a rocprofv3 rocpd database: tools/timeline.py results.db [step]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end,stream_id,grid_x,grid_y,workgroup_x from kernels order by start").fetchall()
def short(n):
    m = re.search(r'(den_\w+|num_\w+|rescale_kernel|__amd\w+|\w+_kernel\w*)', n)
    return (m.group(1) if m else n)[:30]
idx = [i for i, r in enumerate(rows) if 'num_prep' in r[0]]
if not idx: idx = [i for i, r in enumerate(rows) if 'zero_words' in r[0]]      # a denominator-only workload
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
i0 = idx[k]
i1 = idx[k + 1] if k + 1 < len(idx) else len(rows)
t0 = rows[i0][1]
for r in rows[max(0, i0 - 3):i1]:
    print("%-30s %9.1f %9.1f %8.1f  s%s grid %dx%d" % (short(r[0]), (r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[4] // max(1, r[6]), r[5]))
