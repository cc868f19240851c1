#!/usr/bin/env python3
"""Per-kernel average of every PMC counter found in rocprofv3 rocpd databases.
usage: pmc_report.py db1 [db2 ...]"""
import sqlite3, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
    q = "select kernel_name, counter_name, value, dispatch_id from counters_collection" if "kernel_name" in cols else None
    if q is None:
        print(path, "columns:", cols); continue
    per = defaultdict(float)
    names = {}
    for k, c, v, d in db.execute(q):
        per[(k, c, d)] += v
    for (k, c, d), v in per.items():
        acc[k][c].append(v)
for k, cs in acc.items():
    if "den_" not in k and "num_kernel" not in k: continue
    print(k[:100])
    for c, vs in sorted(cs.items()):
        print("   %-28s n=%-3d avg=%.4g" % (c, len(vs), sum(vs) / len(vs)))
