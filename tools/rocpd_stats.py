#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (ROCm 7.2 default output of
`rocprofv3 --kernel-trace --stats`) as the per-kernel stats table that gets committed
under profiles/.  usage: rocpd_stats.py results.db [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(grid_x), max(grid_y), max(workgroup_x), max(lds_size), max(vgpr_count), max(sgpr_count) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % | grid | wg | LDS B | VGPR | SGPR |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        name = r[0] if len(r[0]) < 90 else r[0][:87] + "..."
        lines.append("| %s | %d | %.3f | %.1f | %.1f | %.1f | %.1f | %dx%d | %d | %d | %d | %d |" % (
            name, r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot,
            r[6], r[7], r[8], r[9], r[10], r[11]))
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
