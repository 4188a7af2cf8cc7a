#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) as a per-kernel stats table
(the same columns `--stats` prints: calls, total, average, min, max, percentage)."""
import sqlite3
import sys


def main(db, out=sys.stdout):
    con = sqlite3.connect(db)
    rows = con.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(workgroup_x), max(grid_x) from kernels group by name "
        "order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out.write("| kernel | calls | total_ms | avg_us | min_us | max_us | pct | vgpr | sgpr | lds_B | wg | grid_x |\n")
    out.write("|---|---|---|---|---|---|---|---|---|---|---|---|\n")
    for n, c, s, a, mn, mx, vg, sg, lds, wg, gx in rows:
        name = n.split("(")[0].replace("void ", "").split("<")[0]
        out.write("| %s | %d | %.3f | %.1f | %.1f | %.1f | %.2f | %s | %s | %s | %s | %s |\n"
                  % (name, c, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / total, vg, sg, lds, wg, gx))


if __name__ == "__main__":
    main(sys.argv[1])
