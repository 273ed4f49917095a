#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg/min/max duration) from a rocprofv3 rocpd SQLite database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes NAME_results.db on ROCm 7.2)."""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
    scol = [r[1] for r in c.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else scol[-1])
    q = ("select s.%s, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
         "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
         "group by s.%s order by 3 desc" % (name_col, name_col))
    rows = list(c.execute(q))
    tot = sum(r[2] for r in rows) or 1
    print("%-90s %8s %14s %14s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
    for name, n, t, a, mn, mx in rows:
        nm = name if len(name) <= 90 else name[:87] + "..."
        print("%-90s %8d %14d %14.0f %12d %12d %6.2f%%" % (nm, n, t, a, mn, mx, 100.0 * t / tot))


if __name__ == "__main__":
    main(sys.argv[1])
