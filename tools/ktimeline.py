#!/usr/bin/env python3
"""Kernel timeline (start offset, duration, gap to the previous end) from a rocprofv3 rocpd database."""
import sqlite3
import sys
c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
scol = [r[1] for r in c.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else scol[-1])
rows = list(c.execute("select s.%s, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start" % name_col))
t0 = rows[0][1] if rows else 0
prev_end = None
for name, s, e in rows:
    if pat and pat not in name:
        continue
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%10.3f ms  dur %9.3f ms  gap %8.1f us  %s" % ((s - t0) / 1e6, (e - s) / 1e6, gap, name[:70]))
    prev_end = max(prev_end or e, e)
