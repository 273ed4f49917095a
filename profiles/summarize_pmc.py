#!/usr/bin/env python3
"""Per-kernel sums of PMC counters from a rocprofv3 rocpd SQLite database (--pmc run)."""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    scol = [r[1] for r in c.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in scol else scol[-1]
    pcols = [r[1] for r in c.execute("pragma table_info(rocpd_pmc_event)")]
    icols = [r[1] for r in c.execute("pragma table_info(rocpd_info_pmc)")]
    ecols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
    # rocpd_pmc_event(event_id -> rocpd_event.id), rocpd_kernel_dispatch(event_id)
    q = ("select s.%s, p.name, count(distinct d.id), sum(e.value) from rocpd_pmc_event e "
         "join rocpd_info_pmc p on e.pmc_id = p.id "
         "join rocpd_kernel_dispatch d on d.event_id = e.event_id "
         "join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
         "group by s.%s, p.name order by 4 desc" % (name_col, name_col))
    try:
        rows = list(c.execute(q))
    except Exception as ex:  # schema differences: show what exists
        print("query failed:", ex)
        print("pmc_event cols", pcols, "info_pmc cols", icols, "dispatch cols", ecols)
        return
    print("%-70s %-34s %8s %20s" % ("kernel", "counter", "calls", "sum"))
    for name, cn, n, v in rows:
        if v is None or v == 0:
            continue
        print("%-70s %-34s %8d %20.0f" % (name[:70], cn, n, v))


if __name__ == "__main__":
    main(sys.argv[1])
