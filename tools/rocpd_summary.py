#!/usr/bin/env python3
"""Turn rocprofv3's rocpd SQLite output (<name>_results.db) into the text summaries kept under
profiles/: per-kernel time (like `--stats`) and per-kernel PMC counter averages.

usage: rocpd_summary.py <results.db> [--pmc] [--filter k_]"""
import sqlite3
import sys


def short(name, n=96):
    name = name.replace("void ", "")
    return name if len(name) <= n else name[:n - 3] + "..."


def main():
    path = sys.argv[1]
    pmc = "--pmc" in sys.argv
    db = sqlite3.connect(path)
    if not pmc:
        rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                          "from kernels group by name order by 3 desc").fetchall()
        tot = sum(r[2] for r in rows)
        print("%-98s %6s %14s %14s %14s %14s %7s" % ("KERNEL", "CALLS", "TOTAL_ns", "AVG_ns", "MIN_ns", "MAX_ns", "PCT"))
        for r in rows[:25]:
            print("%-98s %6d %14d %14.0f %14d %14d %6.2f%%" % (short(r[0]), r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
    else:
        rows = db.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration), max(vgpr_count), "
                          "max(lds_block_size), max(scratch_size) from counters_collection "
                          "group by kernel_name, counter_name order by 4 desc").fetchall()
        print("%-98s %-12s %6s %18s %12s %6s %8s %8s" % ("KERNEL", "COUNTER", "CALLS", "AVG_VALUE", "AVG_ns", "VGPR", "LDS", "SCRATCH"))
        for r in rows[:40]:
            print("%-98s %-12s %6d %18.3f %12.0f %6d %8d %8d" % (short(r[0]), r[1], r[2], r[3], r[4] or 0, r[5] or 0, r[6] or 0, r[7] or 0))


if __name__ == "__main__":
    main()
