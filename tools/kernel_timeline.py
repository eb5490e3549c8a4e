#!/usr/bin/env python3
"""Per-dispatch durations, in launch order, of the kernels whose name contains a substring (rocprofv3 rocpd .db).
usage: kernel_timeline.py <results.db> <substring> [<substring> ...]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
t0 = rows[0][1] if rows else 0
for name, s, e in rows:
    if any(k in name for k in sys.argv[2:]):
        print("%10.3f ms  +%8.3f ms  %s" % ((s - t0) / 1e6, (e - s) / 1e6, name[:60]))
