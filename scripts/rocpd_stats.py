#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd SQLite result (`*_results.db`, the default output of
`rocprofv3 --kernel-trace --stats`) into the per-kernel summary CSV we commit
under profiles/.   usage: rocpd_stats.py results.db out.csv"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
    for r in rows:
        w.writerow([r[0], r[1], f"{r[2]:.3f}", f"{r[3]:.3f}", f"{r[4]:.3f}"])
print(f"{len(rows)} kernels -> {sys.argv[2]}")
