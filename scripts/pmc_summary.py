#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per (kernel, counter) mean value per dispatch.
usage: pmc_summary.py <dir-with-*counter_collection.csv> [...]   -> CSV on stdout"""
import csv
import glob
import os
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0.0, 0])
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                short = name.split("(")[0].replace("void ", "")
                key = (short, row.get("Counter_Name", ""))
                acc[key][0] += float(row.get("Counter_Value", 0) or 0)
                acc[key][1] += 1
w = csv.writer(sys.stdout)
w.writerow(["Kernel", "Counter", "Dispatches", "MeanPerDispatch"])
for (k, c), (tot, cnt) in sorted(acc.items()):
    w.writerow([k, c, cnt, f"{tot / max(cnt, 1):.1f}"])
