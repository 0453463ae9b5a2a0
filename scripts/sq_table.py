#!/usr/bin/env python3
"""One line per kernel from a pmc_summary.py CSV of SQ counters (scripts/gpu_sq_*.sh): waves, wave cycles, share of them spent
waiting, wave-instructions by kind.  usage: sq_table.py sq_summary.csv"""
import collections
import csv
import sys

rows = list(csv.reader(open(sys.argv[1])))
d = collections.defaultdict(dict)
n = {}
for r in rows[1:]:
    d[r[0]][r[1]] = float(r[3]) * float(r[2])           # totals over the dispatches of the run
    n[r[0]] = int(r[2])
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = v.get("SQ_WAVE_CYCLES", 0)
    if wc < 5e6:
        continue
    M = lambda c: v.get(c, 0) / 1e6
    print(f"{k[:64]:64s} x{n[k]:<4d} wavecyc {wc/1e6:8.0f}M wait {100*v.get('SQ_WAIT_ANY',0)/wc:3.0f}% valu {M('SQ_INSTS_VALU'):7.1f}M "
          f"salu {M('SQ_INSTS_SALU'):7.1f}M lds {M('SQ_INSTS_LDS'):6.1f}M vmem_rd {M('SQ_INSTS_VMEM_RD'):6.2f}M "
          f"valu_busy {100*4*v.get('SQ_ACTIVE_INST_VALU',0)/max(wc,1):3.0f}%")
