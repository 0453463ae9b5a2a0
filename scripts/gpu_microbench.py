#!/usr/bin/env python3
"""Memory-system micro-benchmarks of the engine (sfx_microbench) -> JSON lines.
Run on the GPU box:  python scripts/gpu_microbench.py > gpurun_out/microbench.jsonl"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import suffix_amd  # noqa: E402
import _devlib

eng = _devlib.engine()
eng.require_device()
E = eng
rows = []


def run(name, kind, nbytes, param=0, param2=0, reps=5):
    g = E.microbench(kind, nbytes, param, param2, reps)
    row = {"bench": name, "bytes": nbytes, "param": param, "param2": param2, "GBps": round(g, 1)}
    rows.append(row)
    print(json.dumps(row), flush=True)


run("copy16", E.MB_COPY, 1 << 30)
run("copy16", E.MB_COPY, 100_000_000)
for nb in (400_000_000, 4_000_000_000):
    run("scatter4", E.MB_SCATTER4, nb)
    run("gather4", E.MB_GATHER4, nb)
for nb in (100_000_000, 1_000_000_000):
    run("gather1", E.MB_GATHER1, nb)
for run_bytes in (8, 32, 64, 128, 256, 512, 1024, 4096):
    for aligned in (0, 1):
        run("runscatter8", E.MB_RUNSCATTER, 800_000_000, run_bytes, aligned)
