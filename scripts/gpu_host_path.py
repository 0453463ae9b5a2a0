#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point sfx_build_sa_u32 (what the Rust shim
calls: pageable text in, pageable Vec<u32> out; hipMalloc/hipFree of staging + workspace per
call) on BASELINE config 2.  Never bench.py's `value` -- reported beside it in DESIGN.md."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _gen  # noqa: E402
import suffix_amd  # noqa: E402
import _devlib

eng = _devlib.engine()
eng.require_device()
n = 100_000_000
text = _gen.dna(n, seed=0x5AF1C5 + 1)
sa = np.empty(n, dtype=np.uint32)
best = None
for _ in range(4):
    t0 = time.perf_counter()
    eng.check(eng.lib.sfx_build_sa_u32(text.ctypes.data, n, sa.ctypes.data), "sfx_build_sa_u32")
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
print(json.dumps({"entry": "sfx_build_sa_u32 (host pointers)", "n": n, "best_ms": round(best * 1e3, 1),
                  "MBps_pcie_inclusive": round(n / best / 1e6, 1), "sa_head": sa[:4].tolist()}))
