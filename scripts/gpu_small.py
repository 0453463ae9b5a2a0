#!/usr/bin/env python3
"""Latency of the device-resident and host-pointer builds on the reference's own benchmark
inputs (tests/bench.rs: AP009048 10 KB / 100 KB DNA, README.md:111-116) -- launch-bound sizes."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import suffix_amd  # noqa: E402
import _devlib
from suffix_amd import device as sdev  # noqa: E402

eng = _devlib.engine()
eng.require_device()
z = np.load(os.path.join(ROOT, "tests", "golden", "fasta_fixtures.npz"))
for name in ("AP009048_10000", "AP009048_100000"):
    host = np.ascontiguousarray(z[name])
    n = host.size
    text = torch.from_numpy(host).cuda()
    ws = sdev.sa_workspace(n, text.device)
    sa = torch.empty(n, dtype=torch.int32, device="cuda")
    for _ in range(3):
        sdev.build_sa(text, out=sa, workspace=ws)
    torch.cuda.synchronize()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        sdev.build_sa(text, out=sa, workspace=ws)
    torch.cuda.synchronize()
    dev_us = (time.perf_counter() - t0) / reps * 1e6
    out = np.empty(n, dtype=np.uint32)
    for _ in range(3):
        eng.check(eng.lib.sfx_build_sa_u32(host.ctypes.data, n, out.ctypes.data), "host")
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.check(eng.lib.sfx_build_sa_u32(host.ctypes.data, n, out.ctypes.data), "host")
    host_us = (time.perf_counter() - t0) / reps * 1e6
    eng.profile(True); eng.profile_reset(); sdev.build_sa(text, out=sa, workspace=ws); torch.cuda.synchronize()
    rep = eng.profile_report(); eng.profile(False)
    print(json.dumps({"input": name, "n": int(n), "device_resident_us": round(dev_us, 1), "host_pointers_us": round(host_us, 1),
                      "kernel_launches": sum(r["launches"] for r in rep), "kernel_us_sum": round(sum(r["total_ms"] for r in rep) * 1e3, 1),
                      "reference_README_ns_per_iter": {"AP009048_10000": 712938, "AP009048_100000": 7514327}[name]}))
