#!/usr/bin/env python3
"""LCP on large texts: per-kernel times of the Phi/PLCP path and of the direct path.
   gpurun --timeout 900 -- 'python scripts/gpu_lcp_prof.py [dna1g eng400m utf400m]'"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _gen  # noqa: E402
import suffix_amd  # noqa: E402
import _devlib
from suffix_amd import device as sdev  # noqa: E402

eng = _devlib.engine()
dev = torch.device("cuda", 0)
GENS = {
    "dna1g": lambda: _gen.dna(1_000_000_000, seed=7),
    "dna100m": lambda: _gen.dna(100_000_000),
    "eng400m": lambda: _gen.english_like(400_000_000, seed=3),
    "utf400m": lambda: _gen.utf8_mixed(400_000_000),
    "eng1g": lambda: _gen.english_like(1_000_000_000),
    "utf1g": lambda: _gen.utf8_mixed(1_000_000_000),
}
for name in (sys.argv[1:] or ["dna1g", "eng400m"]):
    t = torch.from_numpy(GENS[name]()).to(dev)
    n = t.numel()
    sa = sdev.build_sa(t)
    torch.cuda.synchronize()
    out = {"text": name, "n": n, "env": {a: b for a, b in os.environ.items() if a.startswith("SFX_")}}
    for label, fn in (("lcp", lambda: sdev.build_lcp(t, sa)), ("lcp_direct_slice", lambda: sdev.build_lcp_range(t, sa))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = fn()
        torch.cuda.synchronize()
        out[label + "_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        eng.profile(True); eng.profile_reset()
        fn(); torch.cuda.synchronize()
        out[label + "_kernels"] = {r["name"]: round(r["total_ms"], 2) for r in eng.profile_report()}
        eng.profile(False)
        out[label + "_sum"] = int(res.view(torch.int32).to(torch.int64).sum())
        del res
    out["same"] = out["lcp_sum"] == out["lcp_direct_slice_sum"]
    print(json.dumps(out), flush=True)
    del t, sa
