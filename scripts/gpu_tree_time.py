#!/usr/bin/env python3
"""Suffix-tree topology (sfx_lcp_intervals_dev) timed on the LCP array of a full-size config (development)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import _gen, suffix_amd
import _devlib
from suffix_amd import device as sdev
eng = _devlib.engine(); eng.require_device()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
for name, host in (("english", _gen.english_like(n)), ("dna", _gen.dna_fast(n, seed=0x5AF1C5 + 2))):
    text = torch.from_numpy(host).cuda()
    sa, lcp = sdev.build_sa_lcp(text)
    torch.cuda.synchronize()
    del sa
    out = sdev.lcp_intervals(lcp); torch.cuda.synchronize(); del out
    eng.profile(True); eng.profile_reset()
    t0 = time.perf_counter(); out = sdev.lcp_intervals(lcp); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    k = {r["name"]: round(r["total_ms"], 2) for r in eng.profile_report()}; eng.profile(False)
    nodes = int((out["node"] == torch.arange(n, device="cuda", dtype=torch.int32)).sum())
    print(json.dumps({"text": name, "n": n, "ms": round(dt * 1e3, 1), "kernel_ms": k, "internal_nodes": nodes,
                      "max_depth": int(lcp.max())}), flush=True)
    del text, lcp, out
    torch.cuda.empty_cache()
