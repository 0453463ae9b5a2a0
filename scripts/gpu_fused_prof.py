#!/usr/bin/env python3
"""SuffixTable::new + lcp_lens as one engine call on the 1 GB configs: per-kernel times of the fused build next to the
plain SA build (what does the LCP cost where?).   gpurun -- 'python scripts/gpu_fused_prof.py [c3 c5 dup]'"""
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
N = int(os.environ.get("SFX_N", "1000000000"))
GENS = {"c3": lambda: _gen.english_like(N), "c5": lambda: _gen.utf8_mixed(N), "dup": lambda: _gen.near_duplicates(N),
        "dna": lambda: _gen.dna_fast(N, seed=7)}
for name in (sys.argv[1:] or ["c3"]):
    t = torch.from_numpy(GENS[name]()).to(dev)
    n = t.numel()
    ws = sdev.sa_lcp_workspace(n, dev)
    sa = torch.empty(n, dtype=torch.int32, device=dev)
    lcp = torch.empty(n, dtype=torch.int32, device=dev)
    out = {"text": name, "n": n}
    for label, fn in (("sa", lambda: sdev.build_sa(t, out=sa, workspace=ws)),
                      ("sa_lcp", lambda: sdev.build_sa_lcp(t, out_sa=sa, out_lcp=lcp, workspace=ws))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(); torch.cuda.synchronize()
        out[label + "_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        eng.profile(True); eng.profile_reset()
        fn(); torch.cuda.synchronize()
        out[label + "_kernels"] = {r["name"]: round(r["total_ms"], 2) for r in sorted(eng.profile_report(), key=lambda r: -r["total_ms"]) if r["total_ms"] >= 0.3}
        eng.profile(False)
    ks, kf = out["sa_kernels"], out["sa_lcp_kernels"]
    out["delta_ms"] = {k: round(kf.get(k, 0) - ks.get(k, 0), 2) for k in sorted(set(ks) | set(kf)) if abs(kf.get(k, 0) - ks.get(k, 0)) >= 0.2}
    print(json.dumps(out), flush=True)
    del t, sa, lcp, ws
    torch.cuda.empty_cache()
