#!/usr/bin/env python3
"""Diagnostic: fused SA + LCP on a skewed 4-letter text (development)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import _gen, oracle, suffix_amd
import _devlib
from suffix_amd import device as sdev
oracle.build()
eng = _devlib.engine(); eng.require_device()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
rng = np.random.default_rng(12)
host = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.choice(4, size=n, p=[0.55, 0.05, 0.05, 0.35])]
t = host.tobytes()
text = torch.from_numpy(np.ascontiguousarray(host)).cuda()
exp = oracle.sais(t); el = oracle.lcp_kasai(t, exp)
sa, lcp = sdev.build_sa_lcp(text)
st = eng.build_stats()
got = lcp.cpu().numpy().view(np.uint32)
bad = np.flatnonzero(got != el)
print("fused: sa ok", np.array_equal(sa.cpu().numpy().view(np.uint32), exp), "lcp mismatches", bad.size, bad[:8], got[bad[:8]], el[bad[:8]],
      {k: st[k] for k in ("rounds", "tile_sorted", "large_sorted", "small_bucket_resolved", "text_rounds", "rank_rounds")})
if os.environ.get("SKEW_DETAIL"):
    for r in bad[:3].tolist():
        a, b = int(exp[r - 1]), int(exp[r])
        print(" rank", r, "suffixes", a, b, t[a:a + 34], t[b:b + 34])
        # the bucket of suffixes sharing 16 symbols with them, and the class sharing 16 + 15
        lo = r
        while lo > 0 and el[lo] >= 16: lo -= 1
        hi = r
        while hi + 1 < n and el[hi + 1] >= 16: hi += 1
        print("  16-symbol bucket: ranks", lo, hi, "size", hi - lo + 1, "offset in bucket", r - lo)
        print("  lcp around:", el[max(lo, r - 4):r + 5], "got", got[max(lo, r - 4):r + 5])
        print("  31-symbol classes in the bucket:", int((el[lo + 1:hi + 1] < 31).sum()) + 1)
