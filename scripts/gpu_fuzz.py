#!/usr/bin/env python3
"""One-off randomized parity run on the GPU beyond the sizes of the test suite: python scripts/gpu_fuzz.py [iters] [max_len] [seed] [min_len]
Random length, alphabet, symbol distribution (uniform / Zipf / two dominant symbols) and repeat structure (planted copies,
periodic stretches, trailing runs, few distinct words); every text through new(), lcp_lens(), the one-call entry and a few
queries, all compared with the oracle (tests/_cases.check_text; Kasai instead of the quadratic LCP for speed)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle, suffix_amd
import _devlib
from suffix_amd import SuffixTable
oracle.build()
eng = _devlib.engine(); eng.require_device()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
max_len = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1234)
min_len = int(sys.argv[4]) if len(sys.argv) > 4 else 50_000          # (below 16 KiB: the single-workgroup build)
t0 = time.time()
for it in range(iters):
    n = int(rng.integers(min_len, max_len))
    sigma = int(rng.choice([2, 4, 5, 17, 20, 64, 66, 100, 141, 256]))
    dist = it % 3
    if dist == 0 or sigma < 5:
        body = rng.integers(0, sigma, n, dtype=np.uint8)
    elif dist == 1:
        p = 1.0 / np.arange(1, sigma + 1) ** float(rng.uniform(0.8, 1.6)); p /= p.sum()
        body = rng.choice(sigma, size=n, p=rng.permutation(p)).astype(np.uint8)
    else:
        p = np.full(sigma, 0.1 / (sigma - 2)); p[:2] = 0.45
        body = rng.choice(sigma, size=n, p=rng.permutation(p)).astype(np.uint8)
    if sigma < 200:
        body = body + np.uint8(rng.integers(0, 256 - sigma))
    t = bytearray(body.tobytes())
    kind = (it // 3) % 5
    if kind == 1:
        for _ in range(int(rng.integers(1, 6))):
            a = int(rng.integers(0, n - 20)); ln = int(rng.integers(10, min(50_000, n - a)))
            t += t[a:a + ln]
    elif kind == 2:
        t += bytes(t[:int(rng.integers(1, 8))]) * int(rng.integers(10, 3000))
    elif kind == 3:
        t += bytes([min(t)]) * int(rng.integers(1, 200))
    elif kind == 4:
        words = [bytes(t[i:i + int(rng.integers(2, 9))]) for i in rng.integers(0, n - 10, 12)]
        t = bytearray(b" ".join(words[int(k)] for k in rng.integers(0, 12, n // 8)))
    t = bytes(t)
    exp = oracle.sais(t)
    st = SuffixTable(t, engine=eng)
    assert np.array_equal(st.table(), exp), ("SA", it, len(t), sigma, dist, kind)
    stats = eng.build_stats()
    want = oracle.lcp_kasai(t, exp)
    assert np.array_equal(st.lcp_lens(), want), ("LCP", it)
    st2, lcp2 = SuffixTable.new_with_lcp(t, engine=eng)
    assert np.array_equal(st2.table(), exp) and np.array_equal(lcp2, want), ("one-call", it)
    qs = [t[int(a):int(a) + int(rng.integers(1, 12))] for a in rng.integers(0, len(t), 4)] + [b"\x00", t[-3:]]
    s, e = st.positions_batch(qs)
    for k, q in enumerate(qs):
        assert (int(s[k]), int(e[k])) == oracle.positions(t, exp, q), ("query", it, q)
    print(it, len(t), sigma, dist, kind, stats["key_bits"], stats["text_rounds"], stats["rank_rounds"], flush=True)
print("fuzz ok", iters, round(time.time() - t0, 1), "s")
