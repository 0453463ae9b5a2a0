#!/usr/bin/env python3
"""How small are the buckets a deep-round wave sees?  (Round 5, VERDICT item 4: a smaller compare network for k_deep_wave.)

CPU only.  100 MB of the config-3 generator, the oracle's SA and LCP array; for the depths d that leave about the share of
suffixes tied that config 3's initial sort leaves at 10^9 bytes (54 %), the buckets are the maximal runs of lcp >= d.  The
active list is laid out as the engine lays it out (buckets in SA order), cut into the 128-position stretches one wave owns,
and for every stretch the largest bucket of <= 128 members whose head lies in it is taken: a wave could use a cheaper
ordering only if that maximum is small.  Output: profiles/r5_bucket_stats.txt."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _gen  # noqa: E402
import oracle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
t = _gen.english_like(n)
sa = oracle.sais(t)
lcp = oracle.lcp_kasai(t, sa)
for d in (8, 10, 12):
    x = np.concatenate(([0], (lcp >= d).view(np.int8)[1:], [0]))
    dx = np.diff(x)
    starts, ends = np.nonzero(dx == 1)[0], np.nonzero(dx == -1)[0]
    sizes = ends - starts + 1
    members = int(sizes.sum())
    print(f"depth {d}: {members / n:.3f} of the suffixes tied, {len(sizes)} buckets, mean size {members / len(sizes):.2f}")
    heads = np.cumsum(np.concatenate(([0], sizes[:-1])))
    stretch = heads // 128
    small = sizes <= 128
    nst = int(stretch.max()) + 1
    mx = np.zeros(nst, dtype=np.int64)
    np.maximum.at(mx, stretch[small], sizes[small])
    cnt = np.zeros(nst, dtype=np.int64)
    np.add.at(cnt, stretch[small], sizes[small])
    has = cnt > 0
    for S in (2, 4, 8, 16, 32):
        sel = has & (mx <= S)
        print(f"   waves whose largest owned bucket is <= {S:2d}: {sel.sum() / has.sum():.3f} of the waves, {cnt[sel].sum() / cnt[has].sum():.3f} of the members")
    sm = sizes[small]
    print(f"   members in owned buckets of <= 4: {sm[sm <= 4].sum() / sm.sum():.3f}, <= 8: {sm[sm <= 8].sum() / sm.sum():.3f}, <= 16: {sm[sm <= 16].sum() / sm.sum():.3f}")
