#!/usr/bin/env python3
"""Out-of-bounds check of the kernels' global-memory accesses: the product .hip sources built
with AddressSanitizer against the fiber emulator (make -C tests/emu asan), driven through the
same C ABI.  Not part of the pytest suite (several minutes); run by hand after kernel changes:

    make -C tests/emu asan
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) \\
    ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1 \\
    SFX_LCP_DIRECT_MIN=8 python tests/asan_check.py            # variants: add SFX_PARTITION_MIN=1 SFX_MAX_GRID=3 SFX_QUERY_PHASE_MIN=1 SFX_HYBRID_MIN=1 [SFX_HYBRID_CAP=100]
                                                                # SFX_FORCE_KEY64=1 SFX_HT_MIN=1 [SFX_DEEP_ITERS=1 SFX_TILE_SMALL=1 SFX_SEG_SMALL=1]
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import numpy as np  # noqa: E402

import _cases  # noqa: E402
import _gen  # noqa: E402
import oracle  # noqa: E402
from suffix_amd import Engine, SuffixTable  # noqa: E402

oracle.build()
eng = Engine(os.path.join(HERE, "emu", "asan", "libsuffix_emu.so"))
texts = [_gen.dna(30000, seed=9).tobytes(), _gen.english_like(9000, seed=3).tobytes(),
         b"AAAAAAAAAAAAAAAAAAAAAAAC" * 300, _gen.dna(5000, seed=3).tobytes() + b"A" * 1500,
         _gen.utf8_mixed(4001).tobytes(), b"a", b"ab", b"",
         _gen.dna(20000, seed=11).tobytes() + b"A" * 1200 + _gen.dna(1000, seed=12).tobytes()]
for t in texts:
    st = SuffixTable(t, engine=eng)
    exp = oracle.sais(t)
    assert np.array_equal(st.table(), exp)
    assert np.array_equal(st.lcp_lens(), oracle.lcp_quadratic(t, exp))
    if len(t) > 10:
        q = t[5:9]
        s, e = oracle.positions(t, exp, q)
        assert sorted(st.positions(q).tolist()) == sorted(exp[s:e].tolist())
print("tables ok")
for nr in (1, 3, 11):
    _cases.range_slices(eng, oracle, _gen.dna(3001, seed=8).tobytes(), nr, packed=True)
    _cases.range_slices(eng, oracle, _gen.english_like(2503).tobytes(), nr)
    _cases.range_slices(eng, oracle, b"ab" * 150 + b"b", nr)
print("ranges ok")
if os.environ.get("SFX_HYBRID_MIN"):
    # hybrid initial sort: two sub-buckets whose counts wrap the 16-bit counters of the histogram (seen in the total:
    # the build takes the four-pass sort)
    t = b"AC" * 70000 + _gen.dna(3000, seed=5).tobytes()
    assert np.array_equal(SuffixTable(t, engine=eng).table(), oracle.sais(t))
    print("hybrid counter wrap ok")
_cases.directory_queries(eng, oracle)              # resident index: directory, 16-byte key tree, both query phases
print("index queries ok")
_cases.suffix_tree_topology(eng, oracle)
_cases.fused_lcp_tails(eng, oracle, iters=6)
print("tree + fused lcp ok")
# large buckets of the refinement rounds: the three LDS size classes of k_seg_single (2000, 5000 and 12000 copies of
# one word) and, with SFX_PARTITION_MIN=1, rank rounds whose pair histogram is counted inside groups_apply
rng = np.random.default_rng(4)
def planted(copies, alphabet, wlen, tail):
    w = bytes(rng.choice(list(alphabet), wlen).tolist())
    return b"".join(w + bytes(rng.choice(list(alphabet), tail).tolist()) for _ in range(copies))
for t in (planted(5000, b"ACGT", 16, 20) + planted(2000, b"ACGT", 16, 24), planted(12000, b"ACGT", 16, 12),
          b"AAAAAAAAAAAAAAAAAAAAAAAC" * 600 + _gen.english_like(15000, seed=3).tobytes() * 3):
    st = SuffixTable(t, engine=eng)
    assert np.array_equal(st.table(), oracle.sais(t))
print("large buckets + rank rounds ok")
# round 3: deep text rounds (per-bucket depths, residues, fused LCP emission), compressed 64-bit keys (SFX_FORCE_KEY64=1
# SFX_HT_MIN=1), the suffix-tree topology's open list on a monotone LCP array, slices through the hybrid route
# (SFX_HYBRID_MIN=1: keys relative to the range's first key)
rng2 = np.random.default_rng(2024)
for sigma, n0 in ((4, 9000), (60, 6000), (200, 6000)):
    body = rng2.integers(0, sigma, n0, dtype=np.uint8)
    parts = [body.tobytes()]
    for _ in range(8):
        a = int(rng2.integers(0, n0 - 400)); ln = int(rng2.integers(20, 300)); cp = int(rng2.integers(2, 30))
        for _ in range(cp):
            parts.append(body[a:a + ln].tobytes() + bytes(rng2.integers(0, sigma, 3, dtype=np.uint8).tolist()))
    t = b"".join(parts)
    exp = oracle.sais(t)
    st2, lcp2 = SuffixTable.new_with_lcp(t, engine=eng)
    assert np.array_equal(st2.table(), exp) and np.array_equal(lcp2, oracle.lcp_kasai(t, exp))
zipf = np.minimum(np.random.default_rng(99).zipf(1.3, 12000), 200).astype(np.uint8).tobytes()
for t in (zipf + bytes([250]) + zipf[:2000], _gen.utf8_mixed(9000).tobytes(), _gen.english_like(20000, seed=5).tobytes()):
    assert np.array_equal(SuffixTable(t, engine=eng).table(), oracle.sais(t))
print("deep rounds + compressed keys ok")
_cases.suffix_tree_at_scale(eng, oracle, b"a" * 9000 + b"b" + _gen.dna(7000, seed=2).tobytes(), device="cpu")
for nr in (1, 3):
    _cases.range_slices(eng, oracle, _gen.dna(56001, seed=8).tobytes(), nr, packed=True)
print("tree at scale + slice hybrid ok")
