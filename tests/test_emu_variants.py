"""Non-default code paths of the engine, on the CPU emulator: the tuning / test-hook
environment variables are read once per process, so every variant runs in a subprocess.
  SFX_RADIX_SWEEP / _NW / _KPT / _RANK   radix schedules, tile geometries, ranking methods
  SFX_MAX_GRID                           multi-tile chunks per workgroup on small inputs
  SFX_PARTITION_MIN                      partitioned (cache-confined) rank / Phi scatters
  SFX_LCP_DIRECT_MIN                     sampled choice between direct and Phi/PLCP LCP, cap + fallback
  SFX_TILE_SMALL / SFX_FORCE_KEY64       small LDS windows of the refinement rounds; 64-bit initial keys
  SFX_SEG_SMALL                          small tiles in the segmented sort of the large buckets
  SFX_HYBRID_MIN / _GEOM / _CAP / _PARTITION   hybrid initial sort (two device-wide passes + LDS sort of the sub-buckets)
Every run compares SA and LCP with the oracle on a few texts that exercise the path."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

SCRIPT = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {here!r})
import numpy as np
import _gen, oracle
from suffix_amd import Engine, SuffixTable
oracle.build()
eng = Engine(os.path.join({here!r}, "emu", "libsuffix_emu.so"))
texts = [_gen.dna(24000, seed=9).tobytes(),                      # E64 path, several tiles
         _gen.english_like(10000, seed=3).tobytes(),             # 64-bit keys, rank rounds
         b"AAAAAAAAAAAAAAAAAAAAAAAC" * 400,                      # long repeats: many rounds
         _gen.dna(6000, seed=3).tobytes() + b"A" * 2000]
if os.environ.get("SFX_LCP_DIRECT_MIN"):
    # the sample says "low LCP", one run reaches the cap of the direct path -> Phi/PLCP redoes the array;
    # then short texts whose last windows run off the end
    texts.append(_gen.dna(50000, seed=11).tobytes() + b"A" * 1500 + _gen.dna(3000, seed=12).tobytes())
    texts += [b"abcabcabcabcabcabcabx", b"a" * 40, _gen.utf8_mixed(5000).tobytes()]
    # packed windows at 1, 3 and 4 bits per symbol, with repeats longer than one window
    rng = np.random.default_rng(77)
    for sigma in (2, 6, 12):
        body = rng.integers(0, sigma, 12000, dtype=np.uint8) + 97
        rep = body[1000:1000 + 90 + 40 * sigma].tobytes()
        texts.append(body.tobytes() + rep + b"a" + rep)
if os.environ.get("SFX_HYBRID_MIN"):
    # keys of 32 bits (sigma 2, 4, 16) and of 30 bits (sigma 5: 14 low bits, second LDS digit of 6 bits).  (A wrapped
    # 16-bit counter of the histogram needs 65536 equal prefixes in one workgroup's stretch: tests/test_gpu_parity.py
    # and tests/asan_check.py.)
    # four 8-symbol blocks, each followed by 8 random symbols: a few sub-buckets of ~900 suffixes whose low 16 key
    # bits are spread out (the grouped all-pairs path of the LDS sort), next to sub-buckets of one or two
    rngh = np.random.default_rng(5)
    blocks = [bytes(rngh.choice(list(b"ACGT"), 8).tolist()) for _ in range(4)]
    texts.append(b"".join(blocks[int(k)] + bytes(rngh.choice(list(b"ACGT"), 8).tolist()) for k in rngh.integers(0, 4, 3600)))
    texts += [_gen.uniform_bytes(12000, 5, 2, base=65).tobytes(), _gen.uniform_bytes(12000, 16, 3, base=65).tobytes(),
              _gen.uniform_bytes(14000, 2, 4, base=65).tobytes()]
    # one 8-symbol block followed by one of 700 10-symbol tails, every tail twice: a sub-bucket of ~1400 suffixes in 700 runs of
    # two equal keys (equal over the 18 symbols of the longer keys of round 6 as well), spread over the groups of the fast path
    # and side by side in the sorted order
    b8 = bytes(rngh.choice(list(b"ACGT"), 8).tolist())
    tails = [bytes(rngh.choice(list(b"ACGT"), 10).tolist()) for _ in range(700)]
    order = rngh.permutation(1400) % 700
    runs_of_two = b"".join(b8 + tails[int(k)] + bytes(rngh.choice(list(b"ACGT"), 9).tolist()) for k in order)
    texts.append(runs_of_two)
    # one 15-symbol block, each copy followed by 3 random symbols: a sub-bucket of ~700 suffixes that all fall into ONE group of
    # the LDS sort's fast path (the group is the top 10 of the low key bits) -> the stable LSD rounds (three of them over the 20 low
    # bits of the longer keys), with runs of ~11 equal 18-symbol keys: the tie bits (round 6) from neighbours in the staging
    # buffer instead of from the group scan
    b13 = bytes(rngh.choice(list(b"ACGT"), 15).tolist())
    runs_of_eleven = _gen.dna(9000, seed=21).tobytes() + b"".join(b13 + bytes(rngh.choice(list(b"ACGT"), 3).tolist()) for _ in range(700))
    texts.append(runs_of_eleven)
    from suffix_amd import device as sdev
    for t in texts:                                     # the fused SA + LCP entry over the same initial sort (round 6: tie bits AND sorted keys)
        import torch
        d = torch.frombuffer(bytearray(t), dtype=torch.uint8)
        sa, lcp = sdev.build_sa_lcp(d, engine=eng)
        exp = oracle.sais(t)
        assert np.array_equal(sa.numpy().view(np.uint32), exp) and np.array_equal(lcp.numpy().view(np.uint32), oracle.lcp_kasai(t, exp))
    # which route the first text takes (random DNA: no sub-bucket above a handful) and a text with three planted
    # sub-buckets of ~130 suffixes (oversized under SFX_HYBRID_CAP=100: gathered, sorted device-wide, copied back)
    planted = _gen.dna(44000, seed=3).tobytes() + b"".join(
        blocks[int(k)] + bytes(rngh.choice(list(b"ACGT"), 8).tolist()) for k in rngh.integers(0, 3, 400))
    texts.append(planted)
    def kernels_of(t):
        eng.profile(True); eng.profile_reset()
        SuffixTable(t, engine=eng).table()
        # (the LDS sort of the sub-buckets reports as bucket_sort_ties when it leaves tie bits, bucket_sort_lds when sorted keys)
        names = ["bucket_sort_lds" if r["name"].startswith("bucket_sort_ties") else r["name"] for r in eng.profile_report()]
        eng.profile(False)
        return names
    cap = int(os.environ.get("SFX_HYBRID_CAP", "100000"))
    ties_on = os.environ.get("SFX_HYBRID_TIES", "1") != "0"
    names = kernels_of(texts[0])
    # (cap 3: 6 % of that text sits in sub-buckets of more than 3 suffixes -- above the 1/64 the route tolerates)
    assert ("bucket_sort_lds" in names) == (cap > 10) and "oversize_gather" not in names, names
    # round 6: with no oversized sub-bucket the LDS sort names the tied elements itself -- no keys written, none read back
    assert ("tie_direct" in names) == (cap > 10 and ties_on) and ("groups_reduce" in names) != ("tie_direct" in names), names
    if cap >= 100000 and ties_on:
        # random DNA: every stretch of tied slots ordered where it is, no list at all; the 700 runs of two lie side by side in their
        # sub-bucket -- ONE stretch of 1400 tied slots, more than k_tie_direct takes -- and the runs of eleven are too long: ALL
        # runs become the first active list (k_tie_heads tells the runs of a stretch apart)
        names2 = kernels_of(runs_of_two)
        assert "tie_list" not in names and "small_groups" not in names, names
        assert "tie_heads" in names2 and "tie_list" in names2 and "small_groups" in names2, names2
        names2 = kernels_of(runs_of_eleven)
        assert "tie_list_count" in names2 and "tie_list" in names2 and "deep_wave" in names2, names2    # (runs of eleven: no direct pass of the list)
    names = kernels_of(planted)                                       # 0.8 % of it in the three planted sub-buckets
    assert ("bucket_sort_lds" in names) == (cap > 10) and ("oversize_gather" in names) == (10 < cap < 400), names
    assert ("tie_direct" in names) == (cap >= 400 and ties_on), names   # (an oversized sub-bucket: the sorted keys, as before)
    # most of the suffixes in oversized sub-buckets: the four-pass sort
    skewed = np.frombuffer(b"ACGT", dtype=np.uint8)[rngh.choice(4, size=20000, p=[0.85, 0.05, 0.05, 0.05])].tobytes()
    texts.append(skewed)
    if cap < 400:
        names = kernels_of(skewed)
        assert "radix_hist16_text" in names and "bucket_sort_lds" not in names, names
    # slices of the partitioned build: the same route over elements that exist already (keys relative to the range's first
    # key, sub-bucket = their top 16 bits), for 1, 3 and 7 virtual ranks, over 2- and 4-bit symbols
    import _cases
    # (a slice needs 16384 elements for one partial histogram to fit the idle element buffer: 7 ranks take four passes)
    for nr in ((3, 7) if cap < 400 else (1, 3)):
        eng.profile(True); eng.profile_reset()
        _cases.range_slices(eng, oracle, _gen.dna(56001, seed=8).tobytes(), nr, packed=True)
        seen = set("bucket_sort_lds" if r["name"].startswith("bucket_sort_ties") else r["name"] for r in eng.profile_report())
        eng.profile(False)
        # (one rank = the whole key space: no filter, the text-fed route of the full build)
        assert ("radix_hist16_elems" in seen) == (nr == 3) and ("radix_hist16_text" in seen) == (nr == 1), (nr, seen)
        assert ("bucket_sort_lds" in seen) == (cap > 10 and nr < 7) and ("range_emit" in seen) == (nr > 1), (nr, seen)
        assert ("tie_direct" in seen) == (cap > 10 and nr < 7 and ties_on), (nr, seen)       # (slices take the records too)
    if cap < 400:
        _cases.range_slices(eng, oracle, planted, 3, packed=True)
        _cases.range_slices(eng, oracle, skewed, 2, packed=True)
    else:
        _cases.range_slices(eng, oracle, _gen.uniform_bytes(40000, 16, 3, base=65).tobytes(), 2)
if os.environ.get("SFX_HT_MIN"):
    # compressed keys: skewed symbol counts (long and short codes side by side), a symbol that occurs once, runs of the
    # smallest symbol (its code is all zeros, like the padding past the end) at the end of the text and before it
    rngh2 = np.random.default_rng(99)
    zipf = np.minimum(rngh2.zipf(1.3, 10000), 200).astype(np.uint8)
    texts.append(zipf.tobytes() + bytes([250]) + zipf[:3000].tobytes())
    low = bytes([int(zipf.min())])
    texts.append(zipf[:6000].tobytes() + low * 40 + zipf[5000:8000].tobytes() + low * 25)
    texts.append(_gen.utf8_mixed(8000).tobytes())
if os.environ.get("SFX_DEEP_ITERS") or os.environ.get("SFX_DEEP_KPT"):
    # deep text rounds: buckets finished inside one wave, members that stay tied leaving with their own depth (capped
    # iterations: every round leaves such buckets), buckets above the wave's window on the large path, fused LCP values
    # from the keys of the iteration that splits a pair.  Repeats of 20 .. 300 symbols in 2 .. 40 copies over three alphabets.
    rngd = np.random.default_rng(2024)
    for sigma, n0 in ((4, 7000), (60, 5000), (200, 5000)):
        body = rngd.integers(0, sigma, n0, dtype=np.uint8)
        parts = [body.tobytes()]
        for _ in range(8):
            a = int(rngd.integers(0, n0 - 400)); ln = int(rngd.integers(20, 300)); cp = int(rngd.integers(2, 24))
            for _ in range(cp):
                parts.append(body[a:a + ln].tobytes() + bytes(rngd.integers(0, sigma, 3, dtype=np.uint8).tolist()))
        texts.append(b"".join(parts))
    texts.append(_gen.english_like(7000, seed=8).tobytes() * 2 + b"!")
    for t in texts:
        exp = oracle.sais(t)
        st2, lcp2 = SuffixTable.new_with_lcp(t, engine=eng)
        assert np.array_equal(st2.table(), exp), ("fused SA", len(t))
        assert np.array_equal(lcp2, oracle.lcp_kasai(t, exp)), ("fused LCP", len(t))
    texts = texts[:2] + texts[-2:]                     # (the separate calls below: the same kernels without the LCP emission)
# (TEST_TEXTS: how many of the texts a variant needs -- the radix schedules see every pass on the first two, the index
# variants have their own cases below)
texts = texts[:int(os.environ.get("TEST_TEXTS", len(texts)))]
for t in texts:
    st = SuffixTable(t, engine=eng)
    exp = oracle.sais(t)
    assert np.array_equal(st.table(), exp), ("SA", len(t))
    want = oracle.lcp_kasai(t, exp) if len(t) > 50000 else oracle.lcp_quadratic(t, exp)     # (quadratic: hopeless on long periodic texts)
    assert np.array_equal(st.lcp_lens(), want), ("LCP", len(t))
if os.environ.get("SFX_LCP_DIRECT_MIN"):
    def lcp_kernels(t):
        st = SuffixTable(t, engine=eng)
        st.table()
        eng.profile(True); eng.profile_reset()
        st.lcp_lens()
        names = [r["name"] for r in eng.profile_report()]
        eng.profile(False)
        return names
    k = lcp_kernels(texts[0])                          # DNA: direct, on packed symbols
    assert "lcp_windows_packed" in k and "plcp" not in k, k
    k = lcp_kernels(texts[1])                          # sigma > 16: direct, on the raw bytes
    assert "lcp_windows" in k and "plcp" not in k, k
    k = lcp_kernels(texts[2])                          # the sample says no
    assert k == ["lcp_sample", "phi_scatter", "plcp", "lcp_gather"], k
    k = lcp_kernels(texts[4])                          # cap reached: Phi/PLCP redoes the array
    assert "lcp_windows_packed" in k and k[-3:] == ["phi_scatter", "plcp", "lcp_gather"], k
if os.environ.get("SFX_INDEX_TREE") or os.environ.get("SFX_QUERY_PHASE_MIN"):
    import _cases
    _cases.directory_queries(eng, oracle)
print("OK")
"""

VARIANTS = {
    "chunked-multi-tile": {"SFX_RADIX_SWEEP": "0", "SFX_MAX_GRID": "2", "TEST_TEXTS": "2"},
    "one-sweep-4-waves-kpt16-ballot": {"SFX_RADIX_NW": "4", "SFX_RADIX_KPT": "16", "SFX_RADIX_RANK": "0", "TEST_TEXTS": "2"},
    "one-sweep-8-waves-kpt8": {"SFX_RADIX_NW": "8", "SFX_RADIX_KPT": "8", "SFX_MAX_GRID": "3", "TEST_TEXTS": "2"},
    "partitioned-scatter": {"SFX_PARTITION_MIN": "1"},
    "direct-lcp": {"SFX_LCP_DIRECT_MIN": "8"},
    # the byte-window kernel at both window widths (round 5: 32 bytes where the sample's mean LCP is >= 16 bytes, sfx_lcp.hip), whatever
    # the sample says
    "direct-lcp-32-byte-windows": {"SFX_LCP_DIRECT_MIN": "8", "SFX_LCP_WINDOW": "4"},
    # 256-element LDS windows: buckets cross tile boundaries, > 128 members take the large-bucket path
    "small-tiles": {"SFX_TILE_SMALL": "1"},
    # ... and 4096-element tiles in the segmented sort of the large buckets: multi-tile segments, look-back inside a segment
    "small-tiles-small-segments": {"SFX_TILE_SMALL": "1", "SFX_SEG_SMALL": "1"},
    "small-tiles-key64-multi-tile": {"SFX_TILE_SMALL": "1", "SFX_FORCE_KEY64": "1", "SFX_MAX_GRID": "3", "SFX_SEG_SMALL": "1"},
    "key64": {"SFX_FORCE_KEY64": "1"},
    # 64-bit keys: the middle passes move 12-byte (key, suffix) elements (KV12, round 5) -- here through the chunked schedule's
    # kernel with multi-tile chunks, and the (key array, value array) form of rounds 1-4 in every pass
    "key64-chunked-multi-tile": {"SFX_FORCE_KEY64": "1", "SFX_RADIX_SWEEP": "0", "SFX_MAX_GRID": "2", "TEST_TEXTS": "3"},
    "key64-split-arrays": {"SFX_FORCE_KEY64": "1", "SFX_RADIX_KV12": "0", "TEST_TEXTS": "3"},
    # ... and k_radix_sweep (one workgroup per CU) instead of k_radix_sweep_duo (two, the default of the one-sweep KV passes since
    # round 5); the duo kernel with multi-tile inputs: 64-bit keys on every text with the grid capped
    "key64-one-workgroup-per-cu": {"SFX_FORCE_KEY64": "1", "SFX_RADIX_DUO": "0", "TEST_TEXTS": "3"},
    "key64-duo-multi-tile": {"SFX_FORCE_KEY64": "1", "SFX_MAX_GRID": "2"},
    # the same kernel over 8-byte elements (development route: measured, not adopted), incl. the rank-update partition passes
    "e64-duo-multi-tile": {"SFX_RADIX_DUO_E64": "1", "SFX_MAX_GRID": "2", "SFX_PARTITION_MIN": "1", "SFX_HYBRID": "0"},
    # hybrid initial sort forced on small inputs
    "hybrid-initial-sort": {"SFX_HYBRID_MIN": "1"},
    # ... with the partition passes as one 16-wave workgroup per CU (round 4's geometry) / four 4-wave ones instead of two 8-wave ones
    "hybrid-initial-sort-16-wave-partition": {"SFX_HYBRID_MIN": "1", "SFX_PARTITION_WAVES": "16", "TEST_TEXTS": "2"},
    "hybrid-initial-sort-4-wave-partition": {"SFX_HYBRID_MIN": "1", "SFX_PARTITION_WAVES": "4", "SFX_MAX_GRID": "3", "TEST_TEXTS": "2"},
    # ... with the stable one-sweep passes of rounds 2-3 instead of the partition passes (k_partition)
    "hybrid-initial-sort-one-sweep-passes": {"SFX_HYBRID_MIN": "1", "SFX_HYBRID_PARTITION": "0", "TEST_TEXTS": "2"},
    # ... with the sorted keys written and read back by the bucket pass, as rounds 3-5 (round 6: the LDS sort leaves tie records)
    "hybrid-initial-sort-sorted-keys": {"SFX_HYBRID_MIN": "1", "SFX_HYBRID_TIES": "0"},
    # ... the tie bits over elements of 32 + 32 bits (round 6's default lets a text-fed key grow by one symbol into the bits a suffix
    # index of <= 2^28 leaves free)
    "hybrid-initial-sort-32-bit-keys": {"SFX_HYBRID_MIN": "1", "SFX_HYBRID_KEY36": "0", "TEST_TEXTS": "9"},
    # ... and two more symbols of DNA instead of one (a 12-byte load per element: measured, not the default)
    "hybrid-initial-sort-two-more-symbols": {"SFX_HYBRID_MIN": "1", "SFX_HYBRID_KEY36": "2", "TEST_TEXTS": "9"},
    # ... the tie records from the 1024 x 16 geometry (sub-buckets of up to 16384), several sub-buckets per workgroup
    "hybrid-initial-sort-1024x16": {"SFX_HYBRID_MIN": "1", "SFX_HYBRID_GEOM": "2", "SFX_MAX_GRID": "2"},
    # a few oversized sub-buckets (gathered, sorted device-wide, copied back), the 256 x 16 geometry, several sub-buckets
    # per workgroup
    "hybrid-initial-sort-oversized": {"SFX_HYBRID_MIN": "1", "SFX_HYBRID_CAP": "100", "SFX_MAX_GRID": "3", "SFX_HYBRID_GEOM": "1"},
    "index-directory-only": {"SFX_INDEX_TREE": "0", "TEST_TEXTS": "0"},
    # queries longer than the tree's keys listed for a second launch (batches of >= 4096 by default), with and
    # without the (opt-in) ordering of the batch
    "index-two-phase-queries": {"SFX_QUERY_PHASE_MIN": "1", "TEST_TEXTS": "0"},
    "index-two-phase-ordered": {"SFX_QUERY_PHASE_MIN": "1", "SFX_QUERY_ORDER": "1", "TEST_TEXTS": "0"},
    # rank rounds from the first round on (SFX_START_RANKS: a development route -- measured in round 5 on texts whose 64-bit keys leave
    # >= 95 % of the suffixes tied and NOT adopted, sort_and_refine), over compressed 64-bit keys; with the fused LCP (the deep-round
    # texts) the values are bounds from the start
    "start-with-rank-rounds-compressed-keys": {"SFX_START_RANKS": "1", "SFX_FORCE_KEY64": "1", "SFX_HT_MIN": "1", "SFX_DEEP_ITERS": "24"},
    # context codes (round 6, k_ht_keys_ctx): one order-preserving code per class of the preceding symbol, the number of symbols a
    # key holds in its low 4 bits -- forced on small inputs (the build takes them from 2^24 bytes on, where they buy 10 % more symbols)
    "context-codes": {"SFX_FORCE_KEY64": "1", "SFX_HT_MIN": "1", "SFX_HT_CTX": "2", "SFX_HT_CTX_MIN": "1"},
    # ... and left to the build's own decision: the 10 % rule on the code lengths, then the pilot sort (here of 2^6 .. n / 4 suffixes)
    "context-codes-by-pilot": {"SFX_FORCE_KEY64": "1", "SFX_HT_MIN": "1", "SFX_HT_CTX": "1", "SFX_HT_CTX_MIN": "1", "SFX_HT_CTX_PILOT": "6"},
    "context-codes-small-tiles": {"SFX_FORCE_KEY64": "1", "SFX_HT_MIN": "1", "SFX_HT_CTX": "2", "SFX_HT_CTX_MIN": "1", "SFX_TILE_SMALL": "1",
                                  "SFX_MAX_GRID": "3"},
    # rank rounds through round 1's composite-key sort (the fallback for key2 = rank + h beyond 32 bits)
    "composite-rank-rounds": {"SFX_FORCE_COMPOSITE": "1"},
    "tile-1024x4-pair32": {"SFX_TILE_GEOM": "1", "SFX_TILE_PAIR": "32"},
    "tile-512x4-key64": {"SFX_TILE_GEOM": "3", "SFX_FORCE_KEY64": "1"},
    # deep text rounds (k_deep_wave): one / two iterations per round (every round leaves tied members with their own
    # depth), 128-position windows (more buckets on the large path, more waves per text), the other window sizes
    "deep-one-iteration": {"SFX_DEEP_ITERS": "1"},
    "deep-two-iterations-small-windows": {"SFX_DEEP_ITERS": "2", "SFX_TILE_SMALL": "1", "SFX_SEG_SMALL": "1"},
    "deep-small-windows-key64": {"SFX_DEEP_ITERS": "24", "SFX_TILE_SMALL": "1", "SFX_FORCE_KEY64": "1"},
    "deep-512-position-windows": {"SFX_DEEP_KPT": "8", "SFX_DEEP_ITERS": "3"},
    # buckets that would pass the depth limit leave the deep kernel (evicted from the wave's slots with the depth they have
    # reached) while the wave's other buckets go on.  Compressed keys start the buckets of one wave at different depths, so with
    # the limit lowered to 24 symbols a wave evicts SOME of its buckets (checked once with a counter: 106 partial
    # evictions on the doubled English-like text, 3 complete ones); fixed-width keys evict a wave's buckets together
    "deep-buckets-past-the-depth-limit-compressed-keys": {"SFX_FORCE_KEY64": "1", "SFX_HT_MIN": "1", "SFX_DEEP_ITERS": "24", "SFX_DEEP_MAX_DEPTH": "24"},
    # 64-bit initial keys in an order-preserving prefix code (k_ht_keys): buckets of different depths from the first list on
    "compressed-keys": {"SFX_FORCE_KEY64": "1", "SFX_HT_MIN": "1", "SFX_DEEP_ITERS": "24"},
    "compressed-keys-small-windows-one-iteration": {"SFX_FORCE_KEY64": "1", "SFX_HT_MIN": "1", "SFX_DEEP_ITERS": "1", "SFX_TILE_SMALL": "1",
                                                    "SFX_SEG_SMALL": "1", "SFX_MAX_GRID": "3"},
}


def _run_variant(name, script_path):
    env = dict(os.environ, SFX_TINY="0", **VARIANTS[name])        # (SFX_TINY=0: the variants are about the general build)
    return subprocess.run([sys.executable, script_path], env=env, capture_output=True, text=True, timeout=1500)


@pytest.fixture(scope="module")
def variant_runs(request, tmp_path_factory):
    """Every selected variant is an independent subprocess (its own emulator library instance, its own environment): they are
    started together, a few at a time, when the first of them is asked for, and each test waits for its own -- the 30-odd
    variants cost the wall time of the longest few instead of their sum."""
    import concurrent.futures
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(HERE, "emu")])
    script = tmp_path_factory.mktemp("variants") / "variant.py"
    script.write_text(SCRIPT.format(root=ROOT, here=HERE))
    wanted = []
    for item in request.session.items:
        if item.name.startswith("test_engine_variant_on_emulator[") and item.name.endswith("]"):
            wanted.append(item.name[len("test_engine_variant_on_emulator["):-1])
    workers = max(1, min(4, (os.cpu_count() or 2) // 2))
    pool = concurrent.futures.ThreadPoolExecutor(max_workers=workers)
    futures = {name: pool.submit(_run_variant, name, str(script)) for name in wanted if name in VARIANTS}
    yield futures, str(script)
    pool.shutdown(wait=False, cancel_futures=True)


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_engine_variant_on_emulator(variant_runs, name):
    futures, script = variant_runs
    out = futures[name].result() if name in futures else _run_variant(name, script)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
