"""GPU parity tests (run with -m gpu on a real MI355X): the product library
through the C ABI vs the oracle / golden vectors -- bit-exact."""
import hashlib

import numpy as np
import pytest
import torch  # noqa: F401  (first: one HIP runtime per process, see suffix_amd/_lib.py)

import _cases
import _gen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import suffix_amd
    e = suffix_amd.default_engine()
    e.require_device()                      # fail loudly: no CPU fallback
    assert e.path.endswith("libsuffix_hip.so")
    return e


def test_first_build_on_a_fresh_thread_with_spaces(eng, oracle):
    """the polled read-back kept its sequence word inside the page that also staged the 2 KiB copy of the
    byte bins -- bins[0x20] of a text with spaces left 1 (or the count of spaces) where the next post polled for its sequence
    number, the poll ended before the kernel had run and the build went on with stale round totals.  Deterministic for the first
    general build of a thread (sequence number 1) on a text with spaces below 65536 bytes, and whenever the count of spaces
    met the running sequence number.  Fresh threads (fresh thread-local state), general build forced, texts with 1 .. 12 spaces
    and ordinary text; through the host-pointer entry point and through the device entry point."""
    import threading
    rng = np.random.default_rng(20)
    texts = [_gen.english_like(20000, seed=77).tobytes()]
    for spaces in range(1, 13):
        body = bytearray(rng.integers(97, 123, 17000 + 371 * spaces, dtype=np.uint8).tobytes())
        for pos in rng.choice(len(body), size=spaces, replace=False).tolist():
            body[pos] = 0x20
        texts.append(bytes(body))
    errors = []

    def work(t, dev_entry):
        try:
            from suffix_amd import SuffixTable
            exp = oracle.sais(t)
            if dev_entry:
                from suffix_amd import device as sdev
                got = sdev.build_sa(torch.frombuffer(bytearray(t), dtype=torch.uint8).cuda(), engine=eng).cpu().numpy().view(np.uint32)
            else:
                got = SuffixTable(t, engine=eng).table()
            if not np.array_equal(got, exp):
                errors.append(("SA differs", len(t), t.count(b" "), dev_entry))
        except Exception as exc:                                       # noqa: BLE001
            errors.append((repr(exc), len(t), dev_entry))

    with _cases.general_build(eng):
        for t in texts:
            for dev_entry in (False, True):
                th = threading.Thread(target=work, args=(t, dev_entry))
                th.start()
                th.join()
    assert not errors, errors


def test_literals(eng, oracle, golden):
    _cases.literals(eng, oracle, golden)


def test_search_known_answers(eng, golden):
    _cases.search_known_answers(eng, golden)


def test_parts_roundtrip(eng):
    _cases.parts_roundtrip(eng)


@pytest.mark.parametrize("name", ["AP009048_10000", "AP009048_100000"])
def test_fasta_fixtures(eng, oracle, golden, fasta, name):
    _cases.fasta_fixture(eng, oracle, golden, fasta, name)


def test_random_small(eng, oracle):
    _cases.random_small(eng, oracle, iters=600, max_len=200, seed=2026)


def test_unicode(eng, oracle):
    _cases.unicode_strings(eng, oracle, iters=300, seed=9)


def test_structured(eng, oracle):
    _cases.structured(eng, oracle, scale=40)


def test_generated_medium(eng, oracle):
    _cases.generated(eng, oracle, n_dna=3_000_000, n_text=1_500_000)


def test_random_medium_sweep(eng, oracle):
    _cases.random_medium_sweep(eng, oracle, iters=60, max_len=300_000)


def test_index_directory_queries(eng, oracle):
    _cases.directory_queries(eng, oracle, device="cuda", scale=20)


def test_single_workgroup_build(eng, oracle):
    _cases.tiny_build(eng, oracle)


def test_suffix_tree_topology_and_doc_lookup(eng, oracle):
    _cases.suffix_tree_topology(eng, oracle, device="cuda", scale=3)


def test_suffix_tree_topology_at_scale(eng, oracle):
    """12 MB of DNA and of English-like text: lb / rb / node / parent / leaf_parent of every boundary against the
    oracle's linear restatement of to_suffix_tree (suffix_tree/src/lib.rs:392-505)."""
    import _gen
    nd = _cases.suffix_tree_at_scale(eng, oracle, _gen.dna(12_000_000, seed=77).tobytes())
    ne = _cases.suffix_tree_at_scale(eng, oracle, _gen.english_like(12_000_000).tobytes())
    assert nd > 4_000_000 and ne > 4_000_000, (nd, ne)


def test_fused_sa_lcp(eng, oracle):
    _cases.fused_lcp_tails(eng, oracle, iters=40, scale=50)


def test_planted_repeats_switch_text_to_rank_rounds(eng, oracle):
    _cases.planted_repeats(eng, oracle, 2_000_000)


def test_small_buckets_direct_ordering(eng, oracle):
    _cases.small_buckets(eng, oracle, scale=20)


def test_long_runs_and_repeats(eng, oracle):
    # worst cases for prefix doubling / PLCP: every round keeps every suffix active
    _cases.check_text(eng, oracle, b"a" * 50_000)   # (the oracle's LCP is quadratic here)
    _cases.check_text(eng, oracle, (b"ACGTTGCA" * 8 + b"N") * 3000)
    rep = _gen.english_like(40_000).tobytes()
    _cases.check_text(eng, oracle, rep * 5)


def test_dna_20mb_full_compare(eng, oracle):
    from suffix_amd import SuffixTable
    text = _gen.dna(20_000_000, seed=4242).tobytes()
    st = SuffixTable(text, engine=eng)
    exp = oracle.sais(text)
    assert hashlib.sha256(st.table().tobytes()).hexdigest() == hashlib.sha256(exp.tobytes()).hexdigest()
    assert np.array_equal(st.lcp_lens(), oracle.lcp_kasai(text, exp))
    stats = eng.build_stats()
    assert stats["sigma"] == 4 and stats["key_bits"] == 32 and stats["symbols_per_key"] == 16


def test_hybrid_initial_sort_56mb(eng, oracle):
    """>= 2^25 suffixes with a 32-bit key: two device-wide passes on the top 16 key bits, the 65536 sub-buckets
    sorted in LDS (k_bucket_sort).  The same text with 20000 copies of one 16-mer planted has nine sub-buckets of
    20000 suffixes (above what the LDS holds: gathered, sorted device-wide, copied back) and four of 5000 (the
    1024-thread geometry); a skewed text has most of its suffixes in such sub-buckets and takes the four-pass sort,
    and so does a text with a run that wraps two 16-bit counters of the histogram (noticed in its total).
    Complete SA and LCP against the oracle for all three, through the separate entries and the fused SA + LCP entry."""
    import torch
    from suffix_amd import device as sdev
    n = 56_000_000
    rng = np.random.default_rng(12)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    uniform = _gen.dna(n, seed=31)
    planted = uniform.copy()
    block = np.concatenate([np.tile(letters[rng.integers(0, 4, 16)], (20000, 1)), letters[rng.integers(0, 4, (20000, 16))]], axis=1)
    planted[1_000_000:1_000_000 + block.size] = block.reshape(-1)
    skewed = letters[rng.choice(4, size=n, p=[0.55, 0.05, 0.05, 0.35])]
    wrapped = uniform.copy()                                       # 250000 x "AC": longer than two of the histogram's stretches (229376
    wrapped[30_000_000:30_500_000] = np.tile(np.frombuffer(b"AC", dtype=np.uint8), 250000)   # positions): two 16-bit counters of a workgroup wrap
    for host, lds, over in ((uniform, True, False), (planted, True, True), (skewed, False, False), (wrapped, False, False)):
        text = torch.from_numpy(np.ascontiguousarray(host)).cuda()
        eng.profile(True); eng.profile_reset()
        sa = sdev.build_sa(text)
        torch.cuda.synchronize()
        names = {r["name"] for r in eng.profile_report()}
        eng.profile(False)
        # (round 6: without an oversized sub-bucket the LDS sort leaves tie bits -- bucket_sort_ties, k_tie_direct, no bucket pass over
        # sorted keys; with one it writes the sorted keys as before)
        assert "radix_hist16_text" in names and bool({"bucket_sort_lds", "bucket_sort_ties"} & names) == lds and ("oversize_gather" in names) == over, names
        if lds:
            assert ("bucket_sort_ties" in names) == (not over) and ("tie_direct" in names) == (not over) and ("groups_reduce" in names) == over, names
        exp = oracle.sais(host.tobytes())
        assert np.array_equal(sa.cpu().numpy().view(np.uint32), exp)
        sa2, lcp2 = sdev.build_sa_lcp(text)
        assert np.array_equal(sa2.cpu().numpy().view(np.uint32), exp)
        assert np.array_equal(lcp2.cpu().numpy().view(np.uint32), oracle.lcp_kasai(host.tobytes(), exp))
        del text, sa, sa2, lcp2


def test_compressed_keys_symbol_distributions(eng, oracle):
    """64-bit initial keys in the order-preserving prefix code (k_ht_keys) over byte distributions that stress the code
    construction: Zipf over 120 values (code words of 1 .. 12 bits side by side), two dominant symbols next to 90 rare
    ones (the length limit of 12 bits binds: the counts are floored), a symbol that occurs once, geometric counts, and
    uniform bytes over 97 / 256 values (no code saves 0.75 bits per symbol: fixed-width keys).  9 MB each, complete SA and
    LCP against the oracle, through the separate entries and the one-call entry."""
    import torch
    from suffix_amd import device as sdev
    # (65 .. 128 distinct bytes = 7-bit symbols: texts of more than 2^23 bytes take 64-bit keys -- choose_key, sfx_sa.hip)
    n = 9_000_000
    rng = np.random.default_rng(2718)
    def draw(p):
        p = np.asarray(p, dtype=np.float64)
        return rng.choice(len(p), size=n, p=p / p.sum()).astype(np.uint8)
    zipf = draw(1.0 / np.arange(1, 121) ** 1.2)
    two = draw([0.46, 0.46] + [0.08 / 90] * 90) + 40
    once = zipf.copy(); once[n // 3] = 255
    geom = draw(0.9 ** np.arange(0, 100)) + 1
    words = _gen.english_like(n)
    cases = ((zipf, True), (two, True), (once, True), (geom, True), (words, True),
             (_gen.uniform_bytes(n, 97, 5, base=20), False), (rng.integers(0, 256, n, dtype=np.uint8), False))
    for host, compressed in cases:
        host = np.ascontiguousarray(host)
        text = torch.from_numpy(host).cuda()
        eng.profile(True); eng.profile_reset()
        sa = sdev.build_sa(text)
        torch.cuda.synchronize()
        names = {r["name"] for r in eng.profile_report()}
        eng.profile(False)
        st = eng.build_stats()
        assert ("ht_keys" in names) == compressed and (st["key_bits"] == 64 or not compressed), (sorted(names), st)
        exp = oracle.sais(host.tobytes())
        assert np.array_equal(sa.cpu().numpy().view(np.uint32), exp)
        want = oracle.lcp_kasai(host.tobytes(), exp)
        assert np.array_equal(sdev.build_lcp(text, sa).cpu().numpy().view(np.uint32), want)
        sa2, lcp2 = sdev.build_sa_lcp(text)
        assert np.array_equal(sa2.cpu().numpy().view(np.uint32), exp) and np.array_equal(lcp2.cpu().numpy().view(np.uint32), want)
        del text, sa, sa2, lcp2


def test_context_codes_and_their_pilot_68mb(eng, oracle):
    """Context codes (round 6, k_ht_keys_ctx) at the smallest size that considers them (2^26 bytes): mixed-script UTF-8 of
    Zipf words leaves most of the pilot sort's 2^22 suffixes tied and takes them (bigram pass, stats.reserved bit 1 through the
    key kernel's profile name staying `ht_keys`); the same generator's bytes drawn as independent code points (round 1's form
    of config 5) has the same symbol statistics and no ties: the pilot keeps the order-0 keys.  Complete SA against the
    oracle for both, the LCP array of the first."""
    import torch
    from suffix_amd import device as sdev
    n = (1 << 26) + 1_000_000
    words = np.ascontiguousarray(_gen.utf8_mixed(n))
    rng = np.random.default_rng(99)
    # independent code points of the four scripts, encoded: no word is ever repeated
    cps = np.concatenate([rng.integers(0x61, 0x7B, n // 8), rng.integers(0x410, 0x450, n // 8), rng.integers(0x4E00, 0x9FFF, n // 8),
                          rng.integers(0x1F300, 0x1F600, n // 16)])
    rng.shuffle(cps)
    indep = np.frombuffer("".join(map(chr, cps.tolist())).encode("utf-8"), dtype=np.uint8).copy()      # (n bytes: 1 + 2 + 3 eighths, 4 sixteenths)
    assert len(indep) >= 1 << 26
    for host, taken in ((words, True), (indep, False)):
        text = torch.from_numpy(host).cuda()
        eng.profile(True); eng.profile_reset()
        sa = sdev.build_sa(text)
        torch.cuda.synchronize()
        rep = {r["name"]: r for r in eng.profile_report()}
        eng.profile(False)
        st = eng.build_stats()
        # the pilot sorts before the text does: two key kernels, sixteen passes; the bigram pass only behind a pilot that says "tied"
        assert st["key_bits"] == 64 and rep["ht_keys"]["launches"] == 2 and rep["radix_scatter_u64"]["launches"] == 16, (st, sorted(rep))
        assert ("bigram_hist" in rep) == taken, sorted(rep)
        exp = oracle.sais(host.tobytes())
        assert np.array_equal(sa.cpu().numpy().view(np.uint32), exp)
        if taken:
            assert np.array_equal(sdev.build_lcp(text, sa).cpu().numpy().view(np.uint32), oracle.lcp_kasai(host.tobytes(), exp))
        del text, sa


@pytest.mark.parametrize("sigma", [2, 5, 16])
def test_hybrid_initial_sort_other_alphabets(eng, sigma):
    """The hybrid route on keys of 32 one-bit symbols and 8 four-bit symbols, 56 * 10^6 suffixes each, through the
    size-independent property gate; the fused SA + LCP entry gives the same arrays.  Five symbols (3 bits, 10 per
    word) do not separate 56 * 10^6 suffixes in a 32-bit key: that text takes 64-bit keys and never sees the hybrid
    route (30-bit keys, whose LDS digits have 8 + 6 bits, are covered on the emulator)."""
    import sys, os
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from suffix_amd import device as sdev
    text = torch.from_numpy(_gen.uniform_bytes(56_000_000, sigma, 40 + sigma, base=65)).cuda()
    eng.profile(True); eng.profile_reset()
    sa = sdev.build_sa(text)
    torch.cuda.synchronize()
    names = {r["name"] for r in eng.profile_report()}
    eng.profile(False)
    # (a binary text repeats inside 32 symbols all the time, but its top-16-bit sub-buckets are as even as any)
    assert ("radix_hist16_text" in names) == (sigma != 5) and ("bucket_sort_ties" in names) == (sigma != 5), names
    ok, how = bench.verify_sa_on_device(torch, sdev, text, sa)
    assert ok, how
    sa2, lcp2 = sdev.build_sa_lcp(text)
    assert torch.equal(sa2, sa)
    assert torch.equal(lcp2, sdev.build_lcp(text, sa))


def test_device_resident_100mb_properties(eng):
    """BASELINE config 2 at full size, checked through size-independent properties."""
    import sys, os
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from suffix_amd import device as sdev
    text = torch.from_numpy(_gen.dna(100_000_000)).cuda()
    sa = sdev.build_sa(text)
    torch.cuda.synchronize()
    ok, how = bench.verify_sa_on_device(torch, sdev, text, sa)
    assert ok, how
    # idempotence: a second build gives the same array
    sa2 = sdev.build_sa(text)
    assert torch.equal(sa, sa2)


def test_batched_queries_device(eng, oracle):
    import torch
    from suffix_amd import device as sdev
    text = _gen.utf8_mixed(400_000).tobytes()
    exp = oracle.sais(text)
    rng = np.random.default_rng(3)
    qs = []
    for _ in range(2000):
        a = int(rng.integers(0, len(text)))
        q = text[a:a + int(rng.integers(1, 20))]
        if rng.random() < 0.3:
            q = q[:-1] + bytes([q[-1] ^ 0x55])
        qs.append(q)
    off = np.zeros(len(qs) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(q) for q in qs])
    t = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
    sa = sdev.build_sa(t)
    assert np.array_equal(sa.cpu().numpy().view(np.uint32), exp)
    qb = torch.frombuffer(bytearray(b"".join(qs)), dtype=torch.uint8).cuda()
    s, e, f, a = sdev.query_batch(t, sa, qb, torch.from_numpy(off).cuda())
    s, e, f = s.cpu().numpy(), e.cpu().numpy(), f.cpu().numpy()
    for k, q in enumerate(qs):
        assert (int(s[k]), int(e[k])) == oracle.positions(text, exp, q)
        assert bool(f[k]) == (q in text)


def test_partitioned_build_single_rank_slices(eng, oracle):
    """The multi-GPU range build, driven for 3 'virtual ranks' on one GPU: the
    slices concatenate to the oracle's SA."""
    import ctypes
    import torch
    from suffix_amd import dist as sdist
    from suffix_amd.device import _p
    for text in (_gen.dna(500_000, seed=5).tobytes() + b"AAAA",
                 _gen.english_like(300_000).tobytes()):
        exp = oracle.sais(text)
        n = len(text)
        t = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
        bb = torch.zeros(256, dtype=torch.int64, device="cuda")
        eng.check(eng.lib.sfx_byte_histogram_dev(_p(t), 0, n, _p(bb), None), "bh")
        kb = torch.zeros(1 << 14, dtype=torch.int64, device="cuda")
        eng.check(eng.lib.sfx_key_histogram_dev(_p(t), n, 0, n, _p(bb), 14, _p(kb), None), "kh")
        assert int(kb.sum()) == n
        pieces = []
        for lo, hi, off, cnt in sdist.plan_ranges(kb.cpu(), 3):
            part = torch.empty(max(cnt, 1), dtype=torch.int32, device="cuda")
            ws = torch.empty(int(eng.lib.sfx_sa_range_workspace_bytes(n, max(cnt, 1))), dtype=torch.uint8, device="cuda")
            got = ctypes.c_uint64(0)
            eng.check(eng.lib.sfx_build_sa_range_u32_dev(_p(t), n, _p(bb), 14, lo, hi, max(cnt, 1), _p(part),
                                                         ctypes.byref(got), _p(ws), ws.numel(), None), "range")
            assert int(got.value) == cnt
            pieces.append(part[:cnt].cpu().numpy().view(np.uint32))
        assert np.array_equal(np.concatenate(pieces), exp)
        # per-slice LCP (direct comparison, :348-361) and per-slice queries, then the u64 widening
        from suffix_amd import device as sdev
        exp_lcp = oracle.lcp_quadratic(text, exp)
        qs = [text[1000:1007], text[-9:], b"zzzz", text[77:79], text[5:6]]
        qb = torch.frombuffer(bytearray(b"".join(qs)), dtype=torch.uint8).cuda()
        qoff = torch.tensor(np.concatenate([[0], np.cumsum([len(q) for q in qs])]), dtype=torch.int64).cuda()
        off, prev, total, gstart = 0, None, np.zeros(len(qs), dtype=np.int64), np.full(len(qs), -1, dtype=np.int64)
        for piece in pieces:
            part = torch.from_numpy(piece.view(np.int32)).cuda()
            lcp = sdev.build_lcp_range(t, part, prev, engine=eng).cpu().numpy().view(np.uint32)
            assert np.array_equal(lcp, exp_lcp[off:off + piece.size])
            s_, e_, _f, _a = sdev.query_batch_range(t, part, qb, qoff, engine=eng)
            s_, e_ = s_.cpu().numpy().astype(np.int64), e_.cpu().numpy().astype(np.int64)
            for k in range(len(qs)):
                if e_[k] > s_[k]:
                    if gstart[k] < 0:
                        gstart[k] = off + s_[k]
                    total[k] += e_[k] - s_[k]
            w64 = sdev.widen_u64(part, engine=eng).cpu().numpy()
            assert w64.dtype == np.int64 and np.array_equal(w64.astype(np.uint32), piece)
            if piece.size:
                prev = int(piece[-1])
            off += piece.size
        for k, q in enumerate(qs):
            es, ee = oracle.positions(text, exp, q)
            assert (int(total[k]) == ee - es) and (total[k] == 0 or int(gstart[k]) == es), (q, gstart[k], total[k], es, ee)


def test_range_build_hybrid_slices(eng, oracle):
    """Slices of >= 2^25 suffixes take the hybrid initial sort too (keys relative to the first key of the range, the
    histogram of their top 16 bits counted from the emitted elements): 110 MB of DNA for 3 virtual ranks, and 40 MB for
    one (the whole key space: the text-fed route of the full build); the slices concatenate to the oracle's suffix array."""
    import ctypes
    import torch
    from suffix_amd import dist as sdist
    from suffix_amd.device import _p
    for n, nranges in ((110_000_000, 3), (40_000_000, 1)):
        text = _gen.dna(n, seed=77 + nranges)
        t = torch.from_numpy(text).cuda()
        bb = torch.zeros(256, dtype=torch.int64, device="cuda")
        eng.check(eng.lib.sfx_byte_histogram_dev(_p(t), 0, n, _p(bb), None), "bh")
        kb = torch.zeros(1 << 14, dtype=torch.int64, device="cuda")
        eng.check(eng.lib.sfx_key_histogram_dev(_p(t), n, 0, n, _p(bb), 14, _p(kb), None), "kh")
        pieces = []
        eng.profile(True); eng.profile_reset()
        for lo, hi, off, cnt in sdist.plan_ranges(kb.cpu(), nranges):
            part = torch.empty(max(cnt, 1), dtype=torch.int32, device="cuda")
            ws = torch.empty(int(eng.lib.sfx_sa_range_workspace_bytes(n, max(cnt, 1))), dtype=torch.uint8, device="cuda")
            got = ctypes.c_uint64(0)
            eng.check(eng.lib.sfx_build_sa_range_u32_dev(_p(t), n, _p(bb), 14, lo, hi, max(cnt, 1), _p(part),
                                                         ctypes.byref(got), _p(ws), ws.numel(), None), "range")
            assert int(got.value) == cnt
            pieces.append(part[:cnt].cpu().numpy().view(np.uint32))
            del ws, part
        names = {r["name"] for r in eng.profile_report()}
        eng.profile(False)
        # (one rank = the whole key space: no filter, the text-fed route of the full build)
        assert ("radix_hist16_elems" if nranges > 1 else "radix_hist16_text") in names and "bucket_sort_ties" in names, names
        assert ("range_emit" in names) == (nranges > 1), names
        assert np.array_equal(np.concatenate(pieces), oracle.sais(text.tobytes()))
        del t
        torch.cuda.empty_cache()


@pytest.mark.parametrize("nranges", [2, 9])
def test_range_build_many_ranges(eng, oracle, nranges):
    """Dense (LDS-compacted) and sparse (direct-store) tiles of the range filter, raw and packed input."""
    _cases.range_slices(eng, oracle, _gen.dna(1_000_003, seed=8).tobytes(), nranges, device="cuda", packed=True)
    _cases.range_slices(eng, oracle, _gen.dna(300_001, seed=9).tobytes(), nranges, device="cuda")
    _cases.range_slices(eng, oracle, _gen.english_like(400_000).tobytes(), nranges, device="cuda", packed=(nranges == 9))


def test_lcp_routes_at_scale(eng, oracle):
    """The three routes of lcp_lens above the 1 MB threshold: direct on packed symbols, direct on
    raw bytes, and a pair that reaches the cap of the direct pass (Phi/PLCP then redoes the array)."""
    from suffix_amd import SuffixTable

    def kernels_of_lcp(text):
        st = SuffixTable(text, engine=eng)
        exp_sa = oracle.sais(text)
        assert np.array_equal(st.table(), exp_sa)
        eng.profile(True)
        eng.profile_reset()
        got = st.lcp_lens()
        names = [r["name"] for r in eng.profile_report()]
        eng.profile(False)
        assert np.array_equal(got, oracle.lcp_kasai(text, exp_sa))
        return names

    d = _gen.dna(1_500_000, seed=77).tobytes()
    k = kernels_of_lcp(d)
    assert "lcp_windows_packed" in k and "plcp" not in k, k
    k = kernels_of_lcp(_gen.utf8_mixed(1_200_000).tobytes())
    assert "lcp_windows" in k and "plcp" not in k, k
    k = kernels_of_lcp(d[:1_000_000] + b"A" * 3000 + d[1_000_000:])
    assert "lcp_windows_packed" in k and "plcp" in k, k
    k = kernels_of_lcp((_gen.english_like(30_000).tobytes()) * 40)         # the sample says: repetitive
    assert "lcp_windows" not in k and "lcp_windows_packed" not in k and "plcp" in k, k
