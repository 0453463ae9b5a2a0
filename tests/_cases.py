"""Parity cases shared by the CPU (emulator) and GPU test modules.  Every
function takes an `Engine` (suffix_amd._lib.Engine): the GPU tests pass the
product's libsuffix_hip.so, the CPU tests the same sources compiled against the
fiber emulator (tests/emu).  Expected values always come from the oracle
(oracle/, the C restatement of the reference) or from tests/golden/."""
import hashlib
import random

import numpy as np

import _gen
from suffix_amd import SuffixTable


def sha_u32(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u4").tobytes()).hexdigest()


TINY_OPT = 1                                                            # SFX_OPT_TINY_MAX (include/suffix_hip.h)


class general_build:
    """with general_build(eng): texts of any length take the general build (sfx_set_option(SFX_OPT_TINY_MAX, 0)); the
    single-workgroup build of texts of up to 16384 bytes (sfx_tiny.hip) is restored on exit."""
    def __init__(self, eng):
        self.eng = eng

    def __enter__(self):
        self.old = int(self.eng.lib.sfx_get_option(TINY_OPT))
        assert self.eng.lib.sfx_set_option(TINY_OPT, 0) == 0
        return self

    def __exit__(self, *exc):
        assert self.eng.lib.sfx_set_option(TINY_OPT, self.old) == 0
        return False


def check_text(eng, orc, text, lcp=True, queries=()):
    """Build SA (+LCP, + queries) with the engine; compare bit-exactly with the oracle.  A text short enough for the
    single-workgroup build is built both ways: by it (the default) and by the general build."""
    st = SuffixTable(text, engine=eng)
    exp = orc.sais(st._text)
    assert st.len() == len(st._text)                                    # prop_length
    assert np.array_equal(st.table(), exp), f"SA mismatch on {st._text[:40]!r}..."

    if lcp:
        exp_lcp = orc.lcp_quadratic(st._text, exp)
        assert np.array_equal(st.lcp_lens(), exp_lcp)
        st2, lcp2 = SuffixTable.new_with_lcp(text, engine=eng)          # the fused entry point: same two arrays
        assert np.array_equal(st2.table(), exp) and np.array_equal(lcp2, exp_lcp), "fused SA+LCP"
    if queries:
        s, e = st.positions_batch(queries)
        found, anyp = st.contains_batch(queries)
        for k, q in enumerate(queries):
            es, ee = orc.positions(st._text, exp, q)
            assert (int(s[k]), int(e[k])) == (es, ee), (q, int(s[k]), int(e[k]), es, ee)
            ea = orc.any_position(st._text, exp, q)
            assert bool(found[k]) == (ea is not None)
            if ea is not None:
                qb = q.encode() if isinstance(q, str) else bytes(q)
                a = int(anyp[k])
                assert st._text[a:a + len(qb)] == qb                    # "arbitrary" occurrence
            else:
                assert int(anyp[k]) == 0xFFFFFFFF
    # (last, so that eng.build_stats() after check_text describes the general build, as the callers that look at it expect)
    if 2 <= len(st._text) <= int(eng.lib.sfx_get_option(TINY_OPT)):
        with general_build(eng):
            assert np.array_equal(SuffixTable(text, engine=eng).table(), exp), f"SA mismatch (general build) on {st._text[:40]!r}..."
            if lcp:
                st3, lcp3 = SuffixTable.new_with_lcp(text, engine=eng)  # ... and the fused arrays of the general build
                assert np.array_equal(st3.table(), exp) and np.array_equal(lcp3, orc.lcp_quadratic(st._text, exp)), "fused SA+LCP (general build)"
    return st


def tiny_build(eng, orc):
    """The single-workgroup build (sfx_tiny.hip, texts of up to 16384 bytes): which texts it finishes, which it hands to the
    general build, and that both give the reference's table.  Runs of equal keys: repeats in 2 .. 32 copies (ordered by
    direct comparison inside the kernel), in more than 32 copies and unary / periodic texts (given up: general build),
    suffixes that end inside the key window and look like runs of the smallest symbol."""
    rng = np.random.default_rng(77)
    limit = int(eng.lib.sfx_get_option(TINY_OPT))
    assert limit == 16384

    def build(text):
        eng.profile(True); eng.profile_reset()
        sa = SuffixTable(text, engine=eng).table()
        names = {r["name"] for r in eng.profile_report()}
        eng.profile(False)
        assert np.array_equal(sa, orc.sais(text)), (len(text), text[:30])
        return names

    dna = _gen.dna(16384, seed=11).tobytes()
    finished = [b"ab", b"ba", b"aab", b"banana", b"mississippi", dna, dna[:10001], dna[:1023], dna[:1025],
                _gen.english_like(1200, seed=5).tobytes(), bytes(rng.integers(0, 256, 4000, dtype=np.uint8)),
                _gen.uniform_bytes(5000, 3, 4, base=97).tobytes(), _gen.uniform_bytes(6000, 16, 9, base=65).tobytes(),
                _gen.uniform_bytes(4000, 17, 9, base=65).tobytes(), "\u2603abc\u2603".encode(),
                # repeats of 40 .. 300 symbols in 2 .. 30 copies: runs of equal keys ordered inside the kernel
                b"".join(dna[a:a + ln] + bytes([66 + k % 3]) for k, (a, ln) in enumerate(zip(rng.integers(0, 9000, 30).tolist(), rng.integers(40, 300, 30).tolist()))) * 3,
                # the smallest symbol in runs at the end and before it: suffixes that end inside the key against real runs
                dna[:3000].replace(b"C", b"A") + b"A" * 20, b"A" * 12 + dna[:2000] + b"A" * 14]
    for t in finished:
        names = build(t)
        assert "tiny_sa" in names and "groups_reduce" not in names, (len(t), sorted(names))
    given_up = [b"a" * 5000, b"ab" * 4000 + b"a", _gen.fibonacci_string(19)[:9000], (dna[:200] + b"N") * 60,
                _gen.english_like(9000, seed=5).tobytes(), bytes(rng.integers(0, 256, 7000, dtype=np.uint8))]   # (8-bit symbols, > 4096 of them)
    for t in given_up:
        names = build(t)
        assert "tiny_sa" in names and "groups_reduce" in names, (len(t), sorted(names))
    names = build(dna + b"A")                                              # one byte too long: the general build at once
    assert "tiny_sa" not in names and "groups_reduce" in names
    with general_build(eng):
        assert "tiny_sa" not in build(dna[:5000])
    for t in finished[:8]:
        check_text(eng, orc, t)                                            # (+ LCP, + the fused entry point over the tiny build)


def literals(eng, orc, golden):
    # tests/tests.rs:22-70 + parts()
    for s, exp in golden["sa_literals"].items():
        st = SuffixTable(s, engine=eng)
        assert st.table().tolist() == exp["sa"], s
        assert st.lcp_lens().tolist() == exp["lcp"], s
        naive = SuffixTable.from_parts(s, orc.naive_sa(s), engine=eng)   # new == new_naive
        assert st == naive
        assert st == SuffixTable.new_naive(s, engine=eng)                # ... as tests/tests.rs:18-20 writes it (host mirror's own)


def search_known_answers(eng, golden):
    # tests/tests.rs:100-168, :181-213, doc-tests
    for text, query, pos, found in golden["search"]:
        st = SuffixTable(text, engine=eng)
        assert st.positions(query).tolist() == pos, (text, query)
        assert st.contains(query) == found, (text, query)
        ap = st.any_position(query)
        assert (ap is not None) == found
        if found:
            assert ap in pos


def parts_roundtrip(eng):
    # tests/tests.rs:170-179
    sa = SuffixTable("poëzie", engine=eng)
    text, table = sa.into_parts()
    sa3 = SuffixTable.from_parts(text, table, engine=eng)
    assert sa == sa3
    try:
        SuffixTable.from_parts("abc", np.zeros(2, dtype=np.uint32), engine=eng)
        raise RuntimeError("from_parts accepted mismatched lengths")
    except AssertionError:
        pass
    # from_parts is unchecked by contract (:105-107), but a table entry >= n must come back as an error
    # from the engine, never as an out-of-bounds device access
    from suffix_amd._lib import SuffixHipError
    bad = SuffixTable.from_parts("abcdef", np.array([5, 4, 9, 2, 1, 0], dtype=np.uint32), engine=eng)
    for call in (lambda: bad.positions("a"), lambda: bad.lcp_lens()):
        try:
            call()
            raise RuntimeError("engine accepted a table with an entry >= n")
        except SuffixHipError:
            pass


def fasta_fixture(eng, orc, golden, fasta, name):
    g = golden["fixtures"][name]
    text = fasta[name]
    st = SuffixTable(text, engine=eng)
    assert sha_u32(st.table()) == g["sha256_sa"]
    assert st.table()[:6].tolist() == g["sa_head"] and st.table()[-4:].tolist() == g["sa_tail"]
    lcp = st.lcp_lens()
    assert sha_u32(lcp) == g["sha256_lcp"] and int(lcp.max()) == g["max_lcp"]
    # tests/bench.rs queries
    assert st.positions("ACTTACGTGTCTGC").tolist() == [1825]
    assert st.positions("H").size == 0 and not st.contains("H")
    assert st.positions("C").size == text.count(b"C") and st.contains("C")
    assert sorted(st.positions("C").tolist()) == [i for i in range(len(text)) if text[i:i + 1] == b"C"]


def random_small(eng, orc, iters, max_len, seed):
    # mirrors prop_naive_equals_sais / prop_matches_naive / prop_contains / prop_positions
    rnd = random.Random(seed)
    for _ in range(iters):
        n = rnd.randint(0, max_len)
        sigma = rnd.choice([1, 2, 3, 4, 5, 16, 20, 50, 97, 256])
        t = bytes(rnd.randrange(sigma) for _ in range(n))
        qs = []
        for _ in range(4):
            if n and rnd.random() < 0.7:
                a = rnd.randrange(n)
                qs.append(t[a:a + rnd.randint(1, 6)])
            else:
                qs.append(bytes(rnd.randrange(sigma) for _ in range(rnd.randint(0, 4))))
        check_text(eng, orc, t, queries=qs)


def unicode_strings(eng, orc, iters, seed):
    rnd = random.Random(seed)
    pools = [range(0x20, 0x7F), range(0xA0, 0x250), range(0x4E00, 0x4E40),
             range(0x1F300, 0x1F320), [0, 0x2603]]
    for _ in range(iters):
        n = rnd.randint(0, 24)
        s = "".join(chr(rnd.choice(list(rnd.choice(pools)))) for _ in range(n))
        c = chr(rnd.randrange(128))
        st = check_text(eng, orc, s, queries=[c, s[:2], s[-3:]])
        b = s.encode("utf-8")
        assert st.contains(c) == (c.encode() in b)                      # prop_contains
        exp = [i for i in range(len(b)) if b[i:i + 1] == c.encode()]
        assert sorted(st.positions(c).tolist()) == exp                  # prop_positions


def structured(eng, orc, scale):
    """Adversarial shapes: long runs, periodic, Fibonacci/Thue-Morse (deep recursion in
    SA-IS), texts ending in the smallest symbol (zero-padded key ties)."""
    cases = [b"a" * (37 * scale), b"ab" * (29 * scale), b"abc" * (17 * scale) + b"a",
             _gen.fibonacci_string(8 + scale.bit_length()), _gen.thue_morse(100 * scale).tobytes(),
             bytes(range(256)) * 2, bytes(reversed(range(256))),
             b"\x00" * (10 * scale) + b"\xff" * (10 * scale),
             b"\xff" * (10 * scale) + b"\x00" * (10 * scale),
             b"ACGT" * (8 * scale) + b"AAAA", b"TTTTGGGGCCCCAAAA" * scale + b"A" * 40,
             b"A" * 33 + b"C" + b"A" * 34, (b"x" * 70 + b"y") * (2 * scale)]
    for t in cases:
        check_text(eng, orc, t, queries=[t[:3], t[-5:], b"zz", t[len(t) // 2:len(t) // 2 + 9]])


def planted_repeats(eng, orc, n):
    """Mostly-random text with a few long planted repeats: the initial sort resolves
    almost everything (text-first round), the repeats then force the switch to rank
    rounds (ISA rebuilt from SA + unresolved buckets)."""
    d = _gen.dna(n, seed=31).tobytes()
    t = d + d[n // 3:n // 3 + 400] + b"G" + d[n // 2:n // 2 + 90] + d[n // 3 + 10:n // 3 + 300]
    check_text(eng, orc, t, queries=[d[n // 3:n // 3 + 50], d[n // 2:n // 2 + 91]])
    st = eng.build_stats()
    assert st["rounds"] >= 3, st
    u = _gen.uniform_bytes(n, 256, 5).tobytes()
    check_text(eng, orc, u + u[100:700] + b"\x00\x00", queries=[u[100:130]])


def generated(eng, orc, n_dna, n_text):
    check_text(eng, orc, _gen.dna(n_dna).tobytes(), queries=[b"ACGT", b"TTTTTTTTTTTTTTTTTTTTTT", b"G"])
    check_text(eng, orc, _gen.english_like(n_text).tobytes(), queries=[b"the", b" a ", b". T", b"qzx"])
    check_text(eng, orc, _gen.utf8_mixed(n_text).tobytes(),
               queries=["日".encode(), b" ", "я".encode(), b"\xf0\x9f"])
    check_text(eng, orc, _gen.uniform_bytes(n_text, 256, 11).tobytes(), queries=[b"\x00", b"\xff\xff"])
    check_text(eng, orc, _gen.uniform_bytes(n_text, 2, 12, base=ord("a")).tobytes(), queries=[b"abab", b"bbbbbbbbb"])


def small_buckets(eng, orc, scale=1):
    """Planted short repeats: after the initial sort almost every unresolved bucket has
    2-3 members, which the engine orders by direct comparison (k_small_groups); very long
    repeats exceed its comparison depth and must fall through to the radix rounds; a
    repeat that runs into the end of the text exercises "shorter sorts first"."""
    rng = np.random.default_rng(5)

    def planted(base, nrep, lo, hi, copies=1):
        t = bytearray(base)
        for _ in range(nrep):
            length = int(rng.integers(lo, hi))
            p = int(rng.integers(0, len(base) - length))
            for _c in range(copies):
                t += base[p:p + length] + bytes([base[int(rng.integers(0, len(base)))]])
        return bytes(t)

    dna = _gen.dna(6000 * scale).tobytes()
    cases = {
        "dna pairs": (planted(dna, 80 * scale, 18, 40), True),
        "dna triples": (planted(dna, 40 * scale, 18, 60, 2), True),
        "dna deep": (planted(dna, 6, 300, 400), True),
        "bytes pairs": (planted(_gen.uniform_bytes(5000 * scale, 200, 3).tobytes(), 80 * scale, 10, 30), True),
        "ascii": (planted(_gen.uniform_bytes(5000 * scale, 20, 4).tobytes(), 120 * scale, 12, 50, 2), True),
        "runs into the end": (dna[:3000] + dna[:40], True),
    }
    for name, (text, expect_direct) in cases.items():
        check_text(eng, orc, text)
        st = eng.build_stats()
        if expect_direct:
            assert st["small_bucket_resolved"] > 0, (name, st)
    # the deep repeats cannot all be settled within the comparison depth
    check_text(eng, orc, cases["dna deep"][0], lcp=False)
    assert eng.build_stats()["rounds"] > 0


def random_medium_sweep(eng, orc, iters=40, max_len=200_000, seed=77):
    """Random texts of random length, alphabet and repeat structure (planted copies, periodic stretches, runs of
    the smallest symbol at the end), each through new(), lcp_lens(), the fused entry and a few queries -- the
    sizes at which refinement rounds, LDS bucket sorts and the segmented sort all take part."""
    rng = np.random.default_rng(seed)
    for it in range(iters):
        n = int(rng.integers(1, max_len))
        sigma = int(rng.choice([2, 3, 4, 5, 16, 20, 64, 66, 130, 256]))
        body = rng.integers(0, sigma, n, dtype=np.uint8)
        if sigma < 200:
            body = body + np.uint8(rng.integers(0, 256 - sigma))
        t = bytearray(body.tobytes())
        kind = it % 5
        if kind == 1 and n > 50:                               # planted copies of random stretches
            for _ in range(int(rng.integers(1, 6))):
                a = int(rng.integers(0, n - 20)); ln = int(rng.integers(10, min(5000, n - a)))
                t += t[a:a + ln]
        elif kind == 2 and n > 10:                             # a long periodic stretch
            per = bytes(t[:int(rng.integers(1, 8))])
            t += per * int(rng.integers(10, 3000))
        elif kind == 3:                                        # ends in a run of the smallest symbol
            t += bytes([min(t)]) * int(rng.integers(1, 200))
        elif kind == 4 and n > 100:                            # few distinct "words"
            words = [bytes(t[i:i + int(rng.integers(2, 9))]) for i in rng.integers(0, n - 10, 12)]
            t = bytearray(b" ".join(words[int(k)] for k in rng.integers(0, 12, n // 4)))
        t = bytes(t)
        qs = [t[int(a):int(a) + int(rng.integers(1, 12))] for a in rng.integers(0, len(t), 4)] + [b"\x00", t[-3:]]
        check_text(eng, orc, t, queries=qs)


def fused_lcp_tails(eng, orc, iters=40, scale=1):
    """sfx_build_sa_lcp_u32 on texts whose initial key sort separates most suffixes (the LCP of those
    pairs is read off the sorted keys) and whose END is made of the smallest symbol: suffixes shorter
    than the key have zero-padded keys that overstate the common prefix, which the tail fix-up redoes."""
    rng = np.random.default_rng(3)
    fused = 0
    tiny_was = int(eng.lib.sfx_get_option(TINY_OPT))
    assert eng.lib.sfx_set_option(TINY_OPT, 0) == 0                     # (the fused LCP of the general build is what is tested)
    for it in range(iters):
        n = int(rng.integers(200, 4000)) * scale
        base = _gen.dna(n, seed=100 + it).tobytes()
        tail = [b"", b"A" * int(rng.integers(1, 40)), b"CA" + b"A" * int(rng.integers(1, 20)), b"AAAC", b"T" * 5][it % 5]
        t = base + tail
        st, lcp = SuffixTable.new_with_lcp(t, engine=eng)
        fused += eng.build_stats()["active_after_initial"] * 4 <= len(t)
        exp = orc.sais(t)
        assert np.array_equal(st.table(), exp)
        assert np.array_equal(lcp, orc.lcp_quadratic(t, exp)), (it, len(t))
    for it in range(iters // 3):
        t = _gen.uniform_bytes(3000 * scale, 256, 500 + it).tobytes() + bytes(int(rng.integers(0, 12)))
        st, lcp = SuffixTable.new_with_lcp(t, engine=eng)
        exp = orc.sais(t)
        assert np.array_equal(st.table(), exp) and np.array_equal(lcp, orc.lcp_quadratic(t, exp))
    assert fused >= iters // 2, fused
    fused_lcp_short_suffix_ties(eng, orc)
    assert eng.lib.sfx_set_option(TINY_OPT, tiny_was) == 0
    assert SuffixTable.new_with_lcp(b"", engine=eng)[1].size == 0
    assert SuffixTable.new_with_lcp(b"x", engine=eng)[1].tolist() == [0]


def fused_lcp_short_suffix_ties(eng, orc):
    """A suffix that ends inside the key of a text round ties, through its zero padding, with a longer suffix
    that goes on with the smallest symbol; the round leaves the two in one class, the short one last in list
    order.  The pair (class, next class) must not take its LCP from that member's length: which member ends up
    next to the following class is only known when the class is resolved (found on 40 MB of skewed DNA, where
    the text happened to end in the first 18 symbols of an earlier 25-symbol repeat)."""
    rng = np.random.default_rng(8)
    for wlen, sym in ((16, b"ACGT"), (8, bytes(range(97, 97 + 20)))):          # 32-bit keys of 16 symbols; 64-bit keys of 8
        w = bytes(rng.choice(list(sym[1:]), wlen).tolist())                    # a word without the smallest symbol
        low = sym[:1]
        parts = []
        for _ in range(80):                                                    # one big bucket: a text round sorts it
            parts.append(w + bytes(rng.choice(list(sym), 40).tolist()))
        for k in (1, 2, 5):                                                    # long suffixes: w + x + low * 13 + ...; w + x + low * 7 + high
            x = bytes(rng.choice(list(sym[1:]), k).tolist())
            parts.append(w + x + low * 13 + sym[-1:] * 3)
            parts.append(w + x + low * 7 + sym[-1:] + bytes(rng.choice(list(sym), 9).tolist()))
            t = b"".join(parts) + w + x                                        # ... and the text ends in w + x
            st, lcp = SuffixTable.new_with_lcp(t, engine=eng)
            exp = orc.sais(t)
            assert np.array_equal(st.table(), exp)
            got, want = lcp, orc.lcp_quadratic(t, exp)
            bad = np.flatnonzero(got != want)
            assert bad.size == 0, (wlen, k, bad[:4], got[bad[:4]], want[bad[:4]])


def directory_queries(eng, orc, device="cpu", scale=1):
    """The resident index with its bucket directory (sfx_index_create_dev / sfx_index_query_dev) against the
    undirected search and the oracle: queries shorter than, equal to and longer than the directory key,
    bytes the text does not contain, queries ending in the smallest symbol, texts ending in it."""
    import torch

    from suffix_amd import device as sdev
    rng = np.random.default_rng(11)
    texts = [_gen.dna(30000 * scale, seed=5).tobytes() + b"AAAA", _gen.english_like(20000 * scale).tobytes(),
             _gen.utf8_mixed(20000 * scale).tobytes(), bytes(range(256)) * (8 * scale) + b"\x00\x00",
             b"a" * 300, b"ab" * 200 + b"a", _gen.uniform_bytes(5000 * scale, 3, 9, base=65).tobytes() + b"AA",
             # the 24-byte keys of the B+tree: runs of the padding bytes 0x00 / 0xFF, long shared prefixes
             b"\xff" * 40 + b"\x00" * 40 + b"\xff" * 17 + b"\x00" * 9 + b"\xff" * 16 + b"\x00" * 25 + b"\xff" * 24,
             (b"abcdefgh" * 40 + b"abcdefgx") * 5 + b"abcdefghabcdefgh",
             _gen.english_like(3000).tobytes() * 4 + b"\x00" * 20,
             # 1-bit symbols and n > 2^18: the directory key (17 symbols) is longer than the tree's, descents start at the root
             _gen.uniform_bytes(270_000, 2, 4, base=97).tobytes()]
    for text in texts:
        n = len(text)
        exp = orc.sais(text)
        qs = [b"", text[-1:], text[-2:], text[-3:], text[:1], b"\xfe\xfd", text[:40], b"A", b"AA", b"AAA", b"AAAA", b"AAAAA",
              text[-8:], text[-9:], text[-16:], text[-17:], text[-16:] + b"\x00", text[-8:] + b"\x00", text[-5:] + b"\x00\x00",
              b"\xff" * 8, b"\xff" * 16, b"\xff" * 17, b"\x00" * 8, b"\x00" * 16, b"\x00" * 17,
              text[-24:], text[-25:], text[-24:] + b"\x00", text[-20:] + b"\x00\x00", b"\xff" * 24, b"\xff" * 25, b"\x00" * 24,
              b"\x00" * 25, text[:24], text[:25], text[:23] + b"\xff", text[:16] + b"\x00" * 8, text[:16] + b"\xff" * 8]
        for it in range(400):
            a = int(rng.integers(0, n))
            q = text[a:a + int(rng.integers(1, 14 if it < 200 else (40 if it < 340 else 90)))]
            r = rng.random()
            if r < 0.25:
                q = q[:-1] + bytes([q[-1] ^ 0x55])
            elif r < 0.35:
                q = q + bytes([int(rng.integers(0, 256))])
            elif r < 0.42:
                q = q + b"\x00" * int(rng.integers(1, 4))
            elif r < 0.47:
                q = q + b"\xff"
            qs.append(q)
        off = np.zeros(len(qs) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(q) for q in qs])
        t = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(device)
        sa = torch.from_numpy(exp.view(np.int32).copy()).to(device)
        qb = torch.frombuffer(bytearray(b"".join(qs) + b"\x00"), dtype=torch.uint8).to(device)
        d_off = torch.from_numpy(off).to(device)
        ix = sdev.DeviceIndex(t, sa, engine=eng)
        s, e, f, a = [x.cpu().numpy() for x in ix.query(qb, d_off)]
        s0, e0, f0, a0 = [x.cpu().numpy() for x in sdev.query_batch(t, sa, qb, d_off, engine=eng)]
        assert np.array_equal(s, s0) and np.array_equal(e, e0) and np.array_equal(f, f0) and np.array_equal(a, a0)
        for k, q in enumerate(qs):
            assert (int(s[k]), int(e[k])) == orc.positions(text, exp, q), (q, int(s[k]), int(e[k]))
        ix.close()
    # a table with an entry >= n is refused when the index is made
    from suffix_amd._lib import SuffixHipError
    t = torch.frombuffer(bytearray(b"abcdef"), dtype=torch.uint8).to(device)
    bad = torch.tensor([5, 4, 9, 2, 1, 0], dtype=torch.int32).to(device)
    try:
        sdev.DeviceIndex(t, bad, engine=eng)
        raise RuntimeError("index accepted a table with an entry >= n")
    except SuffixHipError:
        pass


def _stack_sweep_tree(lcp):
    """Restatement of the reference's serial sweep (suffix_tree/src/lib.rs:392-505): walk the ranks left to
    right keeping the stack of open ancestors (the path to the last leaf); an lcp value below the top closes
    nodes, one above it opens a new internal node.  Returns, per closed node, (depth, lb, rb, parent_depth,
    parent_lb) and per leaf its parent's (depth, lb).  A node is identified by (depth, lb)."""
    n = len(lcp)
    nodes, leaf_parent = {}, [None] * n
    stack = [(0, 0)]                                    # (depth, lb) of open nodes; the root is never closed here
    rb_of = {}
    for r in range(n + 1):
        cur = int(lcp[r]) if r < n else 0
        lb = r - 1 if r else 0
        # the leaf r-1 hangs under the deeper of its two boundaries; decided when boundary r is seen
        while stack[-1][0] > cur:
            d, l = stack.pop()
            rb_of[(d, l)] = r - 1
            lb = l
            parent = stack[-1] if stack[-1][0] >= cur else (cur, l)
            nodes[(d, l)] = parent
        if stack[-1][0] < cur:
            stack.append((cur, lb))
        if r < n:
            pass
    # leaf parents: deeper neighbouring boundary; resolve to the enclosing node (depth, lb) by scanning
    return nodes, rb_of


def suffix_tree_topology(eng, orc, device="cpu", scale=1):
    """sfx_lcp_intervals_dev against the definition (nearest smaller values, brute force) and against the
    node set of the reference's stack sweep; sfx_doc_lookup_dev against numpy."""
    import torch

    from suffix_amd import device as sdev
    rng = np.random.default_rng(21)
    texts = [b"banana", b"mississippi", b"a" * 70, b"ab" * 40 + b"a", _gen.fibonacci_string(10), b"x",
             _gen.dna(700 * scale, seed=3).tobytes(), _gen.english_like(900 * scale).tobytes(),
             _gen.uniform_bytes(500 * scale, 3, 4, base=97).tobytes(), bytes(rng.integers(0, 256, 300 * scale, dtype=np.uint8))]
    for text in texts:
        n = len(text)
        sa = orc.sais(text)
        lcp = orc.lcp_kasai(text, sa)
        t = sdev.lcp_intervals(torch.from_numpy(lcp.view(np.int32).copy()).to(device), engine=eng)
        got = {k: v.cpu().numpy().view(np.uint32).astype(np.int64) for k, v in t.items()}
        L = lcp.astype(np.int64)
        exp_nodes = set()
        for p in range(n):
            v = L[p]
            if p == 0 or v == 0:
                assert (got["lb"][p], got["rb"][p], got["node"][p]) == (0, n - 1, 0)
                assert got["parent"][p] == 0xFFFFFFFF           # the root is nobody's child (not its own either)
                continue
            l = p - 1
            while L[l] >= v:
                l -= 1
            r = p + 1
            while r < n and L[r] >= v:
                r += 1
            node = l + 1
            while L[node] > v:
                node += 1
            assert (got["lb"][p], got["rb"][p], got["node"][p]) == (l, r - 1, node), (p, text[:20])
            vl, vr = L[l], (L[r] if r < n else 0)
            assert got["parent"][p] == (got["node"][l] if vl >= vr else got["node"][r]), p
            exp_nodes.add((int(v), l, r - 1))
            # every suffix of the interval shares exactly `depth` symbols
            a, b = int(sa[l]), int(sa[r - 1])
            assert text[a:a + v] == text[b:b + v] and (a + v == n or b + v == n or text[a + v] != text[b + v])
        for r in range(n):
            dl, dr = L[r], (L[r + 1] if r + 1 < n else 0)
            assert got["leaf_parent"][r] == (got["node"][r] if dl >= dr else got["node"][r + 1])
        assert got["parent"][0] == 0xFFFFFFFF
        # the same internal nodes as the reference's sweep builds
        nodes, rb_of = _stack_sweep_tree(lcp)
        sweep = {(d, l, rb_of[(d, l)]) for (d, l) in nodes}
        assert sweep == exp_nodes, (len(sweep), len(exp_nodes), text[:20])
        # ... and every array equal to the oracle's restatement of that sweep (oracle/sfx_oracle.c: orc_suffix_tree_sweep)
        ref = orc.suffix_tree_sweep(lcp)
        for k in ("lb", "rb", "node", "parent", "leaf_parent"):
            assert np.array_equal(got[k], ref[k].astype(np.int64)), (k, text[:20])
        # include/suffix_hip.h: "d_lcp[0] is not looked at" -- a caller whose LCP routine leaves anything there (0xFFFFFFFF, the
        # text length ...) gets the same tree: leaf 0's parent and the parents of the nodes delimited by boundary 0 included
        for junk in (0xFFFFFFFF, n, 1):
            poisoned = lcp.copy()
            poisoned[0] = junk
            t2 = sdev.lcp_intervals(torch.from_numpy(poisoned.view(np.int32).copy()).to(device), engine=eng)
            for k in ("lb", "rb", "node", "parent", "leaf_parent"):
                assert np.array_equal(t2[k].cpu().numpy().view(np.uint32), ref[k]), (k, junk, text[:20])
    # generalized suffix array: positions -> (document, offset)
    docs = [b"alpha beta", b"", b"gamma", b"delta epsilon zeta"]
    starts, blob = [], b""
    for dd in docs:
        starts.append(len(blob))
        blob += dd + b"\x00"
    st = SuffixTable(blob, engine=eng)
    pos = torch.from_numpy(st.table().view(np.int32).copy()).to(device)
    d, off = sdev.doc_lookup(pos, torch.tensor(starts, dtype=torch.int64).to(device), engine=eng)
    exp_d = np.searchsorted(np.array(starts), st.table().astype(np.int64), side="right") - 1
    assert np.array_equal(d.cpu().numpy(), exp_d) and np.array_equal(off.cpu().numpy(), st.table().astype(np.int64) - np.array(starts)[exp_d])
    hits = st.positions("eta")                            # "beta", "zeta": documents 0 and 3
    dh, _ = sdev.doc_lookup(torch.from_numpy(hits.view(np.int32).copy()).to(device),
                            torch.tensor(starts, dtype=torch.int64).to(device), engine=eng)
    assert sorted(dh.cpu().numpy().tolist()) == [0, 3]


def suffix_tree_at_scale(eng, orc, text, device="cuda"):
    """sfx_lcp_intervals_dev on a text of megabytes: all five arrays against the oracle's restatement of the
    reference's sweep (suffix_tree/src/lib.rs:392-505), which is linear."""
    import torch

    from suffix_amd import device as sdev
    t = torch.from_numpy(np.frombuffer(text, dtype=np.uint8).copy()).to(device)
    sa, lcp = sdev.build_sa_lcp(t, engine=eng)
    lcp_h = lcp.cpu().numpy().view(np.uint32)
    assert np.array_equal(lcp_h, orc.lcp_kasai(text, sa.cpu().numpy().view(np.uint32)))
    got = sdev.lcp_intervals(lcp, engine=eng)
    ref = orc.suffix_tree_sweep(lcp_h)
    for k in ("lb", "rb", "node", "parent", "leaf_parent"):
        g = got[k].cpu().numpy().view(np.uint32)
        assert np.array_equal(g, ref[k]), (k, len(text), int(np.flatnonzero(g != ref[k])[0]))
    return int((ref["node"] == np.arange(len(text), dtype=np.uint32)).sum())          # internal nodes (ids = their own boundary)


def range_slices(eng, orc, text, nranges, device="cpu", packed=False, top_bits=14):
    """The range-partitioned build driven for `nranges` virtual ranks in one process: every
    range [lo, hi) of the planned key bins is built on its own and the slices must
    concatenate to the oracle's suffix array.  Many ranges make sparse tiles in the filter
    (direct stores), few make dense ones (LDS-compacted)."""
    import ctypes

    import torch

    from suffix_amd import dist as sdist
    from suffix_amd.device import _p
    exp = orc.sais(text)
    n = len(text)
    t = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(device)
    bb = torch.zeros(256, dtype=torch.int64, device=device)
    eng.check(eng.lib.sfx_byte_histogram_dev(_p(t), 0, n, _p(bb), None), "byte_hist")
    assert np.array_equal(bb.cpu().numpy(), np.bincount(np.frombuffer(text, dtype=np.uint8), minlength=256))
    sigma = int((bb > 0).sum())
    sym_bits = max(1, (max(sigma, 2) - 1).bit_length())
    spw = 32 // sym_bits
    tb = min(top_bits, sym_bits * spw)
    kb = torch.zeros(1 << tb, dtype=torch.int64, device=device)
    # the histogram in two pieces with an odd split point: unaligned chunk starts take the byte path
    cut = (n // 3) | 1
    kb2 = torch.zeros_like(kb)
    eng.check(eng.lib.sfx_key_histogram_dev(_p(t), n, 0, cut, _p(bb), tb, _p(kb), None), "key_hist")
    eng.check(eng.lib.sfx_key_histogram_dev(_p(t), n, cut, n, _p(bb), tb, _p(kb2), None), "key_hist")
    kb += kb2
    assert int(kb.sum()) == n
    d_packed = None
    if packed:
        nw = (n + spw - 1) // spw
        d_packed = torch.zeros(nw + 4, dtype=torch.int32, device=device)
        scratch = torch.empty(256, dtype=torch.uint8, device=device)
        eng.check(eng.lib.sfx_pack_text_dev(_p(t), n, _p(bb), _p(scratch), _p(d_packed), nw, None), "pack")
    pieces = []
    for lo, hi, off, cnt in sdist.plan_ranges(kb.cpu(), nranges):
        cap = max(cnt, 1)
        part = torch.empty(cap, dtype=torch.int32, device=device)
        ws = torch.empty(int(eng.lib.sfx_sa_range_workspace_bytes(n, cap)), dtype=torch.uint8, device=device)
        got = ctypes.c_uint64(0)
        if packed:
            rc = eng.lib.sfx_build_sa_range_packed_u32_dev(_p(d_packed), n, _p(bb), tb, lo, hi, cap, _p(part),
                                                           ctypes.byref(got), _p(ws), ws.numel(), None)
        else:
            rc = eng.lib.sfx_build_sa_range_u32_dev(_p(t), n, _p(bb), tb, lo, hi, cap, _p(part),
                                                    ctypes.byref(got), _p(ws), ws.numel(), None)
        assert off == sum(p.size for p in pieces)
        if rc == 7:          # SFX_ERR_NEEDS_RANKS: repeats too long for a slice on its own -> whole-array build (dist.py's fallback)
            pieces.append(SuffixTable(text, engine=eng).table()[off:off + cnt])
            continue
        eng.check(rc, "range")
        assert int(got.value) == cnt
        pieces.append(part[:cnt].cpu().numpy().view(np.uint32))
    assert np.array_equal(np.concatenate(pieces), exp)
