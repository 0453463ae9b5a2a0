"""CPU tests: pin the oracle (the C restatement of /root/reference/src/table.rs)
against the reference's own known answers, its `naive_table` oracle, the
definition, and the golden fixture hashes."""
import hashlib
import random

import numpy as np
import pytest

import _gen


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u4").tobytes()).hexdigest()


def test_literals_sa_and_lcp(oracle, golden):
    # tests/tests.rs:22-70 (basic1..snowman_is_ok) + parts() text
    for s, exp in golden["sa_literals"].items():
        sa = oracle.sais(s)
        assert sa.tolist() == exp["sa"], s
        assert oracle.naive_sa(s).tolist() == exp["sa"], s
        assert oracle.lcp_quadratic(s, sa).tolist() == exp["lcp"], s
        assert oracle.lcp_kasai(s, sa).tolist() == exp["lcp"], s


def test_survey_literal_values(oracle):
    # SURVEY.md 8c small literals
    assert oracle.sais("banana").tolist() == [5, 3, 1, 0, 4, 2]
    assert oracle.sais("mississippi").tolist() == [10, 7, 4, 1, 0, 9, 8, 6, 3, 5, 2]
    assert oracle.sais("☃abc☃").tolist() == [3, 4, 5, 8, 2, 7, 1, 6, 0]
    assert oracle.lcp_quadratic("mississippi", oracle.sais("mississippi")).tolist() == \
        [0, 1, 1, 4, 0, 0, 1, 0, 2, 1, 3]


def test_search_known_answers(oracle, golden):
    # tests/tests.rs:100-168, :181-213 and the doc-tests
    for text, query, pos, found in golden["search"]:
        sa = oracle.sais(text)
        s, e = oracle.positions(text, sa, query)
        assert sa[s:e].tolist() == pos, (text, query)
        anyp = oracle.any_position(text, sa, query)
        assert (anyp is not None) == found, (text, query)
        if found:
            assert anyp in pos


@pytest.mark.parametrize("name", ["AP009048_10000", "AP009048_100000"])
def test_fasta_fixture_hashes(oracle, golden, fasta, name):
    g = golden["fixtures"][name]
    text = fasta[name]
    assert len(text) == g["len"]
    assert hashlib.sha256(text).hexdigest() == g["sha256_text"]
    sa = oracle.sais(text)
    assert _sha(sa) == g["sha256_sa"]
    assert sa[:6].tolist() == g["sa_head"] and sa[-4:].tolist() == g["sa_tail"]
    lcp = oracle.lcp_quadratic(text, sa)
    assert _sha(lcp) == g["sha256_lcp"]
    assert int(lcp.max()) == g["max_lcp"]
    assert _sha(oracle.lcp_kasai(text, sa)) == g["sha256_lcp"]
    # tests/bench.rs queries (:65-133); counts from SURVEY.md 8c
    s, e = oracle.positions(text, sa, "ACTTACGTGTCTGC")
    assert sa[s:e].tolist() == [1825]
    assert oracle.positions(text, sa, "H") == (0, 0)
    s, e = oracle.positions(text, sa, "C")
    assert e - s == (2511 if name == "AP009048_10000" else 25342)
    assert e - s == text.count(b"C")


def test_prop_sais_equals_naive_random_bytes(oracle):
    # mirrors prop_naive_equals_sais / prop_matches_naive (tests/tests.rs:73-96)
    rnd = random.Random(20260925)
    for _ in range(3000):
        n = rnd.randint(0, 80)
        sigma = rnd.choice([1, 2, 3, 4, 16, 256])
        t = bytes(rnd.randrange(sigma) for _ in range(n))
        sa = oracle.sais(t)
        assert sa.tolist() == oracle.naive_sa(t).tolist(), t
        assert len(sa) == n                                   # prop_length :215-221


def test_prop_unicode_strings(oracle):
    # String: Arbitrary generates arbitrary Unicode -> multi-byte text
    rnd = random.Random(7)
    pools = [range(0x20, 0x7F), range(0xA0, 0x250), range(0x4E00, 0x4E40),
             range(0x1F300, 0x1F320), [0, 0x2603]]
    for _ in range(1000):
        n = rnd.randint(0, 30)
        s = "".join(chr(rnd.choice(list(rnd.choice(pools)))) for _ in range(n))
        b = s.encode("utf-8")
        sa = oracle.sais(b)
        assert sa.tolist() == oracle.definitional_sa(b).tolist(), s
        # prop_contains / prop_positions (tests/tests.rs:223-243)
        c = chr(rnd.randrange(128))
        st, en = oracle.positions(b, sa, c)
        got = sorted(sa[st:en].tolist())
        cb = c.encode()
        exp = [i for i in range(len(b)) if b[i:i + 1] == cb]
        assert got == exp
        assert (oracle.any_position(b, sa, c) is not None) == (cb in b)


def test_structured_strings(oracle):
    cases = [b"a" * 1000, b"ab" * 500, b"abc" * 333 + b"a", _gen.fibonacci_string(14),
             _gen.thue_morse(3000).tobytes(), bytes(range(256)) * 4, bytes(reversed(range(256))),
             b"\x00" * 100 + b"\xff" * 100, b"\xff" * 100 + b"\x00" * 100]
    for t in cases:
        sa = oracle.sais(t)
        assert (sa == oracle.naive_sa(t)).all()
        assert (oracle.lcp_quadratic(t, sa) == oracle.lcp_kasai(t, sa)).all()


def test_medium_inputs_vs_naive(oracle):
    for t in (_gen.dna(200_000).tobytes(), _gen.english_like(150_000).tobytes(),
              _gen.utf8_mixed(100_000).tobytes(), _gen.uniform_bytes(100_000, 96, 3, 32).tobytes()):
        sa = oracle.sais(t)
        assert (sa == oracle.naive_sa(t)).all()
        assert (oracle.lcp_quadratic(t, sa) == oracle.lcp_kasai(t, sa)).all()


def test_generators_are_deterministic():
    assert hashlib.sha256(_gen.dna(100_000).tobytes()).hexdigest() == \
        hashlib.sha256(_gen.dna(100_000).tobytes()).hexdigest()
    d = _gen.dna(64)
    assert set(d.tobytes()) <= set(b"ACGT")
    e = _gen.english_like(5000)
    assert e.size == 5000 and e.max() < 127
    u = _gen.utf8_mixed(5000)
    u.tobytes().decode("utf-8")


def test_suffix_tree_sweep_restatement(oracle):
    """orc_suffix_tree_sweep (suffix_tree/src/lib.rs:392-505 as flat arrays) against the definition: the node of a
    boundary is the lcp-interval around it (nearest smaller values by brute force), ids are leftmost boundaries, a
    leaf hangs under the deeper of its two boundaries; and against the Python restatement of the sweep in _cases."""
    import numpy as np
    import _cases
    import _gen
    rng = np.random.default_rng(5)
    texts = [b"banana", b"mississippi", b"a" * 50, b"ab" * 30 + b"a", _gen.fibonacci_string(11), b"x", b"xy",
             _gen.dna(3000, seed=2).tobytes(), _gen.english_like(3000).tobytes(), bytes(rng.integers(0, 4, 2000, dtype=np.uint8))]
    for text in texts:
        n = len(text)
        sa = oracle.sais(text)
        L = oracle.lcp_kasai(text, sa).astype(np.int64)
        r = {k: v.astype(np.int64) for k, v in oracle.suffix_tree_sweep(L.astype(np.uint32)).items()}
        for p in range(n):
            v = L[p] if p else 0
            if v == 0:
                assert (r["lb"][p], r["rb"][p], r["node"][p], r["parent"][p]) == (0, n - 1, 0, 0xFFFFFFFF)
                continue
            l = p - 1
            while l > 0 and L[l] >= v:
                l -= 1
            q = p + 1
            while q < n and L[q] >= v:
                q += 1
            node = l + 1
            while L[node] > v:
                node += 1
            assert (r["lb"][p], r["rb"][p], r["node"][p]) == (l, q - 1, node), (p, text[:16])
            vl, vr = (L[l] if l else 0), (L[q] if q < n else 0)
            assert r["parent"][p] == (r["node"][l] if vl >= vr else r["node"][q]), (p, text[:16])
        for k in range(n):
            dl, dr = (L[k] if k else 0), (L[k + 1] if k + 1 < n else 0)
            assert r["leaf_parent"][k] == (r["node"][k] if dl >= dr else r["node"][k + 1])
        nodes, rb_of = _cases._stack_sweep_tree(L.astype(np.uint32))
        mine = {(int(L[p]), int(r["lb"][p])) for p in range(1, n) if L[p] > 0}
        assert mine == set(nodes), text[:16]
