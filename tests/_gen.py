"""Deterministic input generators shared by tests and bench.py.

splitmix64-seeded.  DNA and the small structured strings are numpy; the word-model texts (configs 3
and 5, the near-duplicate documents) come from tests/gen/sfxgen.c -- integer arithmetic only, so the
same bytes are produced on every box.
"""
import ctypes
import os
import subprocess

import numpy as np

MASK = (1 << 64) - 1


def splitmix64_stream(seed, count):
    """count 64-bit draws of splitmix64 starting from `seed` (vectorised)."""
    idx = np.arange(1, count + 1, dtype=np.uint64)
    z = (np.uint64(seed & MASK) + idx * np.uint64(0x9E3779B97F4A7C15))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def dna(n, seed=0x5AF1C5 + 1):
    """n bytes uniform over ACGT: each 64-bit draw yields 32 symbols, 2 bits each,
    LSB first, 0->A 1->C 2->G 3->T."""
    words = splitmix64_stream(seed, (n + 31) // 32)
    shifts = (np.arange(32, dtype=np.uint64) * np.uint64(2))
    codes = ((words[:, None] >> shifts[None, :]) & np.uint64(3)).astype(np.uint8).reshape(-1)[:n]
    return np.frombuffer(b"ACGT", dtype=np.uint8)[codes]


def uniform_bytes(n, sigma, seed, base=0):
    """n bytes uniform over [base, base+sigma)."""
    words = splitmix64_stream(seed, (n + 7) // 8)
    b = words.view(np.uint8)[:n]
    if sigma == 256:
        return (b.astype(np.uint16) + base).astype(np.uint8)
    return ((b.astype(np.uint32) * sigma) >> 8).astype(np.uint8) + np.uint8(base)


# ---- word-model texts: C generator (tests/gen/sfxgen.c), bound with ctypes ----------------------
_GEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gen")
_lib = None


def _gen_lib():
    """tests/gen/libsfxgen.so, built on first use (gcc; integer-only arithmetic: same bytes everywhere)."""
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-s", "-C", _GEN_DIR])
        _lib = ctypes.CDLL(os.path.join(_GEN_DIR, "libsfxgen.so"))
        vp, u64, u32, i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
        for name, args in (("sfxgen_english", [vp, u64, u64, i32]), ("sfxgen_utf8", [vp, u64, u64, i32]),
                           ("sfxgen_dna", [vp, u64, u64, i32]), ("sfxgen_dna_slice", [vp, u64, u64, u64, i32]),
                           ("sfxgen_near_duplicates", [vp, u64, u64, u32, u32, i32]),
                           ("sfxgen_queries", [vp, u64, u64, u64, vp, vp])):
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = i32, args
    return _lib


def _threads():
    try:
        return max(1, min(32, len(os.sched_getaffinity(0))))
    except AttributeError:
        return 4


def dna_fast(n, seed=0x5AF1C5 + 1):
    """dna() through the C generator (identical bytes, ~100x faster at 1 GB)."""
    out = np.empty(n, dtype=np.uint8)
    assert _gen_lib().sfxgen_dna(out.ctypes.data, n, seed, _threads()) == 0
    return out


def dna_slice(begin, n, seed=0x5AF1C5 + 1):
    """bytes [begin, begin + n) of dna(begin + n, seed) without making the rest: a rank's shard of ONE text."""
    out = np.empty(n, dtype=np.uint8)
    assert _gen_lib().sfxgen_dna_slice(out.ctypes.data, begin, n, seed, _threads()) == 0
    return out


def english_like(n, seed=0x5AF1C5 + 2):
    """Exactly n bytes of English-like ASCII: splitmix64; Zipf(1.0) draws
    from 50 000 pseudo-words of 1-12 letters (English letter frequencies), joined by ' ' / ', '
    (8 %) / '. ' (6 %, next word capitalised), newline about every 80 characters, 2 % numeric
    tokens.  Made in independent 1 MiB blocks (tests/gen/sfxgen.c)."""
    out = np.empty(n, dtype=np.uint8)
    assert _gen_lib().sfxgen_english(out.ctypes.data, n, seed, _threads()) == 0
    return out


def utf8_mixed(n, seed=0x5AF1C5 + 5):
    """Exactly n bytes of valid UTF-8: the same word model, each
    vocabulary word in one script -- 40 % ASCII, 20 % Cyrillic / Greek (2 B), 30 % CJK U+4E00..
    (3 B), 10 % U+1F300.. (4 B); a word that would be cut by a block end is replaced by spaces."""
    out = np.empty(n, dtype=np.uint8)
    assert _gen_lib().sfxgen_utf8(out.ctypes.data, n, seed, _threads()) == 0
    return out


def near_duplicates(n, seed=0x5AF1C5 + 6, ndocs=16, every=400):
    """High-LCP input: `ndocs` distinct 1 MiB English-like documents repeated round-robin, every
    later copy with a substituted byte about every `every` bytes (near-duplicate documents)."""
    out = np.empty(n, dtype=np.uint8)
    assert _gen_lib().sfxgen_near_duplicates(out.ctypes.data, n, seed, ndocs, every, _threads()) == 0
    return out


def queries(text, nq, seed=0x5AF1C5 + 5):
    """SURVEY.md 8d config-5 queries over a valid UTF-8 `text` (uint8 array): nq substrings of 1-16
    code points starting at code-point boundaries; the second half has its last code point replaced
    (mostly misses).  -> (qbytes uint8 array, qoff int64 array of nq + 1 offsets)."""
    text = np.ascontiguousarray(text, dtype=np.uint8)
    qb = np.empty(64 * nq + 8, dtype=np.uint8)
    off = np.zeros(nq + 1, dtype=np.uint64)
    assert _gen_lib().sfxgen_queries(text.ctypes.data, text.size, nq, seed, qb.ctypes.data, off.ctypes.data) == 0
    return np.ascontiguousarray(qb[:int(off[-1])]), off.astype(np.int64)


def fibonacci_string(k):
    a, b = b"a", b"ab"
    for _ in range(k):
        a, b = b, b + a
    return b


def thue_morse(n):
    i = np.arange(n, dtype=np.uint64)
    bits = np.zeros(n, dtype=np.uint8)
    x = i.copy()
    while x.any():
        bits ^= (x & np.uint64(1)).astype(np.uint8)
        x >>= np.uint64(1)
    return bits + np.uint8(ord("a"))
