"""ROUND-1 input generators, kept only so that round-1 results (profiles/r1*_fullsize_configs*.jsonl,
sha256-pinned) can be reproduced: `english_like` here is a PCG64 Zipf word stream without numeric
tokens and `utf8_mixed` draws code points uniformly -- both easier (mean LCP 13 / 7) than what
SURVEY.md 8d specifies.  tests/_gen.py holds the current generators.

Original header: Deterministic input generators shared by tests and bench.py.

splitmix64-seeded; pure numpy so the same bytes are produced on every box.
"""
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


_LETTERS = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
_LFREQ = np.array([12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8, 2.4,
                   2.4, 2.2, 2.0, 2.0, 1.9, 1.5, 1.0, 0.8, 0.15, 0.15, 0.1, 0.07])


def _ragged_gather(pool, starts, lens):
    """concatenate pool[starts[k] : starts[k]+lens[k]] for all k (vectorised)."""
    total = int(lens.sum())
    out_off = np.cumsum(lens) - lens
    idx = np.arange(total, dtype=np.int64) - np.repeat(out_off, lens) + np.repeat(starts, lens)
    return pool[idx]


def english_like(n, seed=0x5AF1C5 + 2, vocab=50000):
    """Exactly n bytes of English-like ASCII: Zipf(1.0) draws
    from `vocab` pseudo-words of length 1-12 built from English letter frequencies,
    joined by ' ' / ', ' (p=.08) / '. ' (p=.06, next word capitalised); the separator's
    space becomes '\n' roughly every 80 characters.  Fully vectorised (1 GB in ~1 min)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    wlen = rng.integers(1, 13, size=vocab).astype(np.int64)
    cdf = np.cumsum(_LFREQ) / _LFREQ.sum()
    letters = _LETTERS[np.searchsorted(cdf, rng.random(int(wlen.sum())))]
    wstart = np.cumsum(wlen) - wlen
    # token pool: word (lower / capitalised) followed by one of 6 separators
    seps = [b" ", b", ", b". ", b"\n", b",\n", b".\n"]
    pool_parts, p_start, p_len = [], np.zeros((vocab, 2, 6), np.int64), np.zeros((vocab, 2, 6), np.int64)
    lower = letters
    upper = letters.copy()
    upper[wstart] -= 32                                   # capitalise first letters
    off = 0
    for cap, src in enumerate((lower, upper)):
        for si, sp in enumerate(seps):
            lens = wlen + len(sp)
            st = off + np.cumsum(lens) - lens
            buf = np.empty(int(lens.sum()), dtype=np.uint8)
            idx_word = _ragged_gather(np.arange(src.size, dtype=np.int64), wstart, wlen)
            dst = np.arange(buf.size, dtype=np.int64)
            is_sep = np.ones(buf.size, dtype=bool)
            word_dst = _ragged_gather(dst, st - off, wlen)
            buf[word_dst] = src[idx_word]
            is_sep[word_dst] = False
            buf[is_sep] = np.tile(np.frombuffer(sp, dtype=np.uint8), vocab)
            pool_parts.append(buf)
            p_start[:, cap, si] = st
            p_len[:, cap, si] = lens
            off += buf.size
    pool = np.concatenate(pool_parts)
    zipf = 1.0 / np.arange(1, vocab + 1)
    zcdf = np.cumsum(zipf) / zipf.sum()
    out = np.empty(n, dtype=np.uint8)
    filled, col_base, prev_period = 0, 0, True
    chunk = 1 << 22
    while filled < n:
        ws = np.searchsorted(zcdf, rng.random(chunk))
        ps = rng.random(chunk)
        kind = np.where(ps < 0.06, 2, np.where(ps < 0.14, 1, 0))
        cap = np.empty(chunk, dtype=np.int64)
        cap[0] = 1 if prev_period else 0
        cap[1:] = (kind[:-1] == 2)
        base_len = wlen[ws] + np.where(kind == 0, 1, 2)
        cum = np.cumsum(base_len) + col_base
        nl = (cum // 80) != ((cum - base_len) // 80)        # crossed a multiple of 80 columns
        si = kind + 3 * nl
        starts = p_start[ws, cap, si]
        lens = p_len[ws, cap, si]
        piece = _ragged_gather(pool, starts, lens)
        take = min(piece.size, n - filled)
        out[filled:filled + take] = piece[:take]
        filled += take
        col_base = int(cum[-1] % 80)
        prev_period = bool(kind[-1] == 2)
    return out


def utf8_mixed(n, seed=0x5AF1C5 + 5):
    """<= n bytes of valid UTF-8 mixing 1/2/3/4-byte code points (SURVEY.md 8d, config
    5): words of 1-8 code points, script per word 40% ASCII, 20% Cyrillic (2 B), 30% CJK
    (3 B), 10% U+1F300.. (4 B), separated by spaces; truncated at a code-point boundary."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lo = np.array([0x61, 0x0410, 0x4E00, 0x1F300], dtype=np.int64)
    span = np.array([26, 64, 2000, 0x300], dtype=np.int64)
    parts, size = [], 0
    while size < n:
        W = max(1024, min(1 << 21, (n - size) // 6 + 1024))
        sc = np.searchsorted(np.array([0.40, 0.60, 0.90]), rng.random(W), side="right")
        wl = rng.integers(1, 9, size=W)
        tot = int(wl.sum())
        sc_c = np.repeat(sc, wl)
        cp = lo[sc_c] + (rng.random(tot) * span[sc_c]).astype(np.int64)
        # interleave a space after every word
        ends = np.cumsum(wl)
        cps = np.empty(tot + W, dtype=np.int64)
        pos = np.arange(tot) + np.repeat(np.arange(W), wl)
        cps[pos] = cp
        cps[ends + np.arange(W)] = 0x20
        nb = np.where(cps < 0x80, 1, np.where(cps < 0x800, 2, np.where(cps < 0x10000, 3, 4)))
        off = np.cumsum(nb) - nb
        buf = np.empty(int(nb.sum()), dtype=np.uint8)
        m1, m2, m3, m4 = nb == 1, nb == 2, nb == 3, nb == 4
        buf[off[m1]] = cps[m1]
        buf[off[m2]] = 0xC0 | (cps[m2] >> 6)
        buf[off[m2] + 1] = 0x80 | (cps[m2] & 0x3F)
        buf[off[m3]] = 0xE0 | (cps[m3] >> 12)
        buf[off[m3] + 1] = 0x80 | ((cps[m3] >> 6) & 0x3F)
        buf[off[m3] + 2] = 0x80 | (cps[m3] & 0x3F)
        buf[off[m4]] = 0xF0 | (cps[m4] >> 18)
        buf[off[m4] + 1] = 0x80 | ((cps[m4] >> 12) & 0x3F)
        buf[off[m4] + 2] = 0x80 | ((cps[m4] >> 6) & 0x3F)
        buf[off[m4] + 3] = 0x80 | (cps[m4] & 0x3F)
        parts.append(buf)
        size += buf.size
    b = np.concatenate(parts)[:n]
    end = b.size
    while end > 0 and (b[end - 1] & 0xC0) == 0x80:          # strip trailing continuation bytes
        end -= 1
    if end > 0 and b[end - 1] >= 0xC0:                       # and a dangling lead byte
        end -= 1
    elif end < b.size:                                       # continuation bytes followed a complete char?
        lead = b[end - 1] if end else 0
        need = 0 if lead < 0x80 else (2 if lead < 0xE0 else 3 if lead < 0xF0 else 4)
        have = b.size - end + 1
        end = b.size if have == need else end - 1
    return np.ascontiguousarray(b[:end])


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
