"""Deterministic input generators shared by tests and bench.py.

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


def english_like(n, seed=0x5AF1C5 + 2, vocab=50000):
    """~n bytes of English-like ASCII: Zipf(1.0) draws from `vocab` pseudo-words of
    length 1-12 built from English letter frequencies, joined by ' ', ', ' (p=.08)
    or '. ' (p=.06, next word capitalised), '\\n' instead of the space roughly every
    80 chars.  Exactly n bytes are returned."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lens = rng.integers(1, 13, size=vocab)
    cdf = np.cumsum(_LFREQ) / _LFREQ.sum()
    pool = _LETTERS[np.searchsorted(cdf, rng.random(int(lens.sum())))]
    offs = np.concatenate(([0], np.cumsum(lens)))
    words = [pool[offs[i]:offs[i + 1]].tobytes() for i in range(vocab)]
    zipf = 1.0 / np.arange(1, vocab + 1)
    zcdf = np.cumsum(zipf) / zipf.sum()
    out = bytearray()
    col = 0
    cap = True
    chunk = 1 << 16
    while len(out) < n:
        ws = np.searchsorted(zcdf, rng.random(chunk))
        ps = rng.random(chunk)
        for w, p in zip(ws.tolist(), ps.tolist()):
            tok = words[w]
            if cap:
                tok = tok[:1].upper() + tok[1:]
                cap = False
            if p < 0.06:
                sep = b". "
                cap = True
            elif p < 0.14:
                sep = b", "
            else:
                sep = b" "
            col += len(tok) + len(sep)
            if col >= 80:
                sep = sep[:-1] + b"\n"
                col = 0
            out += tok
            out += sep
            if len(out) >= n:
                break
    return np.frombuffer(bytes(out[:n]), dtype=np.uint8)


def utf8_mixed(n, seed=0x5AF1C5 + 5):
    """<= n bytes of valid UTF-8 mixing 1/2/3/4-byte code points (config 5),
    truncated at a code-point boundary."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    size = 0
    while size < n:
        script = rng.random()
        wl = int(rng.integers(1, 9))
        if script < 0.40:
            cps = rng.integers(0x61, 0x7B, size=wl)
        elif script < 0.60:
            cps = rng.integers(0x0410, 0x0450, size=wl)
        elif script < 0.90:
            cps = rng.integers(0x4E00, 0x4E00 + 2000, size=wl)
        else:
            cps = rng.integers(0x1F300, 0x1F600, size=wl)
        w = ("".join(map(chr, cps.tolist())) + " ").encode("utf-8")
        out.append(w)
        size += len(w)
    b = b"".join(out)[:n]
    while True:                      # trim to a code-point boundary
        try:
            b.decode("utf-8")
            break
        except UnicodeDecodeError:
            b = b[:-1]
    return np.frombuffer(b, dtype=np.uint8)


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
