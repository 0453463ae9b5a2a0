"""ctypes front-end of oracle/libsfx_oracle.so (the C restatement of
/root/reference/src/table.rs) plus a tiny pure-numpy definitional oracle.

TEST INFRASTRUCTURE ONLY.  Parity pin: see the header of sfx_oracle.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsfx_oracle.so")
_lib = None

__all__ = ["build", "sais", "naive_sa", "lcp_quadratic", "lcp_kasai", "positions",
           "positions_batch", "any_position", "definitional_sa", "suffix_tree_sweep"]


def build(force=False):
    """Compile the oracle with gcc (seconds)."""
    src = [os.path.join(_HERE, f) for f in ("sfx_oracle.c", "sais_level.inc")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src)):
        return _SO
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libsfx_oracle.so"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        u8p, u32p, u64 = ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64
        L.orc_sais.argtypes = [u8p, u64, u32p]
        L.orc_sais.restype = ctypes.c_int
        L.orc_naive_sa.argtypes = [u8p, u64, u32p]
        L.orc_naive_sa.restype = ctypes.c_int
        L.orc_lcp_quadratic.argtypes = [u8p, u64, u32p, u32p]
        L.orc_lcp_quadratic.restype = None
        L.orc_lcp_kasai.argtypes = [u8p, u64, u32p, u32p, u32p]
        L.orc_lcp_kasai.restype = None
        L.orc_positions.argtypes = [u8p, u64, u32p, u8p, u64,
                                    ctypes.POINTER(u64), ctypes.POINTER(u64)]
        L.orc_positions.restype = None
        L.orc_positions_batch.argtypes = [u8p, u64, u32p, u8p, ctypes.c_void_p, u64, u32p, u32p]
        L.orc_positions_batch.restype = None
        L.orc_any_position.argtypes = [u8p, u64, u32p, u8p, u64,
                                       ctypes.POINTER(ctypes.c_uint32)]
        L.orc_any_position.restype = ctypes.c_int
        L.orc_suffix_tree_sweep.argtypes = [u32p, u64, u32p, u32p, u32p, u32p, u32p]
        L.orc_suffix_tree_sweep.restype = None
        _lib = L
    return _lib


def _bytes_arr(text):
    if isinstance(text, str):
        text = text.encode("utf-8")
    if isinstance(text, (bytes, bytearray, memoryview)):
        return np.frombuffer(bytes(text), dtype=np.uint8)
    return np.ascontiguousarray(text, dtype=np.uint8)


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


def sais(text):
    """Reference algorithm (SA-IS restatement).  -> np.uint32[n]"""
    t = _bytes_arr(text)
    sa = np.zeros(t.size, dtype=np.uint32)
    if _load().orc_sais(_p(t), t.size, _p(sa)) != 0:
        raise OverflowError("text longer than u32::MAX (src/table.rs:380)")
    return sa


def naive_sa(text):
    """Reference `naive_table` restatement (qsort of suffixes)."""
    t = _bytes_arr(text)
    sa = np.zeros(t.size, dtype=np.uint32)
    if _load().orc_naive_sa(_p(t), t.size, _p(sa)) != 0:
        raise OverflowError("text longer than u32::MAX")
    return sa


def definitional_sa(text):
    """Pure-Python definition: sort byte suffixes (tiny inputs only)."""
    b = bytes(_bytes_arr(text))
    return np.array(sorted(range(len(b)), key=lambda i: b[i:]), dtype=np.uint32)


def lcp_quadratic(text, sa):
    t = _bytes_arr(text)
    sa = np.ascontiguousarray(sa, dtype=np.uint32)
    lcp = np.zeros(t.size, dtype=np.uint32)
    _load().orc_lcp_quadratic(_p(t), t.size, _p(sa), _p(lcp))
    return lcp


def lcp_kasai(text, sa):
    t = _bytes_arr(text)
    sa = np.ascontiguousarray(sa, dtype=np.uint32)
    lcp = np.zeros(t.size, dtype=np.uint32)
    inv = np.zeros(t.size, dtype=np.uint32)
    _load().orc_lcp_kasai(_p(t), t.size, _p(sa), _p(inv), _p(lcp))
    return lcp


def positions(text, sa, query):
    """-> (start, end): half-open SA interval, (0, 0) when empty."""
    t = _bytes_arr(text)
    q = _bytes_arr(query)
    sa = np.ascontiguousarray(sa, dtype=np.uint32)
    s, e = ctypes.c_uint64(0), ctypes.c_uint64(0)
    _load().orc_positions(_p(t), t.size, _p(sa), _p(q), q.size,
                          ctypes.byref(s), ctypes.byref(e))
    return int(s.value), int(e.value)


def positions_batch(text, sa, qbytes, qoff):
    """positions() for every query k = qbytes[qoff[k]:qoff[k+1]] -> (start, end) uint32 arrays
    (OpenMP over the host cores)."""
    t = _bytes_arr(text)
    sa = np.ascontiguousarray(sa, dtype=np.uint32)
    qb = np.ascontiguousarray(qbytes, dtype=np.uint8)
    off = np.ascontiguousarray(qoff, dtype=np.uint64)
    nq = off.size - 1
    s = np.zeros(nq, dtype=np.uint32)
    e = np.zeros(nq, dtype=np.uint32)
    _load().orc_positions_batch(_p(t), t.size, _p(sa), _p(qb), _p(off), nq, _p(s), _p(e))
    return s, e


def any_position(text, sa, query):
    """-> position or None."""
    t = _bytes_arr(text)
    q = _bytes_arr(query)
    sa = np.ascontiguousarray(sa, dtype=np.uint32)
    pos = ctypes.c_uint32(0)
    ok = _load().orc_any_position(_p(t), t.size, _p(sa), _p(q), q.size, ctypes.byref(pos))
    return int(pos.value) if ok else None


def suffix_tree_sweep(lcp):
    """to_suffix_tree (suffix_tree/src/lib.rs:392-505) over an LCP array, as flat arrays: dict(lb, rb, node, parent,
    leaf_parent), n u32 each (see orc_suffix_tree_sweep)."""
    lcp = np.ascontiguousarray(lcp, dtype=np.uint32)
    n = lcp.size
    out = {k: np.zeros(n, dtype=np.uint32) for k in ("lb", "rb", "node", "parent", "leaf_parent")}
    if n:
        _load().orc_suffix_tree_sweep(lcp.ctypes.data, n, *(out[k].ctypes.data for k in ("lb", "rb", "node", "parent", "leaf_parent")))
    return out
