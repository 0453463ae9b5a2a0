"""`SuffixTable` -- host-side mirror of the reference's public type
(/root/reference/src/table.rs:54-294) on top of the MI355X engine.

Same method names, argument meaning and error behaviour as the Rust API:

    reference (Rust)                         here
    SuffixTable::new(text)            :78    SuffixTable.new(text) / SuffixTable(text)
    SuffixTable::new_naive(text)      :93    SuffixTable.new_naive(text)   (doc-hidden upstream: the definition, on the host)
    SuffixTable::from_parts(t, sa)    :111   SuffixTable.from_parts(text, table)
    .into_parts()                     :125   .into_parts()
    .lcp_lens()                       :130   .lcp_lens()
    .table() / .text()                :142   .table() / .text()
    .len() / .is_empty()              :156   .len() / .is_empty()
    .suffix(i) / .suffix_bytes(i)     :168   .suffix(i) / .suffix_bytes(i)
    .contains(q)                      :197   .contains(q)
    .positions(q)                     :223   .positions(q)
    .any_position(q)                  :279   .any_position(q)
    (none)                                   .positions_batch(qs) / .contains_batch(qs)

Text is indexed by BYTES (every UTF-8 byte offset has a suffix, :29-31 of the
crate docs and :379); `str` input is encoded as UTF-8.  Construction, LCP and
all queries run on the GPU through the C ABI; there is no CPU path.
"""
import ctypes

import numpy as np

from ._lib import default_engine

_NONE = 0xFFFFFFFF


def _as_bytes(x):
    if isinstance(x, str):
        return x.encode("utf-8")
    if isinstance(x, (bytes, bytearray, memoryview)):
        return bytes(x)
    if isinstance(x, np.ndarray):
        return np.ascontiguousarray(x, dtype=np.uint8).tobytes()
    raise TypeError("text/query must be str, bytes or a uint8 array")


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data if a.size else 0)


class SuffixTable:
    def __init__(self, text, _table=None, engine=None):
        self._eng = engine or default_engine()
        self._was_str = isinstance(text, str)
        self._text = _as_bytes(text)
        self._tarr = np.frombuffer(self._text, dtype=np.uint8)
        self._index = None
        if _table is None:
            # sais_table (:378-386): assert len <= u32::MAX, allocate, fill
            n = self._tarr.size
            table = np.zeros(n, dtype=np.uint32)
            if n:
                self._eng.require_device()
            self._eng.check(self._eng.lib.sfx_build_sa_u32(_ptr(self._tarr), n, _ptr(table)),
                            "SuffixTable::new")
            self._table = table
        else:
            self._table = _table

    # -- constructors -----------------------------------------------------------------
    @classmethod
    def new(cls, text, engine=None):
        return cls(text, engine=engine)

    @classmethod
    def new_naive(cls, text, engine=None):
        """SuffixTable::new_naive (:93-100, #[doc(hidden)]) -> naive_table (:367-376): the definition -- every byte suffix
        sorted by comparison on the host, O(n^2 log n).  Upstream keeps it as the known answer of its own tests
        (tests/tests.rs:18-20) and so does this mirror: it is what a caller compares new() against, never a fallback of
        new() (which has none: it raises without the HIP library or a device)."""
        t = _as_bytes(text)
        if len(t) > 0xFFFFFFFF:
            raise OverflowError("SuffixTable::new_naive: text longer than u32::MAX")      # the table is Vec<u32> (:57)
        table = np.array(sorted(range(len(t)), key=lambda i: t[i:]), dtype=np.uint32)
        return cls(text, _table=table, engine=engine)

    @classmethod
    def new_with_lcp(cls, text, engine=None):
        """SuffixTable::new + lcp_lens in one engine call (sfx_build_sa_lcp_u32): -> (table, lcp array).
        The pair of calls suffix_tree/src/lib.rs:71 + :413 makes; same arrays as new() then lcp_lens()."""
        eng = engine or default_engine()
        tarr = np.frombuffer(_as_bytes(text), dtype=np.uint8)
        n = tarr.size
        table = np.zeros(n, dtype=np.uint32)
        lcp = np.zeros(n, dtype=np.uint32)
        if n:
            eng.require_device()
        eng.check(eng.lib.sfx_build_sa_lcp_u32(_ptr(tarr), n, _ptr(table), _ptr(lcp)), "SuffixTable::new + lcp_lens")
        return cls(text, _table=table, engine=engine), lcp

    @classmethod
    def from_parts(cls, text, table, engine=None):
        """Unchecked, like the reference (:105-119): only the lengths must agree."""
        t = np.ascontiguousarray(table, dtype=np.uint32)
        if len(_as_bytes(text)) != t.size:
            raise AssertionError("text.len() != table.len()")        # assert_eq! :117
        return cls(text, _table=t, engine=engine)

    def into_parts(self):
        text = self._text.decode("utf-8") if self._was_str else self._text
        return text, self._table

    def __del__(self):
        ix, self._index = getattr(self, "_index", None), None
        if ix:
            try:
                self._eng.lib.sfx_index_destroy(ix)
            except Exception:
                pass

    # -- accessors ----------------------------------------------------------------------
    def table(self):
        return self._table

    def text(self):
        return self._text.decode("utf-8") if self._was_str else self._text

    def len(self):
        return int(self._table.size)

    __len__ = len

    def is_empty(self):
        return self.len() == 0

    def suffix_bytes(self, i):
        return self._text[int(self._table[i]):]

    def suffix(self, i):
        # the reference slices a &str and panics off a char boundary (:168-170)
        return self.suffix_bytes(i).decode("utf-8")

    def __eq__(self, other):                      # derive(PartialEq) on (text, table), :54
        return (isinstance(other, SuffixTable) and self._text == other._text
                and np.array_equal(self._table, other._table))

    # -- LCP ------------------------------------------------------------------------------
    def lcp_lens(self):
        n = self.len()
        lcp = np.zeros(n, dtype=np.uint32)
        if n:
            self._eng.require_device()
        self._eng.check(self._eng.lib.sfx_build_lcp_u32(_ptr(self._tarr), n, _ptr(self._table),
                                                         _ptr(lcp)), "lcp_lens")
        return lcp

    # -- queries ----------------------------------------------------------------------------
    def _ensure_index(self):
        if self._index is None:
            self._eng.require_device()
            h = ctypes.c_void_p()
            self._eng.check(self._eng.lib.sfx_index_create(_ptr(self._tarr), self.len(),
                                                           _ptr(self._table), ctypes.byref(h)),
                            "sfx_index_create")
            self._index = h
        return self._index

    @staticmethod
    def _pack(queries):
        qs = [_as_bytes(q) for q in queries]
        off = np.zeros(len(qs) + 1, dtype=np.uint64)
        if qs:
            off[1:] = np.cumsum([len(q) for q in qs], dtype=np.uint64)
        blob = np.frombuffer(b"".join(qs), dtype=np.uint8)
        return blob, off

    def positions_batch(self, queries):
        """-> (start, end) uint32 arrays; positions(q_k) == table()[start[k]:end[k]]."""
        blob, off = self._pack(queries)
        nq = off.size - 1
        start = np.zeros(nq, dtype=np.uint32)
        end = np.zeros(nq, dtype=np.uint32)
        if nq:
            self._eng.check(self._eng.lib.sfx_positions_batch(self._ensure_index(), _ptr(blob),
                                                               _ptr(off), nq, _ptr(start), _ptr(end)),
                            "positions_batch")
        return start, end

    def contains_batch(self, queries):
        """-> (found bool array, any_position uint32 array with 0xFFFFFFFF = None)."""
        blob, off = self._pack(queries)
        nq = off.size - 1
        found = np.zeros(nq, dtype=np.uint8)
        anyp = np.full(nq, _NONE, dtype=np.uint32)
        if nq:
            self._eng.check(self._eng.lib.sfx_contains_batch(self._ensure_index(), _ptr(blob),
                                                              _ptr(off), nq, _ptr(found), _ptr(anyp)),
                            "contains_batch")
        return found.astype(bool), anyp

    def positions(self, query):
        """Unordered (SA-order) occurrences of `query`, a slice of table() (:223-259)."""
        if self.len() == 0 or len(_as_bytes(query)) == 0:
            return self._table[0:0]
        s, e = self.positions_batch([query])
        return self._table[int(s[0]):int(e[0])]

    def any_position(self, query):
        if self.len() == 0 or len(_as_bytes(query)) == 0:
            return None                                              # :281-283
        _, anyp = self.contains_batch([query])
        return None if int(anyp[0]) == _NONE else int(anyp[0])

    def contains(self, query):
        return self.any_position(query) is not None                  # :197-199

    def __repr__(self):                                              # Debug, :296-312
        lines = ["", "-----------------------------------------", "SUFFIX TABLE",
                 f"text: {self.text()}"]
        for rank, s in enumerate(self._table.tolist()):
            lines.append(f"suffix[{rank}] {s}, {self._text[s:].decode('utf-8', 'replace')}")
        lines.append("-----------------------------------------")
        return "\n".join(lines) + "\n"
