"""Device-resident entry points: torch tensors are used only as HBM buffers and
for the current HIP stream; all compute is the C ABI's `*_dev` functions.

Inputs stay in HBM, outputs are left in HBM, scratch comes from a caller-owned
(or freshly allocated) workspace tensor -- the engine itself never allocates
device memory on this path, so a build is a pure kernel sequence on the
caller's stream.
"""
import contextlib
import ctypes

import torch

from ._lib import default_engine


def _on(t):
    """Make the tensor's device the current one for the duration of an engine call: the C side launches
    on, allocates pinned staging for and pools scratch by the CURRENT device, and the stream handed over
    belongs to the tensor's device."""
    if t is not None and t.is_cuda:
        return torch.cuda.device(t.device)
    return contextlib.nullcontext()


def _stream_ptr(t):
    if t.is_cuda:
        return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return ctypes.c_void_p(0)


def _p(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None and t.numel() else 0)


def _check_u8(text):
    if text.dtype != torch.uint8 or text.dim() != 1 or not text.is_contiguous():
        raise TypeError("text must be a contiguous 1-D uint8 tensor")


def sa_workspace(n, device, engine=None):
    eng = engine or default_engine()
    return torch.empty(int(eng.lib.sfx_sa_workspace_bytes(int(n))), dtype=torch.uint8, device=device)


def build_sa(text, out=None, workspace=None, engine=None):
    """Suffix array (uint32, viewed through torch.int32 storage) of a uint8 tensor."""
    eng = engine or default_engine()
    _check_u8(text)
    n = text.numel()
    if text.is_cuda:
        eng.require_device()
    if out is None:
        out = torch.empty(n, dtype=torch.int32, device=text.device)
    if workspace is None:
        workspace = sa_workspace(n, text.device, eng)
    with _on(text):
        eng.check(eng.lib.sfx_build_sa_u32_dev(_p(text), n, _p(out), _p(workspace), workspace.numel(),
                                               _stream_ptr(text)), "sfx_build_sa_u32_dev")
    return out


def sa_lcp_workspace(n, device, engine=None):
    eng = engine or default_engine()
    return torch.empty(int(eng.lib.sfx_sa_lcp_workspace_bytes(int(n))), dtype=torch.uint8, device=device)


def build_sa_lcp(text, out_sa=None, out_lcp=None, workspace=None, engine=None):
    """SuffixTable::new + lcp_lens in one call: -> (sa, lcp), both uint32 in int32 storage."""
    eng = engine or default_engine()
    _check_u8(text)
    n = text.numel()
    if text.is_cuda:
        eng.require_device()
    if out_sa is None:
        out_sa = torch.empty(n, dtype=torch.int32, device=text.device)
    if out_lcp is None:
        out_lcp = torch.empty(n, dtype=torch.int32, device=text.device)
    if workspace is None:
        workspace = sa_lcp_workspace(n, text.device, eng)
    with _on(text):
        eng.check(eng.lib.sfx_build_sa_lcp_u32_dev(_p(text), n, _p(out_sa), _p(out_lcp), _p(workspace), workspace.numel(),
                                                   _stream_ptr(text)), "sfx_build_sa_lcp_u32_dev")
    return out_sa, out_lcp


def lcp_workspace(n, device, engine=None):
    eng = engine or default_engine()
    return torch.empty(int(eng.lib.sfx_lcp_workspace_bytes(int(n))), dtype=torch.uint8, device=device)


def build_lcp(text, sa, out=None, workspace=None, engine=None):
    eng = engine or default_engine()
    _check_u8(text)
    n = text.numel()
    if out is None:
        out = torch.empty(n, dtype=torch.int32, device=text.device)
    if workspace is None:
        workspace = lcp_workspace(n, text.device, eng)
    with _on(text):
        eng.check(eng.lib.sfx_build_lcp_u32_dev(_p(text), n, _p(sa), _p(out), _p(workspace),
                                                workspace.numel(), _stream_ptr(text)),
                  "sfx_build_lcp_u32_dev")
    return out


def query_batch(text, sa, qbytes, qoff, engine=None):
    """qbytes: uint8 tensor, qoff: int64 tensor of nq+1 offsets (same device as text).
    -> (start, end, found, any) tensors; positions(q_k) = sa[start[k]:end[k]]."""
    eng = engine or default_engine()
    nq = qoff.numel() - 1
    dev = text.device
    start = torch.empty(nq, dtype=torch.int32, device=dev)
    end = torch.empty(nq, dtype=torch.int32, device=dev)
    found = torch.empty(nq, dtype=torch.uint8, device=dev)
    anyp = torch.empty(nq, dtype=torch.int32, device=dev)
    with _on(text):
        eng.check(eng.lib.sfx_query_batch_dev(_p(text), text.numel(), _p(sa), _p(qbytes), _p(qoff), nq,
                                              _p(start), _p(end), _p(found), _p(anyp), _stream_ptr(text)),
                  "sfx_query_batch_dev")
    return start, end, found, anyp


class DeviceIndex:
    """Resident index over device tensors (text, suffix array): the engine adds its bucket directory
    (sfx_index_create_dev); `query` = batched positions() / contains() / any_position()."""

    def __init__(self, text, sa, engine=None):
        self._eng = engine or default_engine()
        _check_u8(text)
        self._text, self._sa = text, sa                     # (borrowed by the index: keep them alive)
        h = ctypes.c_void_p()
        with _on(text):
            self._eng.check(self._eng.lib.sfx_index_create_dev(_p(text), text.numel(), _p(sa), _stream_ptr(text),
                                                               ctypes.byref(h)), "sfx_index_create_dev")
        self._h = h

    def query(self, qbytes, qoff):
        nq = qoff.numel() - 1
        dev = self._text.device
        start = torch.empty(nq, dtype=torch.int32, device=dev)
        end = torch.empty(nq, dtype=torch.int32, device=dev)
        found = torch.empty(nq, dtype=torch.uint8, device=dev)
        anyp = torch.empty(nq, dtype=torch.int32, device=dev)
        with _on(self._text):
            self._eng.check(self._eng.lib.sfx_index_query_dev(self._h, _p(qbytes), _p(qoff), nq, _p(start), _p(end),
                                                              _p(found), _p(anyp), _stream_ptr(self._text)),
                            "sfx_index_query_dev")
        return start, end, found, anyp

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._eng.lib.sfx_index_destroy(h)

    __del__ = close


def lcp_intervals(lcp, engine=None):
    """Suffix-tree topology as flat arrays from the LCP array (uint32 in int32 storage, on any device the
    engine runs on): -> dict(lb, rb, node, parent, leaf_parent), see include/suffix_hip.h."""
    eng = engine or default_engine()
    n = lcp.numel()
    dev = lcp.device
    out = {k: torch.empty(n, dtype=torch.int32, device=dev) for k in ("lb", "rb", "node", "parent", "leaf_parent")}
    ws = torch.empty(int(eng.lib.sfx_lcp_intervals_workspace_bytes(n)), dtype=torch.uint8, device=dev)
    with _on(lcp):
        eng.check(eng.lib.sfx_lcp_intervals_dev(_p(lcp), n, _p(out["lb"]), _p(out["rb"]), _p(out["node"]), _p(out["parent"]),
                                                _p(out["leaf_parent"]), _p(ws), ws.numel(), _stream_ptr(lcp)),
                  "sfx_lcp_intervals_dev")
    return out


def doc_lookup(positions, doc_starts, engine=None):
    """Generalized suffix array: text positions (uint32 in int32 storage) -> (document index, offset inside it);
    doc_starts = sorted int64 start offsets of the documents inside the concatenated text."""
    eng = engine or default_engine()
    cnt = positions.numel()
    doc = torch.empty(cnt, dtype=torch.int32, device=positions.device)
    off = torch.empty(cnt, dtype=torch.int32, device=positions.device)
    with _on(positions):
        eng.check(eng.lib.sfx_doc_lookup_dev(_p(positions), cnt, _p(doc_starts), doc_starts.numel(), _p(doc), _p(off),
                                             _stream_ptr(positions)), "sfx_doc_lookup_dev")
    return doc, off


def widen_u64(sa32, out=None, engine=None):
    """u32 index tensor (int32 storage) -> int64 tensor holding the same indices (config 4)."""
    eng = engine or default_engine()
    if out is None:
        out = torch.empty(sa32.numel(), dtype=torch.int64, device=sa32.device)
    with _on(sa32):
        eng.check(eng.lib.sfx_widen_u32_to_u64_dev(_p(sa32), sa32.numel(), _p(out), _stream_ptr(sa32)),
                  "sfx_widen_u32_to_u64_dev")
    return out


def build_lcp_range(text, sa_part, prev_suffix=None, engine=None):
    """LCP of one contiguous slice of the suffix array (direct comparison with the predecessor;
    prev_suffix = last suffix of the previous slice, None for the first slice)."""
    eng = engine or default_engine()
    _check_u8(text)
    out = torch.empty(sa_part.numel(), dtype=torch.int32, device=text.device)
    prev = 0xFFFFFFFF if prev_suffix is None else int(prev_suffix) & 0xFFFFFFFF
    with _on(text):
        eng.check(eng.lib.sfx_build_lcp_range_u32_dev(_p(text), text.numel(), _p(sa_part), sa_part.numel(), prev,
                                                      _p(out), _stream_ptr(text)), "sfx_build_lcp_range_u32_dev")
    return out


def query_batch_range(text, sa_part, qbytes, qoff, engine=None):
    """Like query_batch, against one contiguous slice of the suffix array: start/end index the slice."""
    eng = engine or default_engine()
    nq = qoff.numel() - 1
    dev = text.device
    start = torch.empty(nq, dtype=torch.int32, device=dev)
    end = torch.empty(nq, dtype=torch.int32, device=dev)
    found = torch.empty(nq, dtype=torch.uint8, device=dev)
    anyp = torch.empty(nq, dtype=torch.int32, device=dev)
    with _on(text):
        eng.check(eng.lib.sfx_query_batch_range_dev(_p(text), text.numel(), _p(sa_part), sa_part.numel(), _p(qbytes),
                                                    _p(qoff), nq, _p(start), _p(end), _p(found), _p(anyp),
                                                    _stream_ptr(text)), "sfx_query_batch_range_dev")
    return start, end, found, anyp
