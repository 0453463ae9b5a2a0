"""Range-partitioned suffix-array construction across the GPUs of one node:
one process per GPU, torch.distributed over RCCL ("nccl" backend on ROCm).

SA-IS induction does not shard (it sweeps all buckets left-to-right then
right-to-left, src/table.rs:426-448), but the *suffix array itself* does,
because it is unique: split the SA index space at bucket boundaries.

    1. every rank owns a contiguous shard of the byte stream (lengths may differ); the shards are
       all-gathered so each GPU holds the whole text in HBM (the one bulk
       exchange) -- as PACKED symbol codes (bits/8 of the raw bytes over xGMI)
       whenever a shard packs into whole words, after step 2 has fixed the codes;
    2. each rank histograms the bytes of ITS shard; all-reduce(sum) of 256 bins
       -> the global alphabet (dense symbol codes), identical everywhere;
    3. each rank histograms the top `top_bits` key bits of the suffixes starting
       in ITS shard; all-reduce(sum) of 2^top_bits bins = the bucket-boundary
       histogram exchange;
    4. the bins are cut into G contiguous ranges of (nearly) equal suffix count;
    5. rank r sorts only the suffixes of range r (radix sort + text refinement,
       no further communication) -> its slice SA[offset_r : offset_r + count_r].

The collectives move 2 KiB + 128 KiB per rank (latency-bound); the data path has
no collective after step 1.
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

from ._lib import default_engine
from .device import _p, _stream_ptr

TOP_BITS = 14
SFX_ERR_NEEDS_RANKS = 7


def plan_ranges(bins, world):
    """Cut cumulative bin counts into `world` contiguous bin ranges with balanced totals.
    bins: 1-D int64 CPU tensor.  -> list of (bin_lo, bin_hi, offset, count)."""
    csum = torch.cumsum(bins, 0)
    total = int(csum[-1]) if bins.numel() else 0
    cuts = [0]
    for r in range(1, world):
        target = (total * r) // world
        # first bin index whose cumulative count exceeds the target
        idx = int(torch.searchsorted(csum, torch.tensor(target, dtype=csum.dtype), right=True))
        cuts.append(max(cuts[-1], min(idx, bins.numel())))
    cuts.append(bins.numel())
    out = []
    for r in range(world):
        lo, hi = cuts[r], cuts[r + 1]
        off = int(csum[lo - 1]) if lo > 0 else 0
        cnt = (int(csum[hi - 1]) if hi > 0 else 0) - off
        out.append((lo, hi, off, cnt))
    return out


def _gather_flat(dst, src, group=None, async_op=False):
    """All-gather equal-sized pieces into one flat receive buffer (RCCL's native form); backends
    without it fall back to the list form.  Errors of the collective itself propagate."""
    fn = getattr(dist, "all_gather_into_tensor", None)
    if fn is not None:
        try:
            return fn(dst, src, group=group, async_op=async_op)
        except (NotImplementedError, AttributeError):
            pass
        except RuntimeError as e:
            # older stacks raise RuntimeError for a backend without the flat form; anything else is the collective's own
            msg = str(e).lower()
            if not any(w in msg for w in ("not supported", "unsupported", "not implemented", "does not support")):
                raise
    return dist.all_gather(list(dst.split(src.numel())), src, group=group, async_op=async_op)


def gather_shards(shard, lens, group=None):
    """The whole text on every rank from shards of possibly different lengths `lens` (list of ints,
    one per rank): padded all-gather + compaction when they differ."""
    world = len(lens)
    dev = shard.device
    n = sum(lens)
    text = torch.empty(n, dtype=torch.uint8, device=dev)
    if len(set(lens)) == 1:
        _gather_flat(text, shard.contiguous(), group)
        return text
    mx = max(lens)
    padded = torch.zeros(mx, dtype=torch.uint8, device=dev)
    padded[:shard.numel()] = shard
    allp = torch.empty(mx * world, dtype=torch.uint8, device=dev)
    _gather_flat(allp, padded, group)
    off = 0
    for r, ln in enumerate(lens):
        text[off:off + ln] = allp[r * mx:r * mx + ln]
        off += ln
    return text


class _Phase:
    """Per-phase wall times of a partitioned build (bench.py, N > 1): the device is synchronised at
    every phase boundary only when a `timings` dict is passed."""

    def __init__(self, timings, dev):
        self.t, self.dev, self.t0 = timings, dev, None
        if timings is not None:
            import time
            self._now = time.perf_counter
            self._sync()
            self.t0 = self._now()

    def _sync(self):
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)

    def mark(self, name):
        if self.t is None:
            return
        self._sync()
        t1 = self._now()
        self.t[name] = self.t.get(name, 0.0) + (t1 - self.t0) * 1e3
        self.t0 = t1


def build_sa_partitioned(shard, group=None, engine=None, top_bits=TOP_BITS, return_text=False,
                         index_dtype=torch.int32, timings=None):
    """shard: this rank's contiguous uint8 piece of the text, on this rank's device; the shards may
    have different lengths (the text is their concatenation in rank order).  Returns
    (sa_part, offset, n): sa_part is an int32-storage tensor holding u32 suffix indices, the slice
    SA[offset : offset + sa_part.numel()] of the global suffix array.
    index_dtype=torch.int64 widens the slice to u64 indices (BASELINE config 4);
    return_text=True appends the all-gathered text (needed for LCP / queries);
    timings = {} collects per-phase milliseconds (byte_hist, all_gather_issue, key_hist, all_gather_wait, plan, range_build:
    every value a number) and, under the one non-numeric key timings["info"], notes as strings ("text_exchange": how the
    text travelled; "fallback": present when every rank built the whole array)."""
    eng = engine or default_engine()
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = shard.device
    m = shard.numel()
    if dev.type == "cuda":
        torch.cuda.set_device(dev)        # the engine launches on / pools by the CURRENT device (one rank = one GPU)
    ph = _Phase(timings, dev)
    # shard lengths: one tiny all-gather (the byte stream need not divide evenly)
    lens_t = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(lens_t, torch.tensor([m], dtype=torch.int64, device=dev), group=group)
    lens = [int(x) for x in lens_t]
    n = sum(lens)
    begin = sum(lens[:rank])
    if n > 0xFFFFFFFF:
        raise OverflowError("text longer than u32::MAX bytes")     # src/table.rs:380
    stream = _stream_ptr(shard)

    # 2. global alphabet: byte histogram of the own shard, all-reduced
    shard = shard.contiguous()
    byte_bins = torch.zeros(256, dtype=torch.int64, device=dev)
    eng.check(eng.lib.sfx_byte_histogram_dev(_p(shard), 0, m, _p(byte_bins), stream), "sfx_byte_histogram_dev")
    dist.all_reduce(byte_bins, op=dist.ReduceOp.SUM, group=group)
    sigma = int((byte_bins > 0).sum())
    sym_bits = max(1, (max(sigma, 2) - 1).bit_length())
    spw = 32 // sym_bits
    tb = min(top_bits, sym_bits * max(1, spw))
    ph.mark("byte_hist")

    # 1. the text on every GPU -- as packed symbol codes (bits/8 of the raw volume over xGMI, and no rank packs the
    #    whole text) whenever every shard has at least 64 bytes; raw otherwise or on request.  A rank packs the words of
    #    the GLOBAL word grid that lie wholly inside its shard; a word that straddles two shards (ragged shards, or shard
    #    lengths that are no multiple of the symbols per word) and the last, partial word of the text are assembled on
    #    every rank from the first / last 64 bytes of each shard, which are exchanged anyway (halo of the key histogram).
    packed_path = min(lens) >= 64
    text = None
    if packed_path:
        # key bits near the end of a shard reach into the next shard: a 64-byte halo is enough
        edge = torch.cat([shard[:64], shard[-64:]]).contiguous()
        edges = torch.empty(128 * world, dtype=torch.uint8, device=dev)
        _gather_flat(edges, edge, group)
        if rank + 1 < world:
            local = torch.cat([shard, edges[128 * (rank + 1):128 * (rank + 1) + 64]])
        else:
            local = shard
        hist_text, hist_n, hist_lo, hist_hi = local, local.numel(), 0, m
    if not packed_path or return_text:
        text = gather_shards(shard, lens, group)
    if not packed_path:
        hist_text, hist_n, hist_lo, hist_hi = text, n, begin, begin + m

    # the big exchange (the packed shards) starts now and runs under the key histogram
    packed = mine = packed_work = allw = None
    aligned = False
    if packed_path:
        begins = [sum(lens[:r]) for r in range(world)]
        q0 = [(b + spw - 1) // spw for b in begins]                # first word wholly inside shard r
        q1 = [(b + ln) // spw for b, ln in zip(begins, lens)]      # one past its last whole word
        nfull = [max(0, e - a) for a, e in zip(q0, q1)]
        aligned = all(b % spw == 0 for b in begins) and len(set(nfull)) == 1 and n % spw == 0
        packed = torch.zeros((n + spw - 1) // spw + 4, dtype=torch.int32, device=dev)   # + the zero tail keys read into
        scratch = torch.empty(256, dtype=torch.uint8, device=dev)
        mxw = max(nfull)
        mine = torch.zeros(max(mxw, 1), dtype=torch.int32, device=dev)
        skip = q0[rank] * spw - begin
        if nfull[rank]:
            eng.check(eng.lib.sfx_pack_text_dev(_p(shard[skip:]), nfull[rank] * spw, _p(byte_bins), _p(scratch), _p(mine),
                                                nfull[rank], stream), "sfx_pack_text_dev")
        if aligned:
            packed_work = _gather_flat(packed[:mxw * world], mine, group, async_op=True)
        else:
            allw = torch.empty(max(mxw, 1) * world, dtype=torch.int32, device=dev)
            packed_work = _gather_flat(allw, mine, group, async_op=True)
    if timings is not None:
        timings.setdefault("info", {})["text_exchange"] = ("packed words" + ("" if aligned else " (ragged word grid)")) if packed_path else "raw bytes"
    ph.mark("all_gather_issue")

    # 3. bucket-boundary histogram
    key_bins = torch.zeros(1 << tb, dtype=torch.int64, device=dev)
    eng.check(eng.lib.sfx_key_histogram_dev(_p(hist_text), hist_n, hist_lo, hist_hi, _p(byte_bins), tb,
                                            _p(key_bins), stream), "sfx_key_histogram_dev")
    ph.mark("key_hist")
    if packed_work is not None:
        packed_work.wait()
    if allw is not None:
        # ragged: every rank's whole words to their place in the global word grid, then the words that straddle shards
        mxw = max(max(nfull), 1)
        for r in range(world):
            if nfull[r]:
                packed[q0[r]:q0[r] + nfull[r]] = allw[r * mxw:r * mxw + nfull[r]]
        eb = edges.cpu().numpy()
        present = (byte_bins > 0).cpu().numpy()
        code = np.cumsum(present) - 1                              # dense symbol codes, as k_make_lut assigns them
        def symbol(p):                                             # code of text position p near a shard boundary
            r = max(k for k in range(world) if begins[k] <= p)
            o = p - begins[r]
            if o < 64:
                return int(code[eb[128 * r + o]])
            back = begins[r] + lens[r] - p                         # 1 .. 64 from the end of shard r
            assert 1 <= back <= 64, (p, r)
            return int(code[eb[128 * r + 128 - back]])
        idx, val = [], []
        for q in sorted(set([b // spw for b in begins[1:] if b % spw] + ([n // spw] if n % spw else []))):
            w = 0
            for j in range(spw):
                p = q * spw + j
                if p < n:
                    w |= symbol(p) << ((spw - 1 - j) * sym_bits)
            idx.append(q)
            val.append(w - (1 << 32) if w >= (1 << 31) else w)
        if idx:
            packed[torch.tensor(idx, dtype=torch.int64, device=dev)] = torch.tensor(val, dtype=torch.int32, device=dev)
    ph.mark("all_gather_wait")
    dist.all_reduce(key_bins, op=dist.ReduceOp.SUM, group=group)

    # 4. plan (tiny, on the host; identical on every rank)
    lo, hi, offset, count = plan_ranges(key_bins.cpu(), world)[rank]
    ph.mark("plan")

    # 5. this rank's slice
    cap = max(count, 1)
    sa_part = torch.empty(cap, dtype=torch.int32, device=dev)
    ws = torch.empty(int(eng.lib.sfx_sa_range_workspace_bytes(n, cap)), dtype=torch.uint8, device=dev)
    got = ctypes.c_uint64(0)
    if packed_path:
        rc = eng.lib.sfx_build_sa_range_packed_u32_dev(_p(packed), n, _p(byte_bins), tb, lo, hi, cap, _p(sa_part),
                                                       ctypes.byref(got), _p(ws), ws.numel(), stream)
    else:
        rc = eng.lib.sfx_build_sa_range_u32_dev(_p(text), n, _p(byte_bins), tb, lo, hi, cap, _p(sa_part),
                                                ctypes.byref(got), _p(ws), ws.numel(), stream)
    # a slice whose repeats are too long for text-symbol refinement needs ranks of suffixes that other
    # ranks own: every rank then builds the whole suffix array (replicas) and keeps its slice -- correct
    # for any text, not scalable, and announced in `timings`
    # (a rank whose build failed for another reason must not be dragged into the fallback: its error is the answer)
    failed = rc not in (0, SFX_ERR_NEEDS_RANKS)
    need = torch.tensor([1 if rc == SFX_ERR_NEEDS_RANKS else 0, 1 if failed else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(need, op=dist.ReduceOp.MAX, group=group)
    if failed:
        eng.check(rc, "sfx_build_sa_range_u32_dev")
    if int(need[1].item()):
        raise RuntimeError(f"rank {rank}: the range build failed on another rank")
    if int(need[0].item()):
        import warnings
        warnings.warn("suffix_amd.dist: a slice needs rank refinement (a repeat of thousands of symbols): every rank builds "
                      "the whole suffix array and keeps its slice -- correct, but ~50 n bytes of workspace per GPU and no "
                      "multi-GPU speed-up for this text", RuntimeWarning, stacklevel=2)
        del ws
        if text is None:
            text = gather_shards(shard, lens, group)
        from .device import build_sa
        full = build_sa(text, engine=eng)
        part = full[offset:offset + count].clone()
        del full
        if timings is not None:
            timings.setdefault("info", {})["fallback"] = "replicated build (a slice needed rank refinement)"
    else:
        eng.check(rc, "sfx_build_sa_range_u32_dev")
        if int(got.value) != count:
            raise RuntimeError(f"rank {rank}: range build produced {got.value} suffixes, plan said {count}")
        part = sa_part[:count]
    ph.mark("range_build")
    if index_dtype == torch.int64:
        from .device import widen_u64
        part = widen_u64(part, engine=eng)
    return (part, offset, n, text) if return_text else (part, offset, n)


def previous_slice_last(sa_part, group=None):
    """The suffix that precedes this rank's slice in the global suffix array: the last element of
    the nearest non-empty slice before it (None for the first).  One index per rank, all-gathered."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = sa_part.device
    mine = torch.tensor([int(sa_part[-1]) & 0xFFFFFFFF if sa_part.numel() else -1], dtype=torch.int64, device=dev)
    lasts = [torch.empty(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(lasts, mine, group=group)
    for r in range(rank - 1, -1, -1):
        if int(lasts[r]) >= 0:
            return int(lasts[r])
    return None


_FETCH = object()


def build_lcp_partitioned(text, sa_part, group=None, engine=None, prev=_FETCH):
    """LCP of this rank's slice of the partitioned suffix array.  The only exchange is one
    suffix index per rank (previous_slice_last; pass prev= to reuse one already fetched).
    text = the all-gathered text."""
    from .device import build_lcp_range
    if prev is _FETCH:
        prev = previous_slice_last(sa_part, group)
    return build_lcp_range(text, sa_part, prev, engine=engine)


def verify_partitioned(shard, sa_part, n, group=None, engine=None):
    """Size-independent correctness gate for a partitioned suffix array (bench.py, N > 1; outside
    any timed region): (1) the slices together are a permutation of 0..n-1 (per-rank marks,
    all-reduced); (2) every adjacent pair -- inside the slices and across their boundaries -- is
    in strictly increasing suffix order, with the engine's per-slice LCP giving the position of
    the first difference.  -> (ok, description), identical on every rank."""
    world = dist.get_world_size(group)
    dev = shard.device
    lens_t = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(lens_t, torch.tensor([shard.numel()], dtype=torch.int64, device=dev), group=group)
    text = gather_shards(shard.contiguous(), [int(x) for x in lens_t], group)
    assert text.numel() == n
    cur = sa_part.view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    marks = torch.zeros(n, dtype=torch.int32, device=dev)
    marks.index_add_(0, cur, torch.ones(cur.numel(), dtype=torch.int32, device=dev))
    dist.all_reduce(marks, op=dist.ReduceOp.SUM, group=group)
    why = None
    if not bool((marks == 1).all()):
        why = "not a permutation"
    del marks
    prev = previous_slice_last(sa_part, group)
    if why is None and cur.numel():
        lcp = build_lcp_partitioned(text, sa_part, group, engine, prev=prev)
        h = lcp.view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        before = torch.empty_like(cur)
        before[1:] = cur[:-1]
        before[0] = prev if prev is not None else 0
        skip = 1 if prev is None else 0
        pa, pb = (before + h)[skip:], (cur + h)[skip:]
        if bool((pb >= n).any()):
            why = "right suffix exhausted before left one"
        else:
            ca = text[torch.clamp(pa, max=n - 1)].to(torch.int32)
            cb = text[pb].to(torch.int32)
            if not bool(((pa >= n) | (ca < cb)).all()):
                why = "adjacent suffixes out of order"
    bad = torch.tensor([0 if why is None else 1], dtype=torch.int64, device=dev)
    dist.all_reduce(bad, op=dist.ReduceOp.SUM, group=group)
    if int(bad.item()):
        return False, why or "another rank's slice failed"
    return True, f"permutation (marks all-reduced over {world} ranks) + adjacent order of every pair, slice boundaries included"


def positions_partitioned(text, sa_part, offset, qbytes, qoff, group=None, engine=None):
    """positions()/contains() against the partitioned index: every rank searches its own
    slice for ALL queries, then one all-reduce (SUM of match counts, MIN of global starts)
    assembles the global SA intervals -- matches are contiguous in the suffix array, so the
    slices that hold some of them are adjacent.  -> (start, end) int64 tensors, global SA
    positions, start == end == 0 for no match (as sfx_positions_batch)."""
    from .device import query_batch_range
    s, e, _found, _any = query_batch_range(text, sa_part, qbytes, qoff, engine=engine)
    s64 = s.to(torch.int64) & 0xFFFFFFFF
    e64 = e.to(torch.int64) & 0xFFFFFFFF
    cnt = e64 - s64
    big = torch.iinfo(torch.int64).max
    gstart = torch.where(cnt > 0, s64 + int(offset), torch.full_like(s64, big))
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(gstart, op=dist.ReduceOp.MIN, group=group)
    gstart = torch.where(cnt > 0, gstart, torch.zeros_like(gstart))
    return gstart, gstart + cnt
