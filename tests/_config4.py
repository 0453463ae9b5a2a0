"""BASELINE config 4 rehearsed on ONE GPU: n bytes of DNA (4 * 10^9 at full size: positions >= 2^31, a 1 GB packed text under
the range filter, slices of 10^9 suffixes -- above the hybrid route's 2^28 and inside the chunked radix schedule), cut into
`world` ranges exactly as suffix_amd/dist.py cuts them (byte histogram -> packed text -> 2^tb-bin key histogram ->
plan_ranges), every range built by sfx_build_sa_range_packed_u32_dev as one rank would build it, one after another.
Checks: every slice equals its stretch of ONE single-GPU build of the same text (whose own gate is permutation + every
adjacent pair in order, boundaries included: bench.verify_sa_chunked); the slices' sizes sum to n; sha256 of the concatenated
slices = sha256 of the single-GPU array; the u64 widening of a slice holds the same positions; and (round 5) sha256 of the
complete array = the pin that scripts/cpu_config4_oracle.py made from oracle.sais over the same 4 * 10^9 bytes (u32 positions
fit: SURVEY 8(d)'s "u32-oracle cross-check"), with per-2^28-entry chunk hashes to localise a difference.  Returns one record
per rank (+ one for the whole)."""
import ctypes
import hashlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _pins():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "fullsize_pins.json")) as f:
        return json.load(f)


def rehearse(n, world=4, tb=14, seed=0x5AF1C5 + 4, full_gate=True):
    import _gen
    import bench
    import suffix_amd
    from suffix_amd import device as sdev
    from suffix_amd import dist as sdist
    from suffix_amd.device import _p

    eng = suffix_amd.default_engine()
    eng.require_device()
    dev = torch.device("cuda", 0)
    recs = []
    t0 = time.perf_counter()
    host = _gen.dna_fast(n, seed=seed)
    sha_text = hashlib.sha256(memoryview(host)).hexdigest()
    text = torch.from_numpy(host).to(dev)
    del host
    gen_s = time.perf_counter() - t0

    # ---- the single-GPU build of the same text: the array every slice is compared with
    ws = sdev.sa_workspace(n, dev)
    full = torch.empty(n, dtype=torch.int32, device=dev)
    sdev.build_sa(text, out=full, workspace=ws)                   # first touch of the workspace
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sdev.build_sa(text, out=full, workspace=ws)
    torch.cuda.synchronize()
    single_ms = (time.perf_counter() - t0) * 1e3
    del ws
    torch.cuda.empty_cache()
    if full_gate:
        ok, how = bench.verify_sa_chunked(torch, sdev, text, full)
        assert ok, how
    else:
        how = "skipped"

    # ---- what dist.py does before the range build, with the collectives of a 1-rank world
    bb = torch.zeros(256, dtype=torch.int64, device=dev)
    eng.check(eng.lib.sfx_byte_histogram_dev(_p(text), 0, n, _p(bb), None), "sfx_byte_histogram_dev")
    sigma = int((bb > 0).sum())
    sym_bits = max(1, (max(sigma, 2) - 1).bit_length())
    spw = 32 // sym_bits
    nwords = (n + spw - 1) // spw
    packed = torch.zeros(nwords + 4, dtype=torch.int32, device=dev)
    scratch = torch.empty(256, dtype=torch.uint8, device=dev)
    eng.check(eng.lib.sfx_pack_text_dev(_p(text), n, _p(bb), _p(scratch), _p(packed), nwords, None), "sfx_pack_text_dev")
    kb = torch.zeros(1 << tb, dtype=torch.int64, device=dev)
    eng.check(eng.lib.sfx_key_histogram_dev(_p(text), n, 0, n, _p(bb), tb, _p(kb), None), "sfx_key_histogram_dev")
    plan = sdist.plan_ranges(kb.cpu(), world)
    assert sum(c for _, _, _, c in plan) == n and plan[0][2] == 0

    h_slices = hashlib.sha256()
    total = 0
    for rank, (lo, hi, off, cnt) in enumerate(plan):
        assert off == total
        cap = max(cnt, 1)
        part = torch.empty(cap, dtype=torch.int32, device=dev)
        wsr = torch.empty(int(eng.lib.sfx_sa_range_workspace_bytes(n, cap)), dtype=torch.uint8, device=dev)
        got = ctypes.c_uint64(0)
        ms = []
        for rep in range(2):                                       # (the second run is the timed one: workspace touched)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rc = eng.lib.sfx_build_sa_range_packed_u32_dev(_p(packed), n, _p(bb), tb, lo, hi, cap, _p(part), ctypes.byref(got),
                                                           _p(wsr), wsr.numel(), None)
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t0) * 1e3)
            eng.check(rc, "sfx_build_sa_range_packed_u32_dev")
        stats = eng.build_stats()
        assert int(got.value) == cnt, (rank, int(got.value), cnt)
        same = bool(torch.equal(part[:cnt], full[off:off + cnt]))
        # config 4 asks for u64 indices: the widened slice holds the same positions
        wide = sdev.widen_u64(part[:cnt], engine=eng)
        widened_ok = bool(torch.equal(wide & 0xFFFFFFFF, part[:cnt].to(torch.int64) & 0xFFFFFFFF)) and bool((wide >> 32 == 0).all())
        top = int((part[:cnt].to(torch.int64) & 0xFFFFFFFF).max()) if cnt else 0
        del wide
        h_slices.update(memoryview(part[:cnt].cpu().numpy()))
        recs.append({"rank": rank, "world": world, "n": n, "bins": [int(lo), int(hi)], "offset": int(off), "count": int(cnt),
                     "range_build_ms": round(ms[1], 2), "first_run_ms": round(ms[0], 2), "equals_single_gpu_slice": same,
                     "u64_widening_ok": widened_ok, "largest_position": top, "rounds": stats.get("rounds"),
                     "key_bits": stats.get("key_bits"), "active_after_initial": stats.get("active_after_initial")})
        assert same, f"slice of rank {rank} differs from the single-GPU suffix array"
        assert widened_ok
        total += cnt
        del part, wsr
        torch.cuda.empty_cache()
    assert total == n
    full_host = full.cpu().numpy()
    sha_full = hashlib.sha256(memoryview(full_host)).hexdigest()
    assert h_slices.hexdigest() == sha_full
    # the ORACLE's array of the same text (scripts/cpu_config4_oracle.py: oracle.sais over all n bytes, u32 positions since
    # n < 2^32 -- SURVEY 8(d)'s "u32-oracle cross-check"), pinned as sha256 of the whole array and of every 2^28-entry chunk
    pin = _pins().get("c4", {}).get(str(n))
    oracle_pinned = False
    if pin is not None:
        assert sha_text == pin["sha256_text"], "the generator no longer makes the pinned text"
        if sha_full != pin["sha256_sa"]:
            bad = [k for k, want in enumerate(pin["sha256_sa_chunks_2p28"])
                   if hashlib.sha256(memoryview(full_host[k << 28:(k + 1) << 28])).hexdigest() != want]
            raise AssertionError("config 4: suffix array differs from the oracle's in 2^28-entry chunks %r" % bad)
        oracle_pinned = True
    del full_host
    per = [r["range_build_ms"] for r in recs]
    recs.append({"summary": "config 4 input, %d virtual ranks on one GPU" % world, "n": n, "gen_s": round(gen_s, 1),
                 "single_gpu_build_ms": round(single_ms, 1), "single_gpu_gate": how, "range_build_ms_per_rank": per,
                 "max_rank_ms": max(per), "sum_rank_ms": round(sum(per), 1),
                 "compute_efficiency_vs_single_gpu_share": round(single_ms / world / max(per), 3),
                 "sha256_sa": sha_full, "sha256_of_concatenated_slices_equal": True, "sha256_text": sha_text,
                 "sa_equals_oracle_pin": oracle_pinned})
    del full, text, packed
    torch.cuda.empty_cache()
    return recs
