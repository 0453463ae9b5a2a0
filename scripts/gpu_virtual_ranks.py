#!/usr/bin/env python3
"""Per-rank compute of the range-partitioned build on ONE GPU ("virtual ranks"): rank 0's
work for world = 1, 2, 4, 8 with 100 MB shards (text = world * 100 MB, the collectives
replaced by local reductions).  Shows how a rank's time grows with the text it must pack and
filter -- the compute side of the weak-scaling curve the driver measures on 8 GPUs."""
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _gen  # noqa: E402
import suffix_amd  # noqa: E402
import _devlib
from suffix_amd import dist as sdist  # noqa: E402
from suffix_amd.device import _p  # noqa: E402

eng = _devlib.engine()
eng.require_device()
dev = torch.device("cuda", 0)
m = 100_000_000
for world in (1, 2, 4, 8):
    n = m * world
    text = torch.cat([torch.from_numpy(_gen.dna(m, seed=0x5AF1C5 + 1 + r)) for r in range(world)]).to(dev)
    tb = 14
    torch.cuda.synchronize()

    def rank0():
        bb = torch.zeros(256, dtype=torch.int64, device=dev)
        for r in range(world):      # stands in for all-reduce(sum): every shard's histogram (1 per rank in reality)
            part = torch.zeros(256, dtype=torch.int64, device=dev)
            eng.check(eng.lib.sfx_byte_histogram_dev(_p(text), r * m, (r + 1) * m, _p(part), None), "bh")
            bb += part
            if r == 0:
                torch.cuda.synchronize(); t_bh = time.perf_counter()
        kb = torch.zeros(1 << tb, dtype=torch.int64, device=dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.check(eng.lib.sfx_key_histogram_dev(_p(text), n, 0, m, _p(bb), tb, _p(kb), None), "kh")
        torch.cuda.synchronize(); t_kh = time.perf_counter() - t0
        kb = kb * world             # uniform text: the other shards' histograms look the same
        lo, hi, off, cnt = sdist.plan_ranges(kb.cpu(), world)[0]
        cap = int(cnt * 1.05) + 1024
        part = torch.empty(cap, dtype=torch.int32, device=dev)
        ws = torch.empty(int(eng.lib.sfx_sa_range_workspace_bytes(n, cap)), dtype=torch.uint8, device=dev)
        got = ctypes.c_uint64(0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if packed_flow:
            # what dist.py does when shards pack into whole words: every rank packs its own shard
            # (here: rank 0's share of the work = one shard; the others' words are packed untimed)
            wps = m // 16
            eng.check(eng.lib.sfx_pack_text_dev(_p(text), m, _p(bb), _p(scratch), _p(packed), wps, None), "pack")
            torch.cuda.synchronize(); t_pk = time.perf_counter() - t0
            for r in range(1, world):
                eng.check(eng.lib.sfx_pack_text_dev(_p(text[r * m:]), m, _p(bb), _p(scratch), _p(packed[r * wps:]), wps, None), "pack")
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.check(eng.lib.sfx_build_sa_range_packed_u32_dev(_p(packed), n, _p(bb), tb, lo, hi, cap, _p(part),
                                                                ctypes.byref(got), _p(ws), ws.numel(), None), "range")
            torch.cuda.synchronize(); t_rb = time.perf_counter() - t0 + t_pk
        else:
            eng.check(eng.lib.sfx_build_sa_range_u32_dev(_p(text), n, _p(bb), tb, lo, hi, cap, _p(part), ctypes.byref(got),
                                                         _p(ws), ws.numel(), None), "range")
            torch.cuda.synchronize(); t_rb = time.perf_counter() - t0
        results[packed_flow] = part[:int(got.value)].clone()
        return t_kh, t_rb, int(got.value)

    results = {}
    packed = torch.zeros(n // 16 + 4, dtype=torch.int32, device=dev)
    scratch = torch.empty(256, dtype=torch.uint8, device=dev)
    packed_flow = False
    rank0()
    t_kh, t_rb_raw, got = rank0()
    packed_flow = True
    rank0()
    t_kh, t_rb, got = rank0()
    assert torch.equal(results[True], results[False]), "packed and raw range builds differ"
    eng.profile(True); eng.profile_reset(); rank0(); torch.cuda.synchronize()
    rep = {r["name"]: round(r["total_ms"], 3) for r in eng.profile_report()}
    eng.profile(False)
    print(json.dumps({"world": world, "n": n, "rank0_suffixes": got, "key_hist_ms": round(t_kh * 1e3, 3),
                      "range_build_ms": round(t_rb * 1e3, 3), "range_build_raw_text_ms": round(t_rb_raw * 1e3, 3), "kernel_ms": rep}), flush=True)
    del text
