#!/usr/bin/env python3
"""Full-size runs of BASELINE.json configs 3 and 5 (and a 1 GB DNA control) on one
MI355X: device-resident SA (+LCP, + 1M batched queries), size-independent property
checks, sampled comparison with the oracle.  Writes gpurun_out/big/results.jsonl.
Lives under tests/ because it uses the oracle (as checker and as the timed CPU baseline);
it is a script, not a pytest module:
    gpurun --timeout 1500 -- 'python tests/fullsize_configs.py [c3 c5 dna1g dna2g eng2g c4single]'"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _gen  # noqa: E402
import bench  # noqa: E402
import suffix_amd  # noqa: E402
from suffix_amd import device as sdev  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out", "big")
os.makedirs(OUT, exist_ok=True)
eng = suffix_amd.default_engine()
eng.require_device()
dev = torch.device("cuda", 0)


def timed(fn, reps=1):
    torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return r, best


def run(name, host_text, with_lcp=True, queries=0, reps=2, cpu_sample=20_000_000):
    import hashlib
    n = host_text.size
    rec = {"config": name, "n": int(n), "sha256_text": hashlib.sha256(host_text.tobytes()).hexdigest()}
    text = torch.from_numpy(host_text).to(dev)
    ws = sdev.sa_workspace(n, dev)
    sa = torch.empty(n, dtype=torch.int32, device=dev)
    _, t_sa = timed(lambda: sdev.build_sa(text, out=sa, workspace=ws), reps)
    rec["sa_ms"] = round(t_sa * 1e3, 2)
    rec["sa_MBps"] = round(n / t_sa / 1e6, 1)
    rec["build"] = eng.build_stats()
    eng.profile(True); eng.profile_reset()
    sdev.build_sa(text, out=sa, workspace=ws); torch.cuda.synchronize()
    rec["sa_kernel_ms"] = {r["name"]: round(r["total_ms"], 3) for r in eng.profile_report()}
    eng.profile(False)
    del ws
    if with_lcp:
        lws = sdev.lcp_workspace(n, dev)
        lcp = torch.empty(n, dtype=torch.int32, device=dev)
        _, t_lcp = timed(lambda: sdev.build_lcp(text, sa, out=lcp, workspace=lws), reps)
        rec["lcp_ms"] = round(t_lcp * 1e3, 2)
        rec["lcp_MBps"] = round(n / t_lcp / 1e6, 1)
        rec["sa_plus_lcp_MBps"] = round(n / (t_sa + t_lcp) / 1e6, 1)
        rec["sha256_lcp"] = hashlib.sha256(lcp.cpu().numpy().tobytes()).hexdigest()
        rec["max_lcp"] = int((lcp.to(torch.int64) & 0xFFFFFFFF).max())
        rec["mean_lcp"] = float((lcp.to(torch.int64) & 0xFFFFFFFF).double().mean())
        del lcp, lws
    ok, how = bench.verify_sa_on_device(torch, sdev, text, sa) if n <= 1_000_000_000 else \
        bench.verify_sa_chunked(torch, sdev, text, sa)
    rec["verified"], rec["verification"] = bool(ok), how
    rec["sha256_sa"] = hashlib.sha256(sa.cpu().numpy().tobytes()).hexdigest()      # LE u32, as SURVEY.md 8c
    if queries:
        import oracle
        rng = np.random.default_rng(17)
        if os.environ.get("SFX_R1_QUERIES") == "1":       # round 1's query sample (numpy), for its pinned results
            starts = rng.integers(0, n - 64, size=queries)
            lens = rng.integers(1, 33, size=queries)
            for _ in range(3):
                starts = np.where((host_text[starts] & 0xC0) == 0x80, starts + 1, starts)
            ends = starts + lens
            for _ in range(3):
                ends = np.where((host_text[np.minimum(ends, n - 1)] & 0xC0) == 0x80, ends + 1, ends)
            lens = (ends - starts).astype(np.int64)
            off = np.zeros(queries + 1, dtype=np.int64)
            off[1:] = np.cumsum(lens)
            idx = np.arange(off[-1], dtype=np.int64) - np.repeat(off[:-1], lens) + np.repeat(starts, lens)
            qb = host_text[idx].copy()
            miss = np.arange(queries // 2, queries)
            qb[off[miss + 1] - 1] ^= 0x15
        else:                                              # code-point substrings, half of them corrupted
            qb, off = _gen.queries(host_text, queries)
        d_qb, d_off = torch.from_numpy(qb).to(dev), torch.from_numpy(off).to(dev)
        (s, e, f, a), t_q = timed(lambda: sdev.query_batch(text, sa, d_qb, d_off), reps)
        rec["queries"] = queries
        rec["query_ms"] = round(t_q * 1e3, 2)
        rec["Mqueries_per_s"] = round(queries / t_q / 1e6, 2)
        rec["hit_fraction"] = float(f.float().mean())
        s, e, f = s.cpu().numpy(), e.cpu().numpy(), f.cpu().numpy()
        sa_h = sa.cpu().numpy().view(np.uint32)
        tb = host_text.tobytes()
        bad = 0
        for k in rng.integers(0, queries, size=1500).tolist():
            q = qb[off[k]:off[k + 1]].tobytes()
            if (int(s[k]), int(e[k])) != oracle.positions(tb, sa_h, q):
                bad += 1
        rec["query_mismatches_vs_oracle_of_1500"] = bad
    # CPU baseline beside it: the oracle = C restatement of the reference, 1 thread,
    # on a bounded sample (first 20 MB) of the same text: SA (sais), LCP as the reference computes it
    # (lcp_lens_quadratic) and Kasai; the same sample's GPU SA is compared bit-exactly
    if cpu_sample:
        import oracle
        m = min(cpu_sample, n)
        sample = host_text[:m]
        t0 = time.perf_counter(); exp = oracle.sais(sample); t_sa_cpu = time.perf_counter() - t0
        t0 = time.perf_counter(); lq = oracle.lcp_quadratic(sample, exp); t_lq = time.perf_counter() - t0
        t0 = time.perf_counter(); lk = oracle.lcp_kasai(sample, exp); t_lk = time.perf_counter() - t0
        sub = torch.from_numpy(np.ascontiguousarray(sample)).to(dev)
        got = sdev.build_sa(sub)
        got_lcp = sdev.build_lcp(sub, got).cpu().numpy().view(np.uint32)
        rec["cpu_baseline"] = {"sample_bytes": int(m), "kind": "port", "cores": 1,
                               "sa_MBps": round(m / t_sa_cpu / 1e6, 2), "lcp_quadratic_MBps": round(m / t_lq / 1e6, 2),
                               "lcp_kasai_MBps": round(m / t_lk / 1e6, 2),
                               "gpu_sa_bit_exact_on_sample": bool(np.array_equal(got.cpu().numpy().view(np.uint32), exp)),
                               "gpu_lcp_bit_exact_on_sample": bool(np.array_equal(got_lcp, lq) and np.array_equal(lq, lk))}
        del sub, got
    # SFX_FULL_ORACLE=1: the whole text through the oracle (one core, ~1 minute per GB) -- the
    # full-size CPU baseline and a bit-exact comparison of the complete SA and LCP arrays
    if os.environ.get("SFX_FULL_ORACLE") == "1":
        import oracle
        t0 = time.perf_counter(); exp = oracle.sais(host_text); t_cpu = time.perf_counter() - t0
        full = {"sa_seconds": round(t_cpu, 1), "sa_MBps": round(n / t_cpu / 1e6, 2),
                "sa_bit_exact": bool(np.array_equal(sa.cpu().numpy().view(np.uint32), exp))}
        if with_lcp:
            t0 = time.perf_counter(); lq = oracle.lcp_quadratic(host_text, exp); t_lq = time.perf_counter() - t0
            lcp2 = sdev.build_lcp(text, sa)
            full["lcp_quadratic_seconds"] = round(t_lq, 1)
            full["lcp_quadratic_MBps"] = round(n / t_lq / 1e6, 2)
            full["lcp_bit_exact"] = bool(np.array_equal(lcp2.cpu().numpy().view(np.uint32), lq))
            del lcp2, lq
        rec["full_oracle"] = full
        del exp
    print(json.dumps(rec), flush=True)
    with open(os.path.join(OUT, "results.jsonl"), "a") as fh:
        fh.write(json.dumps(rec) + "\n")
    del text, sa
    torch.cuda.empty_cache()


if __name__ == "__main__":
    which = sys.argv[1:] or ["c3", "c5", "dna1g"]
    size = int(os.environ.get("SFX_BIG_N", "1000000000"))
    if "c3" in which:
        t0 = time.time(); h = _gen.english_like(size); print("gen english", round(time.time() - t0, 1), "s", flush=True)
        run("config3: 1 GB English-like ASCII (SURVEY 8d generator), SA + LCP", h)
    if "c5" in which:
        t0 = time.time(); h = _gen.utf8_mixed(size); print("gen utf8", round(time.time() - t0, 1), "s", flush=True)
        run("config5: 1 GB UTF-8 mixed-script (SURVEY 8d generator), SA + LCP + 1M positions() queries", h,
            queries=1_000_000)
    if "dup" in which:
        t0 = time.time(); h = _gen.near_duplicates(size); print("gen near-duplicates", round(time.time() - t0, 1), "s", flush=True)
        run("high-LCP: 1 GB near-duplicate documents (16 x 1 MiB, one substitution per ~400 B), SA + LCP", h)
    if "c3r1" in which or "c5r1" in which:
        import _gen_r1
        if "c3r1" in which:
            t0 = time.time(); h = _gen_r1.english_like(size); print("gen english (r1)", round(time.time() - t0, 1), "s", flush=True)
            run("config3 (round-1 input): 1 GB English-like ASCII, SA + LCP", h)
        if "c5r1" in which:
            os.environ["SFX_R1_QUERIES"] = "1"
            t0 = time.time(); h = _gen_r1.utf8_mixed(size); print("gen utf8 (r1)", round(time.time() - t0, 1), "s", flush=True)
            run("config5 (round-1 input): 1 GB UTF-8 mixed-script, SA + LCP + 1M positions() queries", h, queries=1_000_000)
            os.environ["SFX_R1_QUERIES"] = "0"
    if "dna1g" in which:
        t0 = time.time(); h = _gen.dna_fast(size, seed=0x5AF1C5 + 4); print("gen dna", round(time.time() - t0, 1), "s", flush=True)
        run("control: 1 GB uniform DNA, SA + LCP", h)
    if "dna2g" in which:
        # n >= 2^30: the one-sweep status words no longer fit, the chunked radix schedule takes over;
        # also the largest single-GPU input tried (2 * 10^9 positions, ~93 GB of workspace)
        big = int(os.environ.get("SFX_HUGE_N", "2000000000"))
        t0 = time.time(); h = _gen.dna_fast(big, seed=0x5AF1C5 + 5); print("gen dna", round(time.time() - t0, 1), "s", flush=True)
        run(f"stress: {big} B uniform DNA (n >= 2^30: chunked radix schedule), SA + LCP", h, reps=1)
    if "eng2g" in which:
        big = int(os.environ.get("SFX_HUGE_N", "1500000000"))
        t0 = time.time(); h = _gen.english_like(big); print("gen english", round(time.time() - t0, 1), "s", flush=True)
        run(f"stress: {big} B English-like ASCII (n >= 2^30), SA + LCP", h, reps=1)
    if "c4single" in which:
        # BASELINE config 4's input (4 * 10^9 B of DNA, u64 indices) on ONE GPU: positions still fit
        # u32 (src/table.rs:380), so the u32 engine runs (~198 GB of workspace) and the array is widened
        import hashlib
        big = int(os.environ.get("SFX_HUGE_N", "4000000000"))
        t0 = time.time(); h = _gen.dna_fast(big, seed=0x5AF1C5 + 3); print("gen dna", round(time.time() - t0, 1), "s", flush=True)
        rec = {"config": f"config 4 input on one GPU: {big} B uniform DNA, u32 engine + u64 widening", "n": int(big),
               "sha256_text": hashlib.sha256(h.tobytes()).hexdigest()}
        text = torch.from_numpy(h).to(dev)
        ws = sdev.sa_workspace(big, dev)
        sa = torch.empty(big, dtype=torch.int32, device=dev)
        _, t_sa = timed(lambda: sdev.build_sa(text, out=sa, workspace=ws), 1)
        rec["sa_ms"] = round(t_sa * 1e3, 2); rec["sa_MBps"] = round(big / t_sa / 1e6, 1); rec["build"] = eng.build_stats()
        rec["workspace_GB"] = round(ws.numel() / 1e9, 1)
        del ws
        torch.cuda.empty_cache()
        ok, how = bench.verify_sa_chunked(torch, sdev, text, sa)
        rec["verified"], rec["verification"] = bool(ok), how
        (sa64), t_w = timed(lambda: sdev.widen_u64(sa), 1)
        rec["widen_u64_ms"] = round(t_w * 1e3, 2)
        k = torch.randint(0, big, (1_000_000,), device=dev)
        rec["u64_matches_u32_on_sample"] = bool(((sa[k].to(torch.int64) & 0xFFFFFFFF) == sa64[k]).all())
        print(json.dumps(rec), flush=True)
        with open(os.path.join(OUT, "results.jsonl"), "a") as fh:
            fh.write(json.dumps(rec) + "\n")
