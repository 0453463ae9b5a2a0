#!/usr/bin/env python3
"""bench.py -- headline benchmark of the suffix-array hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size BYTES]

A "step" is one full pass of the hot path -- SuffixTable::new, i.e. suffix-array
construction -- over one batch of synthetic input that is ALREADY RESIDENT IN
HBM when the timed region starts (device text -> device SA; no PCIe inside the
timed region).  N = 1 runs BASELINE.json configs[1]: 100 MB synthetic DNA
(sigma = 4, uniform, splitmix64 seed per SURVEY.md 8d), u32 indices.  N > 1
(launched by torch.distributed.run, one rank per GPU) runs the range-partitioned
build of suffix_amd/dist.py: every rank contributes a 100 MB shard, the job
builds the SA of the N*100 MB text, each rank producing its contiguous slice
("weak" scaling: suffixes sorted per GPU stay fixed).  --total-size T is the
STRONG series instead: ONE text of T bytes (uniform DNA, seed 0x5AF1C5 + 4 --
BASELINE config 4 at T = 4000000000) cut into N equal shards, at N = 1 the
single-GPU build of the same text; the line then says "scaling": "strong".

Rank 0 prints the full records on lines that start with "DETAIL " (headline, then one per full-size config) and, last,
ONE compact JSON line (< 4 KB; see DESIGN.md "Measurement" for every field) that carries, per config, {sa_ms, lcp_ms,
fused_ms, bit_exact, engine_BpB, engine_frac, whole_path_frac, dominant kernel} inside `roofline.configs`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
METRIC = "MB of input text/sec for SuffixTable::new (SA-IS+LCP), 1 GPU; bit-exact SA"


def verify_sa_on_device(torch, sdev, text, sa, n_samples=20000, seed=1):
    """Size-independent correctness gate for a full-size SA held in HBM:
    (1) it is a permutation of 0..n-1; (2) with LCP from the engine, every adjacent
    pair is in strictly increasing suffix order (next symbol after the common prefix
    is larger, or the left suffix ended); (3) the LCP values themselves are checked
    on the host, byte by byte, for a random sample of pairs."""
    n = text.numel()
    sa64 = sa.view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    cnt = torch.zeros(n, dtype=torch.int32, device=text.device)
    cnt.index_add_(0, sa64, torch.ones(n, dtype=torch.int32, device=text.device))
    if not bool((cnt == 1).all()):
        return False, "not a permutation"
    lcp = sdev.build_lcp(text, sa)
    lcp64 = lcp.to(torch.int64) & 0xFFFFFFFF
    a, b, h = sa64[:-1], sa64[1:], lcp64[1:]
    pa, pb = a + h, b + h
    a_end = pa >= n
    if bool((pb >= n).any()):
        return False, "right suffix exhausted before left one"
    ca = text[torch.clamp(pa, max=n - 1)].to(torch.int32)
    cb = text[pb].to(torch.int32)
    ok = a_end | (ca < cb)
    if not bool(ok.all()):
        return False, "adjacent suffixes out of order"
    rng = np.random.default_rng(seed)
    rs = rng.integers(1, n, size=min(n_samples, n - 1))
    t_host = text.cpu().numpy()
    sa_h = sa64[torch.from_numpy(rs).to(text.device)].cpu().numpy()
    sb_h = sa64[torch.from_numpy(rs - 1).to(text.device)].cpu().numpy()
    l_h = lcp64[torch.from_numpy(rs).to(text.device)].cpu().numpy()
    for x, y, l in zip(sb_h.tolist(), sa_h.tolist(), l_h.tolist()):
        if t_host[x:x + l].tobytes() != t_host[y:y + l].tobytes():
            return False, "LCP overstates a common prefix"
    return True, "permutation + adjacent-order (all pairs) + sampled LCP bytes"


def verify_sa_chunked(torch, sdev, text, sa, chunk=200_000_000, n_samples=2000, seed=1):
    """verify_sa_on_device for inputs whose int64 temporaries would not fit next to the data:
    the permutation test counts in place, and the adjacent-order test walks the suffix array in
    slices, taking each slice's LCP from the engine's per-slice routine (direct comparison,
    src/table.rs:348-361) -- every adjacent pair is still checked.  As in verify_sa_on_device the
    LCP values the order test leans on are re-checked on the host, byte by byte, for n_samples
    pairs of every slice: an overstated LCP would let the next-symbol compare look past the first
    mismatch (an understated one fails the compare itself: the symbols at that offset are equal)."""
    n = text.numel()
    rng = np.random.default_rng(seed)
    t_host = text.cpu().numpy()
    cnt = torch.zeros(n, dtype=torch.int8, device=text.device)
    for lo in range(0, n, chunk):
        idx = sa[lo:lo + chunk].to(torch.int64) & 0xFFFFFFFF
        cnt.index_add_(0, idx, torch.ones(idx.numel(), dtype=torch.int8, device=text.device))
        del idx
    if not bool((cnt == 1).all()):
        return False, "not a permutation"
    del cnt
    prev = None
    for lo in range(0, n, chunk):
        part = sa[lo:lo + chunk]
        lcp = sdev.build_lcp_range(text, part, prev)
        cur = part.to(torch.int64) & 0xFFFFFFFF
        h = lcp.to(torch.int64) & 0xFFFFFFFF
        before = torch.empty_like(cur)
        before[1:] = cur[:-1]
        before[0] = prev if prev is not None else 0
        pa, pb = before + h, cur + h
        skip_first = 1 if prev is None else 0
        pa, pb = pa[skip_first:], pb[skip_first:]
        if bool((pb >= n).any()):
            return False, "right suffix exhausted before left one"
        ca = text[torch.clamp(pa, max=n - 1)].to(torch.int32)
        cb = text[pb].to(torch.int32)
        if not bool(((pa >= n) | (ca < cb)).all()):
            return False, "adjacent suffixes out of order"
        if cur.numel() > skip_first:
            rs = torch.from_numpy(rng.integers(skip_first, cur.numel(), size=min(n_samples, cur.numel() - skip_first))).to(text.device)
            for x, y, l in zip(before[rs].cpu().tolist(), cur[rs].cpu().tolist(), h[rs].cpu().tolist()):
                if x + l > n or y + l > n or t_host[x:x + l].tobytes() != t_host[y:y + l].tobytes():
                    return False, "LCP overstates a common prefix"
        prev = int(cur[-1])
        del lcp, cur, h, before, pa, pb, ca, cb
    return True, "permutation + adjacent-order (all pairs, in slices of %d) + sampled LCP bytes per slice" % chunk


def _sha_u32(t):
    """sha256 of a device int32/uint32 tensor as little-endian u32."""
    import hashlib
    return hashlib.sha256(memoryview(t.cpu().numpy())).hexdigest()


def running_commit():
    """The last commit that touched the engine's sources (suffix_amd/csrc, include): git where there is a checkout, else the
    stamp __graft_entry__.build() leaves next to the library (the GPU boxes get a snapshot without .git)."""
    try:
        import subprocess
        out = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h", "--", "suffix_amd/csrc", "include"], capture_output=True,
                             text=True, timeout=5)
        if out.returncode == 0 and out.stdout.strip():
            return out.stdout.strip()
    except (OSError, ValueError, Exception):
        pass
    try:
        return open(os.path.join(ROOT, "suffix_amd", "_build_commit.txt")).read().strip() or "unknown"
    except OSError:
        return os.environ.get("SFX_COMMIT", "unknown")


# kernels whose cost is one random 128-byte line per key they fetch, not the bytes they stream: (statistics field that counts
# their fetches).  The chip's ceiling for that access shape is the random 4-byte gather probe (sfx_microbench MB_GATHER4,
# 210 GB/s of payload = 52.5 G fetches/s, profiles/r1c_microbench.jsonl; PMC: 128 bytes fetched per read, r5_pmc_fullsize.json)
GATHER_BOUND = {"deep_wave": "deep_gathers", "tile_sort": None}      # (tile_sort: one fetch per member = algorithmic bytes / 17)
GATHER_PROBE_GPS = 52.5


def kernel_rooflines(rep, pmc_kernels=None, min_share=0.15, stats=None):
    """Every kernel that takes at least min_share of the profiled build: algorithmic bytes per launch / mean launch time
    against the HBM peak, and the measured HBM traffic per launch where the committed PMC summary has the kernel.
    Gather-bound kernels also carry their key fetches per second against the random-gather probe."""
    total = sum(r["total_ms"] for r in rep) or 1.0
    out = []
    for r in sorted(rep, key=lambda r: -r["total_ms"]):
        share = r["total_ms"] / total
        if share < min_share or r["launches"] == 0 or r["total_ms"] <= 0:
            continue
        per_b = r["algo_bytes"] / r["launches"]
        per_ms = r["total_ms"] / r["launches"]
        ach = per_b / (per_ms * 1e-3) / 1e9
        ent = {"kernel": r["name"], "share_of_build": round(share, 3), "launches": r["launches"], "avg_launch_ms": round(per_ms, 4),
               "algo_bytes_per_launch": per_b, "achieved": round(ach, 1), "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
               "traffic": None}
        pk = (pmc_kernels or {}).get(r["name"])
        if pk and pk.get("launches") not in (None, r["launches"]):
            # a PMC record of another launch sequence (collected before the code changed): no figure rather than a wrong one
            ent["traffic_note"] = f"PMC record holds {pk['launches']} launches of this kernel, this build made {r['launches']}"
        elif pk:
            # (all symbols of the profile name summed over the build / its launches: scripts/pmc_summary.py)
            ent["traffic"] = round(pk["hbm_bytes_per_launch"])
        if r["name"] in GATHER_BOUND:
            fld = GATHER_BOUND[r["name"]]
            fetches = (stats or {}).get(fld, 0) if fld else r["algo_bytes"] / 17.0
            gps = fetches / (r["total_ms"] * 1e-3) / 1e9
            ent["gather"] = {"key_fetches": int(fetches), "G_fetches/s": round(gps, 1), "probe_G/s": GATHER_PROBE_GPS,
                             "frac_of_probe": round(gps / GATHER_PROBE_GPS, 3)}
        out.append(ent)
    return out


def dominant_kernel(rep, within=0.02):
    """The dominant kernel of a profiled build by accumulated time.  Kernels within `within` of it on time are tied with it,
    and of a tie the one with the LOWER fraction of the HBM peak is the one named (round-4 verdict: the text-fed partition
    pass and the element-fed one took the same 0.39 ms at 0.26 and 0.51 of the peak -- naming the flattering one of a tie
    says nothing).  Returns (record, [{kernel, ms_per_step, frac} of every tied kernel, lowest fraction first])."""
    def frac_of(r):
        ms = r["total_ms"] / max(r["launches"], 1)
        return (r["algo_bytes"] / max(r["launches"], 1)) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else 0.0
    top_ms = max(r["total_ms"] for r in rep)
    tied = sorted([r for r in rep if r["total_ms"] >= (1.0 - within) * top_ms], key=frac_of)
    return tied[0], [{"kernel": r["name"], "ms_per_step": round(r["total_ms"], 4), "frac": round(frac_of(r), 4)} for r in tied]


def load_pmc(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except (OSError, ValueError):
        return {}


def load_pmc_fullsize():
    """The newest committed per-config PMC record (profiles/rN_pmc_fullsize.json, scripts/gpu_pmc_fullsize.sh): every entry is
    recomputable from profiles/rN_pmc_summary_<config>.csv with scripts/pmc_summary.py (tests/test_bench_logic.py does it)."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_fullsize.json")):
        m = re.match(r"r(\d+)_pmc_fullsize\.json$", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), os.path.basename(f))
    return load_pmc(best[1]) if best else {}


def compact_config(key, rec):
    """What the final JSON line keeps of a full-size config record (the record itself is printed on a DETAIL line of its
    own, before the final line: the driver keeps a few KB of stdout tail and the scalar keys of the last line)."""
    if "sa_ms" not in rec:
        return {"key": key, "error": rec.get("error") or rec.get("skipped")}
    rf = rec.get("roofline", {})
    out = {"key": key, "n": rec.get("n"), "sa_ms": rec.get("sa_ms"), "lcp_ms": rec.get("lcp_ms"),
           "fused_ms": rec.get("fused_sa_lcp", {}).get("ms"), "bit_exact": rec.get("bit_exact_vs_pins"),
           "engine_BpB": rf.get("engine", {}).get("algo_bytes_per_input_byte"),
           "engine_frac": rf.get("engine", {}).get("frac_of_hbm_peak"),
           "whole_path_frac": rf.get("whole_path", {}).get("frac_of_hbm_peak")}
    ks = rf.get("kernels") or []
    if ks:
        k = ks[0]
        tr = round(k["traffic"] / k["algo_bytes_per_launch"], 2) if k.get("traffic") and k.get("algo_bytes_per_launch") else None
        out["dominant"] = {"kernel": k["kernel"], "frac": k["frac"], "share": k["share_of_build"], "traffic_ratio": tr}
    if "queries" in rec:
        out["Mqueries/s"] = rec["queries"].get("Mqueries/s")
    return out


def fullsize_config(torch, eng, sdev, _gen, key, size, pins, dev):
    """One BASELINE config at full size on one GPU: device-resident SA (+ LCP, + the
    10^6 positions() queries of config 5), timed with the inputs already in HBM; sha256 of the text, SA,
    LCP and query answers compared with tests/golden/fullsize_pins.json, whose values come from a run
    in which the complete arrays were compared element by element with the oracle
    (profiles/r2_fullsize_full_oracle.jsonl) -- so no 3-minute CPU oracle inside the bench."""
    import hashlib
    import _gen_r1
    spec = {"c3r1": ("config 3 on ROUND 1's input (PCG64 word stream, mean LCP 13.4: the input round 1's 217 ms / 4.61 GB/s was "
                     "measured on): %d B, SA + LCP", _gen_r1.english_like),
            "c5r1": ("config 5 on ROUND 1's input (uniformly drawn code points, mean LCP 6.9: round 1's 131 ms): %d B, SA + LCP",
                     _gen_r1.utf8_mixed),
            "c3": ("config 3: %d B English-like ASCII (SURVEY 8d), SA + LCP", _gen.english_like),
            "c5": ("config 5: %d B UTF-8 mixed-script (SURVEY 8d), SA + LCP + 10^6 positions()", _gen.utf8_mixed),
            "dup": ("high-LCP: %d B near-duplicate documents (16 x 1 MiB, one substitution per ~400 B), SA + LCP",
                    _gen.near_duplicates)}[key]
    t0 = time.perf_counter()
    host = spec[1](size)
    n = int(host.size)
    rec = {"config": spec[0] % n, "n": n, "gen_s": round(time.perf_counter() - t0, 2),
           "sha256_text": hashlib.sha256(memoryview(host)).hexdigest()}
    text = torch.from_numpy(host).to(dev)
    ws = sdev.sa_workspace(n, dev)
    sa = torch.empty(n, dtype=torch.int32, device=dev)
    sdev.build_sa(text, out=sa, workspace=ws)                       # warm-up (first touch of 50 GB of workspace)
    torch.cuda.synchronize()
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        sdev.build_sa(text, out=sa, workspace=ws)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    rec["sa_ms"] = round(best * 1e3, 2)
    rec["sa_MB/s"] = round(n / best / 1e6, 1)
    rec["build"] = eng.build_stats()
    eng.profile(True); eng.profile_reset()
    sdev.build_sa(text, out=sa, workspace=ws); torch.cuda.synchronize()
    prof = sorted(eng.profile_report(), key=lambda r: -r["total_ms"])
    eng.profile(False)
    rec["top_kernels_ms"] = {r["name"]: round(r["total_ms"], 2) for r in prof[:8]}
    engine_algo = sum(r["algo_bytes"] for r in prof)
    pmc_cfg = load_pmc_fullsize().get(key, {})
    del ws
    lws = sdev.lcp_workspace(n, dev)
    lcp = torch.empty(n, dtype=torch.int32, device=dev)
    sdev.build_lcp(text, sa, out=lcp, workspace=lws); torch.cuda.synchronize()
    t0 = time.perf_counter()
    sdev.build_lcp(text, sa, out=lcp, workspace=lws); torch.cuda.synchronize()
    t_lcp = time.perf_counter() - t0
    rec["lcp_ms"] = round(t_lcp * 1e3, 2)
    rec["sa_plus_lcp_MB/s"] = round(n / (best + t_lcp) / 1e6, 1)
    # SuffixTable::new + lcp_lens as ONE engine call: the LCP of every pair that the initial sort or a text
    # round separates is read off the keys inside the build; only the rest is compared on the text
    del lws
    ws2 = sdev.sa_lcp_workspace(n, dev)
    sa2 = torch.empty_like(sa)
    lcp2 = torch.empty_like(lcp)
    sdev.build_sa_lcp(text, out_sa=sa2, out_lcp=lcp2, workspace=ws2); torch.cuda.synchronize()
    t0 = time.perf_counter()
    sdev.build_sa_lcp(text, out_sa=sa2, out_lcp=lcp2, workspace=ws2); torch.cuda.synchronize()
    t_fused = time.perf_counter() - t0
    rec["fused_sa_lcp"] = {"ms": round(t_fused * 1e3, 2), "MB/s": round(n / t_fused / 1e6, 1),
                           "same_arrays_as_separate_calls": bool(torch.equal(sa2, sa) and torch.equal(lcp2, lcp))}
    del ws2, sa2, lcp2
    lws = None
    # W_SA(u32) ~ 69 B per input byte for deep-recursion text, W_LCP = 22
    rec["roofline"] = {"whole_path": {"algo_bytes_per_input_byte": 69.0, "achieved_GB/s": round(69.0 * n / best / 1e9, 1),
                                      "frac_of_hbm_peak": round(69.0 * n / best / 1e9 / HBM_PEAK_GBS, 4)},
                       # what the engine itself moves (sum of its kernels' algorithmic bytes) next to SA-IS's 69
                       "engine": {"algo_bytes_per_input_byte": round(engine_algo / n, 1),
                                  "times_sais": round(engine_algo / n / 69.0, 2),
                                  "achieved_GB/s": round(engine_algo / best / 1e9, 1),
                                  "frac_of_hbm_peak": round(engine_algo / best / 1e9 / HBM_PEAK_GBS, 4)},
                       "kernels": kernel_rooflines(prof, pmc_cfg.get("kernels"), 0.10, rec.get("build")),
                       "traffic_commit": pmc_cfg.get("commit"),
                       "lcp": {"algo_bytes_per_input_byte": 22.0, "achieved_GB/s": round(22.0 * n / t_lcp / 1e9, 1)}}
    rec["sha256_sa"] = _sha_u32(sa)
    rec["sha256_lcp"] = _sha_u32(lcp)
    del lcp, lws
    if key == "c5":
        qb, off = _gen.queries(host, 1_000_000)
        d_qb, d_off = torch.from_numpy(qb).to(dev), torch.from_numpy(off).to(dev)
        # (both forms: one untimed call, then the mean of 5 batches of 10^6 queries issued back to back)
        q_reps = 5
        sdev.query_batch(text, sa, d_qb, d_off); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(q_reps):
            s0, e0, f0, a0 = sdev.query_batch(text, sa, d_qb, d_off)
        torch.cuda.synchronize()
        t_plain = (time.perf_counter() - t0) / q_reps
        # the resident index: text + SA stay in HBM, plus a B+tree over the first 16 bytes of every suffix and the
        # bucket directory (the tree's entry point, and the fallback structure)
        t0 = time.perf_counter()
        ix = sdev.DeviceIndex(text, sa); torch.cuda.synchronize()
        t_ix = time.perf_counter() - t0
        ix.query(d_qb, d_off); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(q_reps):
            s, e, f, a = ix.query(d_qb, d_off)
        torch.cuda.synchronize()
        t_q = (time.perf_counter() - t0) / q_reps
        nbytes = int(off[-1])
        rec["queries"] = {"count": 1_000_000, "batches_timed": q_reps, "ms": round(t_q * 1e3, 3), "Mqueries/s": round(1.0 / t_q, 1),
                          "undirected_binary_search": {"ms": round(t_plain * 1e3, 3), "Mqueries/s": round(1.0 / t_plain, 1)},
                          "index_build_ms": round(t_ix * 1e3, 2),
                          "same_answers_as_undirected": bool(torch.equal(s, s0) and torch.equal(e, e0) and torch.equal(f, f0)),
                          "hit_fraction": round(float(f.float().mean()), 4), "mean_query_bytes": round(nbytes / 1e6, 1),
                          # 2 * ceil(log2 n) probes * (4 B SA entry + ~8 compared bytes) = 720 B per query
                          "roofline": {"algo_bytes_per_query": 720, "achieved_GB/s": round(720 * 1e6 / t_q / 1e9, 1),
                                       "note": "SURVEY's per-query figure for the undirected search.  What bounds a query is random 128-byte "
                                               "line fetches: the index answers queries of <= 16 bytes (and misses) from the nodes of a B+tree "
                                               "over 16-byte prefix keys, entered below the bucket directory; longer queries bisect the ranks "
                                               "that share their first 16 bytes on the text, in a second launch of their own (DESIGN.md 6)"},
                          "sha256_start_end": hashlib.sha256(memoryview(torch.stack([s, e]).cpu().numpy())).hexdigest()}
        ix.close()
    pin = (pins or {}).get(key, {}).get(str(n))
    if pin:
        checks = {k: rec.get(k) == pin.get(k) for k in ("sha256_text", "sha256_sa", "sha256_lcp") if pin.get(k)}
        if "queries" in rec and pin.get("sha256_start_end"):
            checks["sha256_start_end"] = rec["queries"]["sha256_start_end"] == pin["sha256_start_end"]
        rec["bit_exact_vs_pins"] = bool(checks) and all(checks.values())
        rec["pin_checks"] = checks
        rec["pin_source"] = pin.get("source")
    else:
        # no pin for this size: fall back to the size-independent property gate
        ok, how = verify_sa_on_device(torch, sdev, text, sa)
        rec["bit_exact_vs_pins"] = None
        rec["verified_by_properties"] = bool(ok)
        rec["verification"] = how
    del text, sa
    torch.cuda.empty_cache()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=100_000_000, help="bytes of text per GPU")
    ap.add_argument("--input", choices=("dna", "periodic"), default="dna",
                    help="dna: BASELINE config 2's uniform DNA (the bench line); periodic: shards of one periodic text -- a "
                         "partitioned build (N > 1) must take the replicated fallback on it (tests only, never the bench line)")
    ap.add_argument("--ragged", action="store_true",
                    help="N > 1, tests only: rank r's shard is 4097 r + 1 bytes shorter (no multiple of the symbols per packed "
                         "word: the packed exchange assembles the words that straddle two shards)")
    ap.add_argument("--cpu-sample", type=int, default=100_000_000,
                    help="bytes of the same text the CPU baseline is timed on (0 = skip)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--calibrate", action="store_true",
                    help="also launch the known-byte-count copy / run-scatter micro-benchmarks once, so a "
                         "rocprofv3 --pmc pass of this command can calibrate FETCH_SIZE / WRITE_SIZE")
    ap.add_argument("--no-microbench", action="store_true", help="skip the scatter/gather roofline probes")
    ap.add_argument("--configs", default="c3,c5,dup,c3r1,c5r1",
                    help="full-size BASELINE configs reported next to the headline at N = 1 (c3 = 1 GB English-like "
                         "SA + LCP, c5 = 1 GB UTF-8 + 10^6 positions(), dup = 1 GB near-duplicate documents, c3r1 / c5r1 = "
                         "configs 3 / 5 on round 1's easier inputs, for a like-for-like comparison with round 1 -- their "
                         "numpy generators take ~1 min each, so they are skipped once the run is past --config-budget seconds); "
                         "'' = none")
    ap.add_argument("--total-size", type=int, default=0,
                    help="STRONG scaling: the bytes of ONE text (uniform DNA, seed 0x5AF1C5 + 4: BASELINE config 4 at 4000000000) cut into "
                         "N equal shards, rank r generating bytes [r T / N, (r + 1) T / N) -- total work fixed as N grows; --size is "
                         "ignored.  The line then says \"scaling\": \"strong\" and names the config in config.workload")
    ap.add_argument("--dev-lib", default=None,
                    help="development A/B runs only (scripts/gpu_ab.sh): bind this build of the C ABI (libsuffix_hip_dev.so, hooks "
                         "compiled in) instead of the product library; the line then carries config.dev_lib")
    ap.add_argument("--config-budget", type=float, default=150.0)
    ap.add_argument("--config-size", type=int, default=1_000_000_000)
    args = ap.parse_args()

    # multi-process GPU work on this host driver needs dmabuf IPC (RCCL fails with
    # hipIpcGetMemHandle: invalid argument otherwise); exported already, kept if it is not
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    import _gen
    import suffix_amd
    from suffix_amd import device as sdev
    from suffix_amd import dist as sdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if args.dev_lib:
        from suffix_amd import _lib as slib
        slib.set_default_engine(suffix_amd.Engine(lib_path=os.path.abspath(args.dev_lib)))
    eng = suffix_amd.default_engine()
    eng.require_device()                       # no CPU fallback: fail loudly without a GPU
    # test hook for 1-GPU boxes: SFX_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and swaps RCCL
    # (which refuses two ranks on one device) for gloo, so the N>1 code path can be rehearsed
    share_gpu = os.environ.get("SFX_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # second test hook: SFX_BENCH_FORCE_PARTITIONED=1 times the partitioned build at world = 1 (its
    # collectives degenerate), i.e. the host-side cost of the multi-GPU path on top of its kernels
    force_part = os.environ.get("SFX_BENCH_FORCE_PARTITIONED") == "1" and world == 1
    if force_part:
        class _Done:
            def wait(self):
                return True

        class _OneRank:                                  # the collectives of a 1-rank world, on the device
            ReduceOp = dist.ReduceOp

            @staticmethod
            def get_world_size(group=None):
                return 1

            @staticmethod
            def get_rank(group=None):
                return 0

            @staticmethod
            def all_reduce(t, op=None, group=None):
                return None

            @staticmethod
            def all_gather_into_tensor(dst, src, group=None, async_op=False):
                dst.copy_(src)
                return _Done() if async_op else None

            @staticmethod
            def all_gather(lst, src, group=None, async_op=False):
                lst[0].copy_(src)
                return _Done() if async_op else None
        sdist.dist = _OneRank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    n_local = args.size - ((4097 * rank + 1) if (args.ragged and world > 1) else 0)
    seed = 0x5AF1C5 + 1 + rank                 # seed = 0x5AF1C5 + config index
    strong = args.total_size > 0
    shard_begin = 0
    if strong:
        # ONE text for every N: rank r holds bytes [r T / N, (r + 1) T / N) of the stream of seed 0x5AF1C5 + 4 (config 4's)
        if args.total_size > 0xFFFFFFFF:
            raise SystemExit("--total-size beyond u32::MAX bytes (src/table.rs:380)")
        shard_begin = args.total_size * rank // world
        n_local = args.total_size * (rank + 1) // world - shard_begin
    if strong:
        host_text = _gen.dna_slice(shard_begin, n_local, seed=0x5AF1C5 + 4)
    elif args.input == "periodic":
        # every shard a stretch of ONE periodic text (period 24, a different phase per rank): suffixes tie for the whole
        # length of the text, the range build reports SFX_ERR_NEEDS_RANKS and every rank builds the whole array
        unit = np.frombuffer(b"ACGTTGCAACGGTTCAGTCATGCA", dtype=np.uint8)
        host_text = np.ascontiguousarray(np.resize(np.roll(unit, -((args.size * rank) % unit.size)), n_local))
    else:
        host_text = _gen.dna(n_local, seed=seed)
    text = torch.from_numpy(host_text).to(dev)
    n_total = sum(args.size - ((4097 * r + 1) if (args.ragged and world > 1) else 0) for r in range(world))
    if strong:
        n_total = args.total_size

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world == 1:
        ws = sdev.sa_workspace(n_local, dev)
        sa = torch.empty(n_local, dtype=torch.int32, device=dev)

        def step():
            sdev.build_sa(text, out=sa, workspace=ws)
        if force_part:
            def step():                                        # noqa: F811  (test hook, see above)
                sa.copy_(sdist.build_sa_partitioned(text)[0])
    else:
        result = {}

        def step():
            result["part"] = sdist.build_sa_partitioned(text)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = elapsed * 1e3 / max(args.steps, 1)
    value = n_total * args.steps / elapsed / 1e6
    stats = eng.build_stats()
    phases = None
    if world > 1:
        # one more, untimed, build with a device synchronisation at every phase boundary: where the
        # partitioned path spends its time on this rank (max over ranks per phase)
        ph = {}
        sdist.build_sa_partitioned(text, timings=ph)
        names = ["byte_hist", "all_gather_issue", "key_hist", "all_gather_wait", "plan", "range_build"]
        tv = torch.tensor([ph.get(k, 0.0) for k in names], dtype=torch.float64, device=dev)
        dist.all_reduce(tv, op=dist.ReduceOp.MAX)
        phases = {k: round(float(v), 3) for k, v in zip(names, tv.tolist())}
        if "fallback" in ph.get("info", {}):
            phases["fallback"] = ph["info"]["fallback"]
        if "text_exchange" in ph.get("info", {}):
            phases["text_exchange"] = ph["info"]["text_exchange"]

    # ---- lcp_lens on the same text (reported next to the headline, never part of `value`:
    # SuffixTable::new builds the suffix array only, src/table.rs:79-91; the LCP array is a
    # separate call, :130-138) ----
    lcp_info = None
    e2e = None
    if world == 1 and not strong:                # (the strong-scaling series times the build alone, at every N)
        lcp_ws = sdev.lcp_workspace(n_local, dev)
        lcp = torch.empty(n_local, dtype=torch.int32, device=dev)
        sdev.build_lcp(text, sa, out=lcp, workspace=lcp_ws)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sdev.build_lcp(text, sa, out=lcp, workspace=lcp_ws)
        barrier()
        lcp_ms = (time.perf_counter() - t0) * 1e3 / max(args.steps, 1)
        lcp_info = {"ms_per_step": round(lcp_ms, 3), "MB/s": round(n_local / lcp_ms / 1e3, 1),
                    "sa_plus_lcp_MB/s": round(n_local / (lcp_ms + ms_per_step) / 1e3, 1),
                    "note": "lcp_lens (src/table.rs:130-138) on the device-resident text and SA; not part of `value`"}
        del lcp_ws
        # SuffixTable::new + lcp_lens as ONE engine call (sfx_build_sa_lcp_u32_dev): the LCP of every pair the
        # initial key sort tells apart is read off the sorted keys inside the build
        ws2 = sdev.sa_lcp_workspace(n_local, dev)
        sa2 = torch.empty_like(sa)
        lcp2 = torch.empty_like(lcp)
        sdev.build_sa_lcp(text, out_sa=sa2, out_lcp=lcp2, workspace=ws2)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sdev.build_sa_lcp(text, out_sa=sa2, out_lcp=lcp2, workspace=ws2)
        barrier()
        fused_ms = (time.perf_counter() - t0) * 1e3 / max(args.steps, 1)
        lcp_info["fused_sa_lcp"] = {"ms_per_step": round(fused_ms, 3), "MB/s": round(n_local / fused_ms / 1e3, 1),
                                    "same_arrays_as_separate_calls": bool(torch.equal(sa2, sa) and torch.equal(lcp2, lcp))}
        del ws2, sa2, lcp2

    # ---- end to end: what the Rust shim calls (SuffixTable::new, src/table.rs:78-85 -> sais_table :378-386): pageable &str
    # in, pageable Vec<u32> out through sfx_build_sa_u32 -- H2D of n bytes, the build, D2H of 4 n bytes, staging and workspace
    # from the library's pool.  PCIe-inclusive, reported beside `value`, never in it ("also report end-to-end").
    if world == 1 and rank == 0:
        sa_host = np.empty(n_local, dtype=np.uint32)
        times = []
        for _ in range(4):                               # (the first call fills the pool: not counted)
            t0 = time.perf_counter()
            eng.check(eng.lib.sfx_build_sa_u32(host_text.ctypes.data, n_local, sa_host.ctypes.data), "sfx_build_sa_u32")
            times.append((time.perf_counter() - t0) * 1e3)
        best, mean = min(times[1:]), sum(times[1:]) / len(times[1:])
        link_gbs = 63.0                                  # PCIe Gen5 x16, one direction
        floor_ms = 5.0 * n_local / (link_gbs * 1e9) * 1e3
        e2e = {"entry": "sfx_build_sa_u32 (host pointers: pageable text in, pageable u32 array out)",
               "ms": round(best, 3), "mean_ms": round(mean, 3), "MB/s": round(n_local / best / 1e3, 1),
               "pcie_floor_ms": round(floor_ms, 2), "pcie_floor_note": "5 n bytes over 63 GB/s; nothing overlaps: the alphabet needs "
               "the whole text, the array is final only when the build ends (+ the build's own ms_per_step)",
               "same_array_as_device_build": bool(np.array_equal(sa_host.view(np.int32), sa.cpu().numpy()))}
        del sa_host

    # ---- per-kernel roofline: HIP events on the launch stream, separate untimed build ----
    eng.profile(True)
    eng.profile_reset()
    step()
    torch.cuda.synchronize()
    rep = eng.profile_report()
    eng.profile(False)
    kernels = {r["name"]: r for r in rep}
    total_ms = sum(r["total_ms"] for r in rep) or 1.0
    dom, kernels_tied = dominant_kernel(rep)
    per_launch_bytes = dom["algo_bytes"] / max(dom["launches"], 1)
    per_launch_ms = dom["total_ms"] / max(dom["launches"], 1)
    achieved = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
    roofline = {
        "bound": "hbm", "kernel": dom["name"], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
        "launches_per_step": dom["launches"], "avg_launch_ms": round(per_launch_ms, 4),
        "algo_bytes_per_launch": per_launch_bytes,
        "share_of_step": round(dom["total_ms"] / total_ms, 3),
        "kernels_tied": kernels_tied,
        "kernel_ms": {k: round(v["total_ms"], 3) for k, v in sorted(kernels.items())},
        # every kernel with at least 15 % of the step, each against the HBM peak (traffic: filled in below)
        "kernels": kernel_rooflines(rep, None, 0.15),
        "engine_algo_bytes_per_input_byte": round(sum(r["algo_bytes"] for r in rep) / max(n_local, 1), 1),
        # whole path at SURVEY.md 8d's figure (65 algorithmic B per input byte, u32 DNA)
        "whole_path": {"algo_bytes_per_input_byte": 65.0,
                       "achieved": round(65.0 * value / 1e3, 1), "unit": "GB/s",
                       "frac": round(65.0 * value / 1e3 / HBM_PEAK_GBS, 4)},
    }

    # ---- HBM traffic of the dominant kernel: PMC counters need their own rocprofv3 passes
    # (MI355X_MICROARCH.md, HBM section), so the per-launch figure comes from the committed
    # summary of those passes over this same command (scripts/gpu_pmc.sh -> profiles/) ----
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path))
            for kr in roofline["kernels"]:
                pk = pmc.get("kernels", {}).get(kr["kernel"])
                if pk:
                    kr["traffic"] = round(pk["hbm_bytes_per_launch"])
            ent = pmc.get("kernels", {}).get(dom["name"])
            if ent:
                roofline["traffic"] = round(ent["hbm_bytes_per_launch"])
                roofline["traffic_detail"] = {k: ent[k] for k in ("fetch_bytes", "write_bytes", "launches")}
                roofline["traffic_source"] = pmc.get("source", "profiles/pmc_latest.json")
                # the counters come from their own rocprofv3 passes (they cannot be collected inside this run):
                # the commit they were collected at, next to the commit that is running, if it can be told
                roofline["traffic_commit"] = pmc.get("commit", "unknown")
                roofline["this_commit"] = running_commit()
        except (ValueError, KeyError, OSError):
            pass

    # ---- what the memory system sustains for the engine's access shapes ----
    if rank == 0 and world == 1 and not args.no_microbench:
        E = eng
        roofline["scatter_bw"] = {
            "unit": "GB/s",
            "copy_16B": round(E.microbench(E.MB_COPY, 1 << 30), 1),
            "random_4B_scatter_into_4n": round(E.microbench(E.MB_SCATTER4, 4 * n_local), 1),
            "random_4B_gather_from_4n": round(E.microbench(E.MB_GATHER4, 4 * n_local), 1),
            "random_1B_gather_from_n": round(E.microbench(E.MB_GATHER1, n_local), 1),
            "runs_128B_unaligned_copy": round(E.microbench(E.MB_RUNSCATTER, 8 * n_local, 128, 0), 1),
            "runs_128B_aligned_copy": round(E.microbench(E.MB_RUNSCATTER, 8 * n_local, 128, 1), 1),
        }
    if args.calibrate and world == 1:
        eng.microbench(eng.MB_COPY, 1 << 30, 0, 0, 1)
        eng.microbench(eng.MB_RUNSCATTER, 8 * n_local, 128, 0, 1)

    # ---- correctness gates ----
    verified, how = None, "skipped"
    if not args.no_verify:
        if world == 1 and n_local > (1 << 30):
            verified, how = verify_sa_chunked(torch, sdev, text, sa)     # (the int64 temporaries of the plain gate would not fit)
        elif world == 1:
            verified, how = verify_sa_on_device(torch, sdev, text, sa)
        else:
            part, offset, n_all = result["part"]
            tot = torch.tensor([part.numel()], dtype=torch.int64, device=dev)
            dist.all_reduce(tot)
            verified, how = bool(int(tot.item()) == n_all), "slice sizes sum to n"
            if verified:
                try:
                    verified, how = sdist.verify_partitioned(text, part, n_all)
                except Exception as exc:       # the gate must never cost the run its result line
                    how += f"; full gate failed to run: {type(exc).__name__}: {exc}"

    # ---- CPU baseline: the oracle (C restatement of the reference's sais), 1 thread ----
    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample > 0 and not strong:
        import oracle
        m = min(args.cpu_sample, n_local)
        sample = host_text[:m]
        # one thread (the reference is single-threaded), pinned to one core, best of 3
        try:
            old_aff = os.sched_getaffinity(0)
            os.sched_setaffinity(0, {sorted(old_aff)[len(old_aff) // 2]})
        except (AttributeError, OSError):
            old_aff = None
        cpu_s = None
        for _ in range(3):
            tc = time.perf_counter()
            exp = oracle.sais(sample)
            dt = time.perf_counter() - tc
            cpu_s = dt if cpu_s is None else min(cpu_s, dt)
        if old_aff is not None:
            os.sched_setaffinity(0, old_aff)
        cpu_model = ""
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        sub = torch.from_numpy(np.ascontiguousarray(sample)).to(dev)
        d_sub_sa = sdev.build_sa(sub)
        got = d_sub_sa.cpu().numpy().view(np.uint32)
        torch.cuda.synchronize()
        same = bool(np.array_equal(got, exp))
        if same and lcp_info is not None:
            tc = time.perf_counter()
            exp_lcp = oracle.lcp_kasai(sample, exp)
            lcp_cpu_s = time.perf_counter() - tc
            got_lcp = sdev.build_lcp(sub, d_sub_sa).cpu().numpy().view(np.uint32)
            lcp_info["bit_exact_vs_oracle"] = bool(np.array_equal(got_lcp, exp_lcp))
            lcp_info["cpu_kasai_MB/s"] = round(m / lcp_cpu_s / 1e6, 2)
            same = same and lcp_info["bit_exact_vs_oracle"]
            del exp_lcp, got_lcp
        verified = bool(verified) and same if verified is not None else same
        how += ("; complete SA bit-exact vs oracle" if m == n_local else "; SA of the CPU sample bit-exact vs oracle") \
            if same else "; MISMATCH vs oracle on CPU sample"
        if lcp_info is not None and lcp_info.get("bit_exact_vs_oracle"):
            how += "; LCP of the same bytes bit-exact vs oracle (Kasai)"
        cpu = {"value": round(m / cpu_s / 1e6, 3), "unit": "MB/s", "cores": 1, "kind": "port",
               "sample": f"{'all' if m == n_local else 'first'} {m} bytes of the same DNA text, oracle.sais (C restatement of "
                         f"src/table.rs:388-574, gcc -O3 -march=native), pinned to one core, best of 3: {cpu_s:.1f} s",
               "host_cpus": os.cpu_count(), "cpu_model": cpu_model}

    configs = None
    if rank == 0 and world == 1 and args.configs and not strong:
        del text, sa
        torch.cuda.empty_cache()
        pins = {}
        try:
            pins = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_pins.json")))
        except (OSError, ValueError):
            pass
        configs = []
        t_cfg0 = time.perf_counter()
        for key in [k for k in args.configs.split(",") if k]:
            if key.endswith("r1") and time.perf_counter() - t_cfg0 > args.config_budget:
                configs.append({"config": key, "skipped": "time budget of the default run (--config-budget)"})
                continue
            try:
                configs.append(fullsize_config(torch, eng, sdev, _gen, key, args.config_size, pins, dev))
            except Exception as exc:                    # a failing extra must not cost the headline its line
                configs.append({"config": key, "error": f"{type(exc).__name__}: {exc}"})

    if rank == 0:
        # DETAIL lines first (one JSON object each, never the last line): everything round 3 packed into one 25 KB line
        workload = (f"{n_local} B synthetic DNA (sigma=4, uniform, splitmix64) per GPU, u32 indices, device-resident text -> device SA"
                    + ("" if world == 1 else f"; range-partitioned over {world} GPUs, text {n_total} B"))
        if strong:
            workload = (f"{n_total} B synthetic DNA (sigma=4, uniform, splitmix64 seed 0x5AF1C5 + 4"
                        + (": BASELINE config 4's text" if n_total == 4_000_000_000 else "")
                        + f"), ONE text cut into {world} shard(s) of {n_total // world} B, u32 indices, device-resident shards -> "
                        + ("device SA" if world == 1 else f"one contiguous slice of the SA per GPU (range-partitioned, suffix_amd/dist.py)"))
        print("DETAIL " + json.dumps({"detail": "headline", "config": {"workload": workload, "text_bytes_total": n_total, "build": stats,
                                                                        "partitioned_phases_ms": phases},
                                      "roofline": roofline, "cpu_baseline": cpu, "lcp": lcp_info, "verification": how}))
        cfg_keys = [k for k in args.configs.split(",") if k] if configs is not None else []
        for key, rec in zip(cfg_keys, configs or []):
            print("DETAIL " + json.dumps({"detail": key, **rec}))
        engine_bpb = roofline["engine_algo_bytes_per_input_byte"]
        short_roof = {"bound": "hbm", "kernel": roofline["kernel"], "achieved": roofline["achieved"], "peak": HBM_PEAK_GBS,
                      "unit": "GB/s", "frac": roofline["frac"], "traffic": roofline.get("traffic"),
                      "avg_launch_ms": roofline["avg_launch_ms"], "algo_bytes_per_launch": roofline["algo_bytes_per_launch"],
                      "share_of_step": roofline["share_of_step"], "kernels_tied": roofline["kernels_tied"],
                      "traffic_commit": roofline.get("traffic_commit"), "this_commit": roofline.get("this_commit"),
                      # the engine's own algorithmic bytes (what its kernels declare) per input byte, and that against the HBM peak
                      "engine_BpB": engine_bpb, "engine_frac": round(engine_bpb * value / 1e3 / HBM_PEAK_GBS, 4),
                      # SURVEY 8d's SA-IS figure (65 B per input byte) against the same time
                      "whole_path_frac": roofline["whole_path"]["frac"],
                      "kernels": [{"kernel": k["kernel"], "frac": k["frac"], "share": k["share_of_build"],
                                   "traffic_ratio": (round(k["traffic"] / k["algo_bytes_per_launch"], 2) if k.get("traffic") else None)}
                                  for k in roofline["kernels"]],
                      "configs": [compact_config(k, r) for k, r in zip(cfg_keys, configs or [])]}
        short_cpu = None
        if cpu:
            short_cpu = {"value": cpu["value"], "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"],
                         "sample": cpu["sample"][:160], "cpu_model": cpu.get("cpu_model")}
        short_lcp = None
        if lcp_info:
            short_lcp = {"ms": lcp_info["ms_per_step"], "fused_sa_lcp_ms": lcp_info.get("fused_sa_lcp", {}).get("ms_per_step"),
                         "bit_exact_vs_oracle": lcp_info.get("bit_exact_vs_oracle")}
        out = {
            "metric": METRIC, "value": round(value, 2), "unit": "MB/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": workload, "text_bytes_total": n_total, "partitioned_phases_ms": phases,
                       **({"dev_lib": os.path.basename(args.dev_lib)} if args.dev_lib else {})},
            "roofline": short_roof, "cpu_baseline": short_cpu, "lcp": short_lcp,
            # the metric's own words, "SuffixTable::new (SA-IS+LCP)": `value` is new() alone (config 2 is SA-only, and new() builds
            # no LCP array, src/table.rs:78-85); new() + lcp_lens() as ONE engine call, both arrays device-resident, is here
            "value_sa_plus_lcp": ({"ms": lcp_info["fused_sa_lcp"]["ms_per_step"], "MB/s": lcp_info["fused_sa_lcp"]["MB/s"],
                                   "entry": "sfx_build_sa_lcp_u32_dev",
                                   "same_arrays_as_separate_calls": lcp_info["fused_sa_lcp"]["same_arrays_as_separate_calls"]}
                                  if lcp_info and lcp_info.get("fused_sa_lcp") else None),
            "end_to_end": e2e,
            "verified": verified, "verification": how[:200],
        }
        line = json.dumps(out)
        if len(line) > 4000:                            # the driver keeps ~7 KB of tail: never let the last line outgrow it
            out["roofline"]["kernels"] = out["roofline"]["kernels"][:2]
            out["verification"] = how[:80]
            line = json.dumps(out)
        print(line)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
