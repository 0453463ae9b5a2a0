#!/usr/bin/env python3
"""Development timing of one full-size build: python scripts/gpu_time_build.py <eng|utf8|dup|dna|engr1|utf8r1> [n]
Prints one JSON line: sa_ms (best of 2), build stats, per-kernel ms, sha256 of the SA (compare across variants)."""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import _gen, suffix_amd
import _devlib
from suffix_amd import device as sdev
kind = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000_000
if kind in ("engr1", "utf8r1"):
    import _gen_r1
    host = (_gen_r1.english_like if kind == "engr1" else _gen_r1.utf8_mixed)(n)
else:
    host = {"eng": _gen.english_like, "utf8": _gen.utf8_mixed, "dup": _gen.near_duplicates,
            "dna": lambda k: _gen.dna_fast(k, seed=0x5AF1C5 + 4)}[kind](n)
eng = _devlib.engine(); eng.require_device()
dev = torch.device("cuda", 0)
text = torch.from_numpy(host).to(dev)
ws = sdev.sa_workspace(n, dev); sa = torch.empty(n, dtype=torch.int32, device=dev)
best = None
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sdev.build_sa(text, out=sa, workspace=ws); torch.cuda.synchronize()
    dt = time.perf_counter() - t0; best = dt if best is None else min(best, dt)
st = eng.build_stats()
eng.profile(True); eng.profile_reset(); sdev.build_sa(text, out=sa, workspace=ws); torch.cuda.synchronize()
k = {r["name"]: round(r["total_ms"], 2) for r in eng.profile_report()}; eng.profile(False)
rec = {"kind": kind, "n": n, "env": {a: b for a, b in os.environ.items() if a.startswith("SFX_")}, "sa_ms": round(best * 1e3, 2),
       "stats": st, "kernel_ms": dict(sorted(k.items(), key=lambda x: -x[1])[:12])}
if os.environ.get("TIME_FUSED") == "1":
    # SuffixTable::new + lcp_lens as one engine call: LCP of the pairs split by the initial sort / the text rounds
    # comes out of the build itself
    del ws
    ws2 = sdev.sa_lcp_workspace(n, dev); lcp = torch.empty(n, dtype=torch.int32, device=dev)
    bf = None
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sdev.build_sa_lcp(text, out_sa=sa, out_lcp=lcp, workspace=ws2); torch.cuda.synchronize()
        dt = time.perf_counter() - t0; bf = dt if bf is None else min(bf, dt)
    eng.profile(True); eng.profile_reset(); sdev.build_sa_lcp(text, out_sa=sa, out_lcp=lcp, workspace=ws2); torch.cuda.synchronize()
    kk = {r["name"]: round(r["total_ms"], 2) for r in eng.profile_report()}; eng.profile(False)
    rec["fused_sa_lcp_ms"] = round(bf * 1e3, 2)
    rec["fused_lcp_kernels_ms"] = {a: b for a, b in kk.items() if a.startswith("lcp") or a in ("plcp", "phi_scatter", "phi_pairs")}
    t0 = time.perf_counter(); l2 = sdev.build_lcp(text, sa); torch.cuda.synchronize()
    rec["separate_lcp_ms_cold"] = round((time.perf_counter() - t0) * 1e3, 2)
    rec["fused_lcp_equals_separate"] = bool(torch.equal(l2, lcp))
    rec["sha256_lcp"] = hashlib.sha256(lcp.cpu().numpy().tobytes()).hexdigest()[:16]
    del l2, lcp, ws2
if os.environ.get("PMC_CALIBRATE") == "1":
    # known byte counts for the counter calibrations of scripts/gpu_pmc_fullsize.sh: a 1 GiB copy, 2^28 random 4-byte reads
    ws = None
    eng.microbench(eng.MB_COPY, 1 << 30, 0, 0, 1)
    eng.microbench(eng.MB_GATHER4, 1 << 30, 0, 0, 1)
if os.environ.get("TIME_SHA", "1") == "1":
    rec["sha256_sa"] = hashlib.sha256(sa.cpu().numpy().tobytes()).hexdigest()[:16]
print(json.dumps(rec), flush=True)
