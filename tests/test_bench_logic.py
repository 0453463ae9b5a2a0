"""bench.py's record-keeping logic on canned profile reports (CPU, no device): which kernel the final line names as dominant
(of kernels tied on time the one with the LOWER fraction of the peak), what kernel_rooflines() keeps, what compact_config()
makes of a full-size record."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _k(name, ms, launches, gbytes):
    return {"name": name, "total_ms": ms, "launches": launches, "algo_bytes": gbytes * 1e9}


def test_dominant_kernel_names_the_lower_fraction_of_a_tie():
    # round 4's headline: the two partition passes tie on time, at 0.26 and 0.51 of the peak
    rep = [_k("radix_scatter_u32", 0.3912, 1, 1.6), _k("radix_scatter_text_u32", 0.3974, 1, 0.825), _k("bucket_sort_lds", 0.354, 1, 1.6)]
    dom, tied = bench.dominant_kernel(rep)
    assert dom["name"] == "radix_scatter_text_u32"
    assert [t["kernel"] for t in tied] == ["radix_scatter_text_u32", "radix_scatter_u32"]
    assert abs(tied[0]["frac"] - 0.2595) < 1e-3 and abs(tied[1]["frac"] - 0.5112) < 1e-3
    # the flattering kernel ahead by less than 2 %: still the lower fraction
    rep[0]["total_ms"], rep[1]["total_ms"] = 0.3974, 0.3912
    assert bench.dominant_kernel(rep)[0]["name"] == "radix_scatter_text_u32"


def test_dominant_kernel_without_a_tie():
    # round 5's headline: the element-fed pass leads by 5 %
    rep = [_k("radix_scatter_u32", 0.376, 1, 1.6), _k("radix_scatter_text_u32", 0.357, 1, 0.825), _k("bucket_sort_lds", 0.363, 1, 1.6)]
    dom, tied = bench.dominant_kernel(rep)
    assert dom["name"] == "radix_scatter_u32" and len(tied) == 1
    assert abs(tied[0]["frac"] - 1.6e9 / 0.376e-3 / 1e9 / bench.HBM_PEAK_GBS) < 1e-3
    # launches are averaged: eight launches of 5.6 ms moving 24 GB each
    dom, _ = bench.dominant_kernel([_k("radix_scatter_u64", 44.8, 8, 8 * 24.0), _k("deep_wave", 17.3, 7, 7 * 0.88)])
    assert dom["name"] == "radix_scatter_u64"


def test_kernel_rooflines_share_and_traffic():
    rep = [_k("radix_scatter_u64", 44.8, 8, 8 * 23.5), _k("deep_wave", 17.3, 7, 7 * 0.88), _k("groups_reduce", 1.6, 1, 8.0)]
    out = bench.kernel_rooflines(rep, {"radix_scatter_u64": {"hbm_bytes_per_launch": 24.7e9}}, 0.15, stats={"deep_gathers": 536_000_000})
    assert [k["kernel"] for k in out] == ["radix_scatter_u64", "deep_wave"]           # groups_reduce is below 15 % of the build
    assert out[0]["traffic"] == 24_700_000_000 and out[1]["traffic"] is None
    assert abs(out[0]["frac"] - 23.5e9 / 5.6e-3 / 1e9 / bench.HBM_PEAK_GBS) < 1e-3
    assert "gather" in out[1] and out[1]["gather"]["key_fetches"] == 536_000_000


def test_compact_config_of_an_error_record():
    assert bench.compact_config("c3", {"config": "c3", "error": "RuntimeError: x"}) == {"key": "c3", "error": "RuntimeError: x"}
    assert bench.compact_config("c5r1", {"config": "c5r1", "skipped": "time budget"})["error"] == "time budget"


# ---- PMC records: every per-config roofline must be reproducible from what profiles/ holds -------------
import csv      # noqa: E402
import glob     # noqa: E402
import json     # noqa: E402
import re       # noqa: E402
import subprocess  # noqa: E402

import pytest  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "scripts"))
import pmc_summary  # noqa: E402


def test_kernel_rooflines_refuse_a_pmc_record_of_another_launch_sequence():
    rep = [_k("radix_scatter_u64", 44.8, 8, 8 * 23.5)]
    out = bench.kernel_rooflines(rep, {"radix_scatter_u64": {"hbm_bytes_per_launch": 21.45e9, "launches": 1}}, 0.15)
    assert out[0]["traffic"] is None and "1 launches" in out[0]["traffic_note"]           # round 5's mis-filed record
    out = bench.kernel_rooflines(rep, {"radix_scatter_u64": {"hbm_bytes_per_launch": 25.6e9, "launches": 8}}, 0.15)
    assert out[0]["traffic"] == 25_600_000_000


def test_every_engine_kernel_has_a_profile_name():
    """rocprofv3 names kernel symbols, the engine's profiler names launches: scripts/pmc_summary.py must know every
    `__global__` kernel of the engine, or a PMC summary silently drops it (round 5: k_radix_sweep_duo, 46 % of config 3)."""
    csrc = os.path.join(ROOT, "suffix_amd", "csrc")
    kernels = set()
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".hpp")):
            src = open(os.path.join(csrc, f)).read()
            for m in re.finditer(r"__global__[^;{]*?\b(k_[a-z0-9_]+)\s*\(", src, flags=re.S):
                kernels.add(m.group(1))
    assert len(kernels) > 80, len(kernels)
    syms = [sym for sym, _ in pmc_summary.NAMES]
    # (a template kernel is known when an entry names one of its instances; a plain one when an entry is a prefix of its name)
    missing = sorted(k for k in kernels
                     if not any(sym.startswith(k + "<") or k.startswith(sym) or sym == "detail::" + k for sym in syms))
    assert not missing, missing
    # the symbols as rocprofv3 prints them
    assert pmc_summary.profile_name("void sfx::k_radix_sweep_duo<sfx::SrcKV12, sfx::DstKV12, 14, 8>(sfx::SrcKV12, ...)") == "radix_scatter_u64"
    assert pmc_summary.profile_name("sfx::k_radix_sweep_duo<sfx::SrcKV12, sfx::DstKV, 14, 8>") == "radix_scatter_u64"
    assert pmc_summary.profile_name("sfx::k_radix_sweep<sfx::SrcKeyIota, sfx::DstKV12, 12, 16>") == "radix_scatter_u64"
    assert pmc_summary.profile_name("sfx::k_radix_pass<sfx::SrcE64, sfx::DstE64, 11, true, true, 16, true>") == "seg_radix_pass"
    assert pmc_summary.profile_name("sfx::k_radix_pass<sfx::SrcE64, sfx::DstE64, 11, true, true, 16, false>") == "radix_scatter_u32"
    assert pmc_summary.profile_name("sfx::k_lcp_intervals_open") == "tree_intervals_open"
    assert pmc_summary.profile_name("at::native::vectorized_elementwise_kernel<4>") is None


def _newest(pattern):
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", pattern)):
        m = re.match(r"r(\d+)_", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    return best


def test_fullsize_pmc_record_is_recomputable_from_the_committed_summaries(tmp_path):
    rnd, path = _newest("r*_pmc_fullsize.json")
    if rnd < 6:
        pytest.skip("no per-config PMC record of round 6 or later under profiles/ yet (scripts/gpu_profiles.sh)")
    rec = json.load(open(path))
    assert set(rec) >= {"c3", "c5", "dup", "c3r1", "c5r1"}, (path, sorted(rec))
    out = str(tmp_path / "re.json")
    for key, ent in rec.items():
        summary = os.path.join(ROOT, "profiles", f"r{rnd}_pmc_summary_{key}.csv")
        assert os.path.exists(summary), summary
        subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_summary.py"), "--fullsize", out, key, str(ent["builds"]), summary],
                       check=True, capture_output=True)
        again = json.load(open(out))[key]
        assert not ent["unmapped_symbols"], (key, ent["unmapped_symbols"])
        assert set(again["kernels"]) == set(ent["kernels"]), key
        for name, k in ent["kernels"].items():
            a = again["kernels"][name]
            assert a["launches"] == k["launches"] and a["symbols"] == k["symbols"], (key, name)
            for fld in ("fetch_bytes", "write_bytes", "hbm_bytes_per_build", "hbm_bytes_per_launch"):
                # (the summaries keep four decimals of a per-dispatch mean in KiB: a few bytes of rounding per dispatch)
                assert abs(a[fld] - k[fld]) <= max(64 * k["launches"] * ent["builds"], 1e-6 * k[fld]), (key, name, fld, a[fld], k[fld])
        # the raw counters are in KiB: the dominant pass's traffic straight from the CSV (FETCH x the copy calibration + WRITE)
        tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
        n = 0
        for row in csv.DictReader(open(summary)):
            if pmc_summary.profile_name(row["Kernel"]) == "radix_scatter_u64" and row["Counter"] in tot:
                tot[row["Counter"]] += float(row["MeanPerDispatch"]) * int(row["Dispatches"])
                n += int(row["Dispatches"]) if row["Counter"] == "FETCH_SIZE" else 0
        by_hand = (tot["FETCH_SIZE"] * ent["fetch_calibration_copy"] + tot["WRITE_SIZE"] * ent["write_calibration_copy"]) * 1024.0 / ent["builds"]
        k = ent["kernels"]["radix_scatter_u64"]
        # (the record keeps its calibration factors to four decimals)
        # (config 5 and its round-1 input sort twice: the pilot of 2^22 suffixes that decides on the context codes, then the text)
        assert abs(by_hand - k["hbm_bytes_per_build"]) <= 1e-4 * k["hbm_bytes_per_build"], key
        assert k["launches"] == n // ent["builds"] == (16 if key in ("c5", "c5r1") else 8), key


def test_committed_bench_record_has_a_measured_traffic_for_every_large_kernel():
    """The bench line of the round (profiles/rN_bench_100MB_dna.txt, a copy of what the driver's command prints): every kernel
    with >= 15 % of its build carries PMC traffic of at least 0.98 x its algorithmic bytes -- never null, never below what
    the kernel must move."""
    rnd, path = _newest("r*_bench_100MB_dna.txt")
    if rnd < 6:
        pytest.skip("no bench record of round 6 or later under profiles/ yet (scripts/gpu_final.sh)")
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import _benchout
    final, details = _benchout.load(open(path))
    for k in final["roofline"]["kernels"]:
        if k["share"] >= 0.15:
            assert k["traffic_ratio"] is not None and k["traffic_ratio"] >= 0.98, ("headline", k)
    seen = set()
    for c in final["roofline"]["configs"]:
        assert c.get("dominant"), c
        seen.add(c["key"])
        assert c["dominant"]["traffic_ratio"] is not None and c["dominant"]["traffic_ratio"] >= 0.98, c
    assert seen >= {"c3", "c5", "dup", "c3r1", "c5r1"}, seen
    for name, d in details.items():
        for k in (d.get("roofline") or {}).get("kernels") or []:
            if name in seen and k["share_of_build"] >= 0.15:
                assert k.get("traffic") and k["traffic"] >= 0.98 * k["algo_bytes_per_launch"], (name, k)
