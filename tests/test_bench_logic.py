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
