#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per (kernel, counter) mean value per dispatch.
usage: pmc_summary.py <dir-with-*counter_collection.csv> [...]   -> CSV on stdout
       pmc_summary.py --json OUT.json <dirs...>                   also writes the per-kernel
           HBM-traffic JSON bench.py reads (profiles/pmc_latest.json): FETCH_SIZE / WRITE_SIZE
           are reported in KiB; they are calibrated on the known byte counts of the
           sfx::k_mb_copy launch of the same run (bench.py --calibrate), as
           MI355X_MICROARCH.md's HBM section prescribes (FETCH_SIZE reads half of a wide
           streaming read on gfx950; WRITE_SIZE is uncalibrated).
       pmc_summary.py --fullsize OUT.json KEY BUILDS <dirs...>    totals per build of every engine kernel of a 1 GB build,
           merged into OUT.json under KEY (profiles/rN_pmc_fullsize.json)

The engine reports its kernels under PROFILE names (sfx_kernel_stat.name, the first argument of SFX_LAUNCH); rocprofv3 reports
kernel SYMBOLS.  NAMES maps one to the other: every `__global__` kernel of suffix_amd/csrc has an entry (longest matching
prefix wins), which tests/test_bench_logic.py checks against the sources -- round 5 shipped without `k_radix_sweep_duo`, the
kernel that is 46 % of config 3, and filed one launch of another kernel under its name."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

# (symbol prefix as rocprofv3 prints it after "sfx::", profile name of the engine).  Longest prefix wins.
NAMES = [
    # -- device-wide passes (sfx_radix.hip)
    ("k_partition<sfx::SrcText32", "radix_scatter_text_u32"), ("k_partition<sfx::SrcText36", "radix_scatter_text_u32"), ("k_partition<sfx::SrcE64", "radix_scatter_u32"),
    ("k_radix_sweep<sfx::SrcE64", "radix_scatter_u32"), ("k_radix_sweep<sfx::SrcText32", "radix_scatter_text_u32"),
    ("k_radix_sweep<sfx::SrcKV", "radix_scatter_u64"), ("k_radix_sweep<sfx::SrcKeyIota", "radix_scatter_u64"),
    ("k_radix_sweep<sfx::SrcText64", "radix_scatter_text_u64"),
    ("k_radix_sweep_duo<sfx::SrcKV", "radix_scatter_u64"), ("k_radix_sweep_duo<sfx::SrcE64", "radix_scatter_u32"),
    ("k_radix_pass<sfx::SrcE64, sfx::DstE64, 11, true, true, 16, true", "seg_radix_pass"),
    ("k_radix_pass<sfx::SrcE64", "radix_scatter_u32"), ("k_radix_pass<sfx::SrcText32", "radix_scatter_text_u32"),
    ("k_radix_pass<sfx::SrcKV", "radix_scatter_u64"), ("k_radix_pass<sfx::SrcKeyIota", "radix_scatter_u64"),
    ("k_radix_pass<sfx::SrcText64", "radix_scatter_text_u64"),
    ("k_radix_hist_all<sfx::SrcText32", "radix_hist_all_text_u32"), ("k_radix_hist_all<sfx::SrcText64", "radix_hist_all_text_u64"),
    ("k_radix_hist_all<sfx::SrcE64", "radix_hist_all_u32"), ("k_radix_hist_all<sfx::SrcKV", "radix_hist_all_u64"),
    ("k_radix_hist_chunk", "radix_hist"), ("k_radix_scan", "radix_scan"),
    ("k_window_hist", "radix_hist_all_text_u32"), ("k_window_fix", "radix_window_fix"), ("k_window_from_hist16", "radix_window_from_hist16"),
    ("k_hist16_text", "radix_hist16_text"), ("k_hist16_e64", "radix_hist16_elems"), ("k_hist16_reduce", "radix_hist16_reduce"),
    ("k_hist16_scan", "radix_hist16_scan"), ("k_hist16_oversize", "radix_hist16_oversize"), ("k_partition_cursors", "partition_cursors"),
    ("k_bucket_sort<4, 8, true, true", "bucket_sort_ties_keys"), ("k_bucket_sort<4, 16, true, true", "bucket_sort_ties_keys"),
    ("k_bucket_sort<16, 16, true, true", "bucket_sort_ties_keys"),
    ("k_bucket_sort<4, 8, true", "bucket_sort_ties"), ("k_bucket_sort<4, 16, true", "bucket_sort_ties"),
    ("k_bucket_sort<16, 16, true", "bucket_sort_ties"), ("k_bucket_sort", "bucket_sort_lds"), ("k_oversize_gather", "oversize_gather"), ("k_oversize_return", "oversize_return"),
    ("k_ht_keys_ctx", "ht_keys"), ("k_ht_keys", "ht_keys"), ("k_bigram_hist", "bigram_hist"), ("k_seg_layout", "seg_layout"), ("k_seg_gather", "seg_gather"), ("k_seg_hist", "seg_hist"),
    ("k_seg_scan", "seg_scan"), ("k_seg_finish", "seg_finish"), ("k_scatter_pairs", "scatter_pairs"),
    # -- the build (sfx_sa.hip, sfx_tile.hip, sfx_tiny.hip)
    ("k_byte_presence", "byte_presence"), ("k_byte_hist", "byte_hist"), ("k_make_lut", "make_lut"), ("k_pack_text", "pack_text"),
    ("k_key_hist_raw", "key_hist"), ("k_range_filter", "range_count"),
    ("k_groups_reduce", "groups_reduce"), ("k_groups_scan", "groups_scan"),
    ("k_groups_apply<unsigned int", "groups_apply_u32"), ("k_groups_apply<unsigned long", "groups_apply_u64"),
    ("k_tie_direct", "tie_direct"), ("k_tie_totals", "tie_totals"), ("k_tie_heads", "tie_heads"), ("k_tie_list", "tie_list"),
    ("k_small_groups", "small_groups"), ("k_flag_compact", "flag_compact"), ("k_scan_block_counts", "flag_scan"),
    ("k_fill_u16", "depth_fill"), ("k_flags_reduce", "flags_reduce"),
    ("k_compose_rank_keys", "compose_rank_keys"), ("k_compose_text_keys", "compose_text_keys"),
    ("k_iota", "rank_iota"), ("k_scatter_by_slot", "rank_active_slots"), ("k_head_slots", "rank_head_slots"),
    ("k_isa_from_sa", "isa_from_sa"), ("k_rank_pairs", "rank_pairs"),
    ("k_tile_sort", "tile_sort"), ("k_deep_wave", "deep_wave"), ("k_deep_totals", "deep_totals"), ("k_seg_single", "seg_single_lds"),
    ("k_tiny_sa", "tiny_sa"), ("k_widen", "widen_u64"),
    # -- LCP, tree, queries
    ("k_lcp_sample", "lcp_sample"), ("k_lcp_windows_packed", "lcp_windows_packed"), ("k_lcp_windows", "lcp_windows"),
    ("k_lcp_direct", "lcp_direct"), ("k_lcp_pending", "lcp_pending"), ("k_lcp_tail_fix", "lcp_tail_fix"), ("k_lcp_gather", "lcp_gather"),
    ("k_phi_scatter", "phi_scatter"), ("k_phi_pairs", "phi_pairs"), ("k_plcp", "plcp"),
    ("k_pyr_reduce", "tree_pyramid"), ("k_lcp_intervals_open", "tree_intervals_open"), ("k_lcp_intervals", "tree_intervals"),
    ("k_tree_parents", "tree_parents"), ("k_tree_leaves", "tree_leaves"), ("k_tree_level", "tree_level"),
    ("k_dir_mark", "dir_mark"), ("k_dir_block_min", "dir_block_min"), ("k_dir_scan_mins", "dir_scan_mins"), ("k_dir_fill", "dir_fill"),
    ("k_doc_lookup", "doc_lookup"), ("k_query_keys", "query_keys"), ("k_query_batch_tree", "query_batch_tree"),
    ("k_query_batch_dir", "query_batch_dir"), ("k_query_tree_long", "query_tree_long"), ("k_query_batch", "query_batch"),
    # -- not part of a build's profile: the polled read-back and the memory-system probes
    ("detail::k_post_words", "post_words"), ("k_mb_copy", "mb_copy"), ("k_mb_gather", "mb_gather"), ("k_mb_scatter", "mb_scatter"),
    ("k_mb_runscatter", "mb_runscatter"),
]


def profile_name(symbol):
    """Profile name of a rocprofv3 kernel symbol ("void sfx::k_x<...>(...)" in any of its spellings), or None."""
    short = symbol.split("(")[0].replace("void ", "")
    if short.startswith("sfx::"):
        short = short[5:]
    best = None
    for sym, prof in NAMES:
        if short.startswith(sym) and (best is None or len(sym) > len(best[0])):
            best = (sym, prof)
    return best[1] if best else None


def read_counters(dirs):
    acc = defaultdict(lambda: [0.0, 0])
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    short = row.get("Kernel_Name", "").split("(")[0].replace("void ", "")
                    key = (short, row.get("Counter_Name", ""))
                    acc[key][0] += float(row.get("Counter_Value", 0) or 0)
                    acc[key][1] += 1
    return acc


def read_summary_csv(paths):
    """The same accumulator from the summaries this script prints (profiles/rN_pmc_summary_*.csv: Kernel, Counter, Dispatches,
    MeanPerDispatch) -- so that every profiles/rN_pmc_fullsize.json entry can be recomputed from what is committed."""
    acc = defaultdict(lambda: [0.0, 0])
    for path in paths:
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                n = int(row["Dispatches"])
                acc[(row["Kernel"], row["Counter"])][0] += float(row["MeanPerDispatch"]) * n
                acc[(row["Kernel"], row["Counter"])][1] += n
    return acc


def totals_by_profile(acc, counter):
    """{profile name: [sum of the counter over all its symbols, dispatches]} + the symbols nobody maps"""
    out, unmapped = defaultdict(lambda: [0.0, 0, set()]), set()
    for (k, c), (t, n) in acc.items():
        if c != counter:
            continue
        prof = profile_name(k)
        if prof is None:
            if k.startswith("sfx::"):
                unmapped.add(k)
            continue
        out[prof][0] += t
        out[prof][1] += n
        out[prof][2].add(k)
    return out, unmapped


def calibration(fetch, write):
    # the 1 GiB streaming copy: 2^30 bytes read, 2^30 bytes written per launch
    known = float(1 << 30)
    fc, wc = fetch.get("mb_copy"), write.get("mb_copy")
    f_cal = known / (fc[0] / fc[1] * 1024.0) if fc and fc[1] else 2.0
    w_cal = known / (wc[0] / wc[1] * 1024.0) if wc and wc[1] else 1.0
    return f_cal, w_cal


def main(args):
    json_out = full_out = full_key = None
    full_builds = 1
    if args and args[0] == "--json":
        json_out, args = args[1], args[2:]
    elif args and args[0] == "--fullsize":
        # (a kernel runs at a different size in every round, so per-launch means of single symbols say nothing: totals per build)
        full_out, full_key, full_builds, args = args[1], args[2], int(args[3]), args[4:]
    # (--fullsize ... file.csv: recompute from a committed summary instead of raw rocprofv3 output)
    from_summary = bool(args) and all(a.endswith(".csv") and os.path.isfile(a) for a in args)
    acc = read_summary_csv(args) if from_summary else read_counters(args)
    w = csv.writer(sys.stdout)
    w.writerow(["Kernel", "Counter", "Dispatches", "MeanPerDispatch"])
    for (k, c), (tot, cnt) in sorted(acc.items()):
        w.writerow([k, c, cnt, f"{tot / max(cnt, 1):.4f}"])
    if not (json_out or full_out):
        return
    fetch, unm = totals_by_profile(acc, "FETCH_SIZE")
    write, _ = totals_by_profile(acc, "WRITE_SIZE")
    f_cal, w_cal = calibration(fetch, write)
    skip = ("mb_copy", "mb_gather", "mb_scatter", "mb_runscatter", "post_words")
    if json_out:
        out = {"commit": os.environ.get("SFX_COMMIT", "unknown"),
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over `bench.py --steps 1 --calibrate` "
                         "(scripts/gpu_pmc.sh); counters in KiB x calibration factor from sfx::k_mb_copy (1 GiB in, 1 GiB out); "
                         "all symbols of one profile name summed, divided by their launches",
               "fetch_calibration": round(f_cal, 4), "write_calibration": round(w_cal, 4), "unmapped_symbols": sorted(unm), "kernels": {}}
        for prof, (f, nf, syms) in sorted(fetch.items()):
            wr = write.get(prof)
            if prof in skip or not nf or not wr or not wr[1]:
                continue
            fb, wb = f / nf * 1024.0 * f_cal, wr[0] / wr[1] * 1024.0 * w_cal
            out["kernels"][prof] = {"fetch_bytes": round(fb), "write_bytes": round(wb), "hbm_bytes_per_launch": round(fb + wb),
                                    "launches": nf, "symbols": sorted(syms)}
        json.dump(out, open(json_out, "w"), indent=1)
    if full_out:
        g = fetch.get("mb_gather")
        ent = {"commit": os.environ.get("SFX_COMMIT", "unknown"), "builds": full_builds,
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over scripts/gpu_time_build.py (scripts/gpu_pmc_fullsize.sh); "
                         "KiB x the streaming-copy calibration; totals per build, all symbols of one profile name summed",
               "fetch_calibration_copy": round(f_cal, 4), "write_calibration_copy": round(w_cal, 4),
               # what ONE random 4-byte read costs at the HBM side with the same (copy) calibration: 2^28 of them per launch
               "gather_calibration": {"reads_per_launch": 1 << 28,
                                      "fetched_bytes_per_read": round(g[0] / g[1] * 1024.0 * f_cal / float(1 << 28), 1) if g and g[1] else None},
               "unmapped_symbols": sorted(unm), "kernels": {}}
        for prof, (f, nf, syms) in sorted(fetch.items()):
            wr = write.get(prof, [0.0, 0, set()])
            if prof in skip or not nf:
                continue
            fb, wb = f * 1024.0 * f_cal / full_builds, wr[0] * 1024.0 * w_cal / full_builds
            launches = nf // full_builds
            ent["kernels"][prof] = {"fetch_bytes": round(fb), "write_bytes": round(wb), "launches": launches,
                                    "hbm_bytes_per_build": round(fb + wb), "hbm_bytes_per_launch": round((fb + wb) / max(launches, 1)),
                                    "symbols": sorted(syms)}
        try:
            allc = json.load(open(full_out))
        except (OSError, ValueError):
            allc = {}
        allc[full_key] = ent
        json.dump(allc, open(full_out, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
