#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per (kernel, counter) mean value per dispatch.
usage: pmc_summary.py <dir-with-*counter_collection.csv> [...]   -> CSV on stdout
       pmc_summary.py --json OUT.json <dirs...>                   also writes the per-kernel
           HBM-traffic JSON bench.py reads (profiles/pmc_latest.json): FETCH_SIZE / WRITE_SIZE
           are reported in KiB; they are calibrated on the known byte counts of the
           sfx::k_mb_copy launch of the same run (bench.py --calibrate), as
           MI355X_MICROARCH.md's HBM section prescribes (FETCH_SIZE reads half of a wide
           streaming read on gfx950; WRITE_SIZE is uncalibrated)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

args = sys.argv[1:]
json_out = None
full_out = full_key = None
full_builds = 1
if args and args[0] == "--json":
    json_out, args = args[1], args[2:]
elif args and args[0] == "--fullsize":
    # --fullsize OUT.json KEY BUILDS dirs...: totals per build of every engine kernel (a kernel runs at a different size in
    # every round, so per-launch means say nothing), merged into OUT.json under KEY (profiles/r3_pmc_fullsize.json)
    full_out, full_key, full_builds, args = args[1], args[2], int(args[3]), args[4:]

# profile name of the engine (sfx_kernel_stat.name) for each kernel symbol
NAMES = [("k_partition<sfx::SrcText32", "radix_scatter_text_u32"), ("k_partition<sfx::SrcE64", "radix_scatter_u32"),
         ("k_radix_sweep<sfx::SrcE64", "radix_scatter_u32"), ("k_radix_sweep<sfx::SrcText32", "radix_scatter_text_u32"),
         ("k_radix_sweep<sfx::SrcKV", "radix_scatter_u64"), ("k_radix_sweep<sfx::SrcKeyIota", "radix_scatter_u64"),
         ("k_radix_sweep<sfx::SrcText64", "radix_scatter_text_u64"), ("k_tiny_sa", "tiny_sa"),
         ("k_radix_pass<sfx::SrcE64, sfx::DstE64, 11, true, true, 16, false", "radix_scatter_u32"), ("k_radix_pass<sfx::SrcE64, sfx::DstSplit32", "radix_scatter_u32"), ("k_radix_pass<sfx::SrcText32", "radix_scatter_text_u32"),
         ("k_radix_pass<sfx::SrcKV", "radix_scatter_u64"), ("k_radix_pass<sfx::SrcText64", "radix_scatter_text_u64"),
         ("k_groups_apply<unsigned int", "groups_apply_u32"), ("k_groups_apply<unsigned long", "groups_apply_u64"),
         ("k_groups_reduce", "groups_reduce"), ("k_radix_hist_all<sfx::SrcText32", "radix_hist_all_text_u32"),
         ("k_pack_text", "pack_text"), ("k_small_groups", "small_groups"), ("k_byte_presence", "byte_presence"),
         ("k_tile_sort", "tile_sort"), ("k_seg_gather", "seg_gather"), ("k_lcp_windows_packed", "lcp_windows_packed"),
         ("k_lcp_pending", "lcp_pending"), ("k_bucket_sort", "bucket_sort_lds"), ("k_hist16_text", "radix_hist16_text"),
         ("k_hist16_reduce", "radix_hist16_reduce"), ("k_radix_pass<sfx::SrcKeyIota", "radix_scatter_u64"),
         ("k_deep_wave", "deep_wave"), ("k_seg_single", "seg_single_lds"), ("k_ht_keys", "ht_keys"),
         ("k_radix_pass<sfx::SrcE64, sfx::DstE64, 11, true, true, 16, true", "seg_radix_pass"), ("k_seg_hist", "seg_hist"),
         ("k_seg_finish", "seg_finish"), ("k_flags_reduce", "flags_reduce"), ("k_scatter_pairs", "scatter_pairs"),
         ("k_rank_pairs", "rank_pairs"), ("k_radix_hist_all<sfx::SrcE64", "radix_hist_all_u32")]

acc = defaultdict(lambda: [0.0, 0])
for d in args:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                short = name.split("(")[0].replace("void ", "")
                key = (short, row.get("Counter_Name", ""))
                acc[key][0] += float(row.get("Counter_Value", 0) or 0)
                acc[key][1] += 1
w = csv.writer(sys.stdout)
w.writerow(["Kernel", "Counter", "Dispatches", "MeanPerDispatch"])
for (k, c), (tot, cnt) in sorted(acc.items()):
    w.writerow([k, c, cnt, f"{tot / max(cnt, 1):.1f}"])

if json_out:
    def mean(kernel_prefix, counter):
        tot = cnt = 0
        for (k, c), (t, n) in acc.items():
            if c == counter and kernel_prefix in k:
                tot += t
                cnt += n
        return (tot / cnt, cnt) if cnt else (None, 0)

    # calibration on the 1 GiB streaming copy: 2^30 bytes read, 2^30 bytes written per launch
    f_copy, _ = mean("k_mb_copy", "FETCH_SIZE")
    w_copy, _ = mean("k_mb_copy", "WRITE_SIZE")
    known = float(1 << 30)
    f_cal = known / (f_copy * 1024.0) if f_copy else 2.0
    w_cal = known / (w_copy * 1024.0) if w_copy else 1.0
    out = {"commit": os.environ.get("SFX_COMMIT", "unknown"),
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over `bench.py --steps 1 --calibrate` "
                     "(scripts/gpu_pmc.sh); counters in KiB x calibration factor from sfx::k_mb_copy (1 GiB in, 1 GiB out)",
           "fetch_calibration": round(f_cal, 4), "write_calibration": round(w_cal, 4), "kernels": {}}
    for sym, prof in NAMES:
        f, nf = mean(sym, "FETCH_SIZE")
        wr, _ = mean(sym, "WRITE_SIZE")
        if f is None or wr is None:
            continue
        fb, wb = f * 1024.0 * f_cal, wr * 1024.0 * w_cal
        out["kernels"][prof] = {"fetch_bytes": round(fb), "write_bytes": round(wb), "hbm_bytes_per_launch": round(fb + wb),
                                "launches": nf, "symbol": sym}
    json.dump(out, open(json_out, "w"), indent=1)

if full_out:
    def total(kernel_prefix, counter):
        tot = cnt = 0
        for (k, c), (t, n) in acc.items():
            if c == counter and kernel_prefix in k:
                tot += t
                cnt += n
        return tot, cnt
    f_copy, nfc = total("k_mb_copy", "FETCH_SIZE")
    w_copy, nwc = total("k_mb_copy", "WRITE_SIZE")
    f_g, nfg = total("k_mb_gather<unsigned int>", "FETCH_SIZE")
    known = float(1 << 30)
    f_cal = known / (f_copy / nfc * 1024.0) if nfc else 2.0
    w_cal = known / (w_copy / nwc * 1024.0) if nwc else 1.0
    ent = {"commit": os.environ.get("SFX_COMMIT", "unknown"), "builds": full_builds,
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over scripts/gpu_time_build.py (scripts/gpu_pmc_fullsize.sh); "
                     "KiB x the streaming-copy calibration; totals per build",
           "fetch_calibration_copy": round(f_cal, 4), "write_calibration_copy": round(w_cal, 4),
           # what ONE random 4-byte read costs at the HBM side with the same (copy) calibration: 2^28 of them per launch
           "gather_calibration": {"reads_per_launch": 1 << 28,
                                  "fetched_bytes_per_read": round(f_g / nfg * 1024.0 * f_cal / float(1 << 28), 1) if nfg else None},
           "kernels": {}}
    seen = {}
    for sym, prof in NAMES:
        f, nf = total(sym, "FETCH_SIZE")
        wr, _ = total(sym, "WRITE_SIZE")
        if not nf:
            continue
        k = seen.setdefault(prof, {"fetch_bytes": 0.0, "write_bytes": 0.0, "launches": 0})
        k["fetch_bytes"] += f * 1024.0 * f_cal / full_builds
        k["write_bytes"] += wr * 1024.0 * w_cal / full_builds
        k["launches"] += nf // full_builds
    for prof, k in seen.items():
        k["hbm_bytes_per_build"] = round(k["fetch_bytes"] + k["write_bytes"])
        k["hbm_bytes_per_launch"] = round((k["fetch_bytes"] + k["write_bytes"]) / max(k["launches"], 1))
        k["fetch_bytes"], k["write_bytes"] = round(k["fetch_bytes"]), round(k["write_bytes"])
        ent["kernels"][prof] = k
    try:
        allc = json.load(open(full_out))
    except (OSError, ValueError):
        allc = {}
    allc[full_key] = ent
    json.dump(allc, open(full_out, "w"), indent=1)
