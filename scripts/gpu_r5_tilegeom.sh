#!/bin/bash
# Round 5: the rank rounds' LDS bucket sort (k_tile_sort) at the geometries the development hook offers (SFX_TILE_GEOM)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r5y
mkdir -p "$OUT"
export TMPDIR=/tmp
for kind in dup utf8; do
  for g in 3 4 5 2; do
    SFX_LIB=suffix_amd/libsuffix_hip_dev.so SFX_TILE_GEOM=$g TIME_SHA=0 timeout 300 python scripts/gpu_time_build.py $kind >> "$OUT/tile_geom_ab.jsonl" 2>> "$OUT/tile_geom_ab.err"
  done
done
python - <<'PY' | tee "$OUT/summary.txt"
import json
for l in open("gpurun_out/r5y/tile_geom_ab.jsonl"):
    r = json.loads(l)
    print(r["kind"], r["env"].get("SFX_TILE_GEOM"), "sa_ms", r["sa_ms"], {k: v for k, v in r["kernel_ms"].items() if k in ("tile_sort", "seg_single_lds", "seg_radix_pass", "seg_gather", "groups_apply_u64")})
PY
