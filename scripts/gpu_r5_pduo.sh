#!/bin/bash
# Round 5: the partition passes of the headline with two workgroups per CU (SFX_PARTITION_DUO: bit 0 = first pass, bit 1 = second)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r5p
mkdir -p "$OUT"
export TMPDIR=/tmp
for duo in 0 1 2 3 0 3; do
  SFX_LIB=suffix_amd/libsuffix_hip_dev.so SFX_PARTITION_DUO=$duo timeout 300 python scripts/gpu_time_build.py dna 100000000 >> "$OUT/pduo_ab.jsonl" 2>> "$OUT/pduo_ab.err"
done
python - <<'PY' | tee "$OUT/summary.txt"
import json
for l in open("gpurun_out/r5p/pduo_ab.jsonl"):
    r = json.loads(l)
    print(r["env"].get("SFX_PARTITION_DUO"), "sa_ms", r["sa_ms"], "sha", r.get("sha256_sa"), {k: v for k, v in r["kernel_ms"].items() if "radix_scatter" in k or "bucket" in k})
PY
