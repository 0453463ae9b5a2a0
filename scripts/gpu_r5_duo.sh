#!/bin/bash
# Round 5: the KV passes with two workgroups per CU (k_radix_sweep_duo, SFX_RADIX_DUO = elements per thread) against k_radix_sweep
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r5f
mkdir -p "$OUT"
export TMPDIR=/tmp
for duo in 0 16 14; do
  SFX_LIB=suffix_amd/libsuffix_hip_dev.so SFX_RADIX_DUO=$duo timeout 300 python scripts/gpu_time_build.py eng >> "$OUT/duo_ab.jsonl" 2>> "$OUT/duo_ab.err"
done
python - <<'PY' | tee "$OUT/summary.txt"
import json
for l in open("gpurun_out/r5f/duo_ab.jsonl"):
    r = json.loads(l)
    print(r["kind"], r["env"].get("SFX_RADIX_DUO"), "sa_ms", r["sa_ms"], "sha", r.get("sha256_sa"), {k: v for k, v in r["kernel_ms"].items() if "radix" in k or "ht_keys" in k})
PY
