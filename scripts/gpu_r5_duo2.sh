#!/bin/bash
# Round 5: the E64 one-sweep passes by k_radix_sweep_duo (SFX_RADIX_DUO_E64) on the texts whose builds have many of them
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r5i
mkdir -p "$OUT"
export TMPDIR=/tmp
for kind in dup dna utf8; do
  for duo in 0 1; do
    SFX_LIB=suffix_amd/libsuffix_hip_dev.so SFX_RADIX_DUO_E64=$duo timeout 300 python scripts/gpu_time_build.py $kind >> "$OUT/duo_e64_ab.jsonl" 2>> "$OUT/duo_e64_ab.err"
  done
done
python - <<'PY' | tee "$OUT/summary.txt"
import json
for l in open("gpurun_out/r5i/duo_e64_ab.jsonl"):
    r = json.loads(l)
    print(r["kind"], r["env"].get("SFX_RADIX_DUO_E64"), "sa_ms", r["sa_ms"], "sha", r.get("sha256_sa"), {k: v for k, v in r["kernel_ms"].items() if "radix" in k})
PY
