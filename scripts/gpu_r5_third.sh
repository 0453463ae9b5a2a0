#!/bin/bash
# Round 5, third GPU call: rank rounds from the first round on (SFX_START_RANKS=0 / 1) on the high-LCP text and on config 5
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r5c
mkdir -p "$OUT"
export TMPDIR=/tmp
for kind in dup utf8; do
  for sr in 0 1; do
    SFX_LIB=suffix_amd/libsuffix_hip_dev.so SFX_START_RANKS=$sr SFX_TRACE=1 timeout 300 python scripts/gpu_time_build.py $kind >> "$OUT/start_ranks_ab.jsonl" 2>> "$OUT/start_ranks_ab.err"
  done
done
python - <<'PY' | tee "$OUT/summary.txt"
import json
for l in open("gpurun_out/r5c/start_ranks_ab.jsonl"):
    r = json.loads(l)
    print(r["kind"], r["env"].get("SFX_START_RANKS"), "sa_ms", r["sa_ms"], "sha", r.get("sha256_sa"), "rounds", r["stats"]["rounds"], r["kernel_ms"])
PY
grep "^round" "$OUT/start_ranks_ab.err" | sort | uniq -c | sort -k2,2n -k3 | head -60 >> "$OUT/summary.txt"
