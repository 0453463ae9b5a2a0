#!/bin/bash
# Round 5, first GPU call: the read-back regression test, a parity subset, the 12-byte-element A/B on configs 3 / 5 (development
# library, SFX_RADIX_KV12=0/1), then the default bench.
#   gpurun --timeout 1200 -- 'bash scripts/gpu_r5_first.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r5a
mkdir -p "$OUT"
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -6 > "$OUT/box.txt"; lscpu | grep "Model name" >> "$OUT/box.txt"; free -g | head -2 >> "$OUT/box.txt"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fresh_thread or literals or fasta or generated_medium or structured" > "$OUT/pytest_subset.log" 2>&1
echo "pytest subset rc=$?" | tee "$OUT/summary.txt"; tail -3 "$OUT/pytest_subset.log" | tee -a "$OUT/summary.txt"
for kind in eng utf8; do
  for kv in 0 1; do
    SFX_LIB=suffix_amd/libsuffix_hip_dev.so SFX_RADIX_KV12=$kv timeout 300 python scripts/gpu_time_build.py $kind >> "$OUT/kv12_ab.jsonl" 2>> "$OUT/kv12_ab.err"
  done
done
python - <<'PY' | tee -a "$OUT/summary.txt"
import json
for l in open("gpurun_out/r5a/kv12_ab.jsonl"):
    r = json.loads(l)
    print(r["kind"], r["env"].get("SFX_RADIX_KV12"), "sa_ms", r["sa_ms"], "sha", r.get("sha256_sa"), {k: v for k, v in r["kernel_ms"].items() if "radix" in k or "ht_keys" in k})
PY
timeout 900 python bench.py --steps 20 --warmup 2 > "$OUT/bench.txt" 2> "$OUT/bench.err"
echo "bench rc=$?" | tee -a "$OUT/summary.txt"; tail -1 "$OUT/bench.txt" | cut -c1-3000 | tee -a "$OUT/summary.txt"; tail -3 "$OUT/bench.err" | tee -a "$OUT/summary.txt"
