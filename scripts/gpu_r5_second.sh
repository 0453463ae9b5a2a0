#!/bin/bash
# Round 5, second GPU call: the whole GPU suite (config 4 against its oracle pin, the new regression / rehearsal tests), then
# the LCP window A/B (16 / 32 bytes) and the KV12 build times with one 12-byte load per element.
#   gpurun --timeout 2400 -- 'bash scripts/gpu_r5_second.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r5b
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc=$?" | tee "$OUT/summary.txt"; tail -40 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"
for kind in eng1g utf1g; do
  for w in 2 4; do
    SFX_LIB=suffix_amd/libsuffix_hip_dev.so SFX_LCP_WINDOW=$w timeout 300 python scripts/gpu_lcp_prof.py $kind >> "$OUT/lcp_window_ab.jsonl" 2>> "$OUT/lcp_window_ab.err"
  done
done
python - <<'PY' | tee -a "$OUT/summary.txt"
import json
for l in open("gpurun_out/r5b/lcp_window_ab.jsonl"):
    r = json.loads(l)
    print(r["text"], r["env"].get("SFX_LCP_WINDOW"), "lcp_ms", r["lcp_ms"], r["lcp_kernels"], "sum", r["lcp_sum"], "same", r["same"])
PY
for kind in eng utf8; do
  SFX_LIB=suffix_amd/libsuffix_hip_dev.so timeout 300 python scripts/gpu_time_build.py $kind >> "$OUT/kv12_fetch.jsonl" 2>> "$OUT/kv12_fetch.err"
done
python - <<'PY' | tee -a "$OUT/summary.txt"
import json
for l in open("gpurun_out/r5b/kv12_fetch.jsonl"):
    r = json.loads(l)
    print(r["kind"], "sa_ms", r["sa_ms"], "sha", r.get("sha256_sa"), {k: v for k, v in r["kernel_ms"].items() if "radix" in k or "ht_keys" in k or "deep" in k})
PY
