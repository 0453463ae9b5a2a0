#!/bin/bash
# radix-pass variant sweep: "<sweep> <nw> <kpt> <rank>" per line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/sweep2; mkdir -p "$OUT"; export TMPDIR=/tmp
S="$OUT/summary.txt"; : > "$S"
for cfg in "1 8 16 1" "1 8 8 1" "1 16 8 1" "1 16 8 0" "0 16 8 1"; do
  set -- $cfg
  tag="sw$1_nw$2_kpt$3_rk$4"
  echo "== $tag" | tee -a "$S"
  SFX_RADIX_SWEEP=$1 SFX_RADIX_NW=$2 SFX_RADIX_KPT=$3 SFX_RADIX_RANK=$4 timeout 200 python bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-microbench > "$OUT/$tag.json" 2> "$OUT/$tag.err"
  echo "rc=$?" | tee -a "$S"
  python - "$OUT/$tag.json" <<'PY' | tee -a "$S"
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value MB/s", d["value"], "ms", d["ms_per_step"], "verified", d["verified"])
    print({k: v for k, v in d["roofline"]["kernel_ms"].items() if k.startswith("radix")})
except Exception as e:
    print("no result:", e)
PY
done
