#!/bin/bash
# One gpurun call: radix-pass variant sweep on the headline workload, then the GPU parity
# tests with the default variant.  Usage: gpurun --timeout 1500 -- 'bash scripts/gpu_sweep.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/sweep
mkdir -p "$OUT"
export TMPDIR=/tmp
S="$OUT/summary.txt"
: > "$S"
echo "== smoke (default variant)" | tee -a "$S"
timeout 400 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
echo "smoke rc=$?" | tee -a "$S"; tail -3 "$OUT/smoke.log" | tee -a "$S"
for cfg in "1 8 1" "1 16 1" "1 8 0" "1 16 0" "0 8 1" "0 16 1" "0 8 0" "0 16 0"; do
  set -- $cfg
  tag="sweep$1_kpt$2_rank$3"
  echo "== $tag" | tee -a "$S"
  SFX_RADIX_SWEEP=$1 SFX_RADIX_KPT=$2 SFX_RADIX_RANK=$3 timeout 200 python bench.py --steps 5 --warmup 1 --cpu-sample 0 \
      > "$OUT/$tag.json" 2> "$OUT/$tag.err"
  echo "rc=$?" | tee -a "$S"
  python - "$OUT/$tag.json" <<'PY' | tee -a "$S"
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value MB/s", d["value"], "ms", d["ms_per_step"], "verified", d["verified"])
    print({k: v for k, v in d["roofline"]["kernel_ms"].items()})
except Exception as e:
    print("no result:", e)
PY
  tail -2 "$OUT/$tag.err" | tee -a "$S"
done
echo "== pytest -m gpu (default variant)" | tee -a "$S"
timeout 900 python -m pytest tests -m gpu -x -q --durations=10 > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc=$?" | tee -a "$S"; tail -20 "$OUT/pytest_gpu.log" | tee -a "$S"
echo "== done" | tee -a "$S"
