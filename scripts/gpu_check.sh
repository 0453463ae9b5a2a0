#!/bin/bash
# One gpurun call: smoke, GPU parity tests, bench, rocprofv3 kernel stats.
# Usage (from the build container):
#   gpurun --timeout 1500 -- 'bash scripts/gpu_check.sh [quick]'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/check
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== rocm-smi" | tee "$OUT/summary.txt"
rocm-smi --showproductname 2>/dev/null | head -8 | tee -a "$OUT/summary.txt"
nproc | tee -a "$OUT/summary.txt"; lscpu | grep "Model name" | tee -a "$OUT/summary.txt"

echo "== smoke" | tee -a "$OUT/summary.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
echo "smoke rc=$?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/smoke.log" | tee -a "$OUT/summary.txt"

echo "== pytest -m gpu" | tee -a "$OUT/summary.txt"
timeout 900 python -m pytest tests -m gpu -x -q --durations=15 > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc=$?" | tee -a "$OUT/summary.txt"; tail -25 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"

echo "== bench" | tee -a "$OUT/summary.txt"
timeout 600 python bench.py --steps 5 --warmup 1 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?" | tee -a "$OUT/summary.txt"; cat "$OUT/bench.json" | tee -a "$OUT/summary.txt"; tail -5 "$OUT/bench.err" | tee -a "$OUT/summary.txt"

if [ "${1:-}" != "quick" ]; then
  echo "== rocprofv3 kernel stats" | tee -a "$OUT/summary.txt"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof" -o bench -- \
      python "$OLDPWD/bench.py" --steps 3 --warmup 1 --cpu-sample 0 --no-verify) > "$OUT/rocprof.log" 2>&1
  echo "rocprof rc=$?" | tee -a "$OUT/summary.txt"
  find "$OUT/prof" -name "*kernel_stats*" | head -3 | tee -a "$OUT/summary.txt"
  f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -25 "$f" | tee -a "$OUT/summary.txt"
  # keep the merge-back small: drop raw traces, keep stats
  find "$OUT/prof" -name "*kernel_trace.csv" -size +20M -delete
fi
echo "== done" | tee -a "$OUT/summary.txt"
