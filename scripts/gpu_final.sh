#!/bin/bash
# The closing call of a round: smoke(), the whole `pytest -m gpu`, the default bench (what the driver runs) and a headline-only
# bench.    gpu_final.sh [TAG=r6]   -> gpurun_out/TAG_final/ -> profiles/TAG_bench_100MB_dna*.txt, profiles/TAG_pytest_gpu.log
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r6}; OUT=gpurun_out/${TAG}_final
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee "$OUT/summary.txt"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=10 > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc=$?" | tee -a "$OUT/summary.txt"; tail -16 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"
timeout 900 python bench.py > "$OUT/bench.txt" 2> "$OUT/bench.err"
echo "bench rc=$?" | tee -a "$OUT/summary.txt"; tail -1 "$OUT/bench.txt" | cut -c1-1200 | tee -a "$OUT/summary.txt"
timeout 900 python bench.py --steps 20 --warmup 2 --configs "" > "$OUT/bench20.txt" 2> "$OUT/bench20.err"
echo "bench20 rc=$?" | tee -a "$OUT/summary.txt"; tail -1 "$OUT/bench20.txt" | cut -c1-400 | tee -a "$OUT/summary.txt"
