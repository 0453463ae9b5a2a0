#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/run3; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python bench.py --steps 5 --warmup 1 --cpu-sample 0 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
timeout 300 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 1500 python scripts/gpu_big.py > $OUT/big.log 2>&1; echo "big rc=$?"; tail -5 $OUT/big.log
cat gpurun_out/big/results.jsonl
