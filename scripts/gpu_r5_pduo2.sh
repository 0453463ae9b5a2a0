#!/bin/bash
# Round 5: headline bench (50 steps) through the development library, partition passes with one / two / four workgroups per CU (SFX_PARTITION_WAVES = 16 / 8 / 4)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r5r
mkdir -p "$OUT"
export TMPDIR=/tmp
for duo in 16 8 4 16 8 4; do
  SFX_LIB=suffix_amd/libsuffix_hip_dev.so SFX_PARTITION_WAVES=$duo timeout 300 python bench.py --steps 50 --warmup 3 --configs "" --cpu-sample 0 --no-microbench 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print(json.dumps({'waves_per_workgroup': $duo, 'ms_per_step': r['ms_per_step'], 'MBps': r['value'], 'verified': r['verified'], 'tied': r['roofline']['kernels_tied'], 'kernels': r['roofline']['kernels'][:3]}))" >> "$OUT/pduo_bench.jsonl"
done
cat "$OUT/pduo_bench.jsonl" | cut -c1-420
