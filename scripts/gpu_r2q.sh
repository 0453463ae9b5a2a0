#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r2q
O=gpurun_out/r2q/builds.jsonl; : > $O
export TIME_FUSED=1
for kind in eng utf8 engr1 dup dna; do
  timeout 400 python scripts/gpu_time_build.py $kind >> $O 2>> gpurun_out/r2q/err.log
done
python -c "
import json
for l in open('$O'):
    d=json.loads(l); print(d['kind'], d['sa_ms'], 'fused', d.get('fused_sa_lcp_ms'), d.get('fused_lcp_equals_separate'), d.get('sha256_sa'), d.get('sha256_lcp')); print('   ', d['kernel_ms'])
"
(timeout 600 python -m pytest tests -m gpu -x -q -k "not fullsize") 2>&1 | tail -3
