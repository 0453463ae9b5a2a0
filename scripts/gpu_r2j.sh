#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r2j
O=gpurun_out/r2j/builds.jsonl; : > $O
export TIME_FUSED=1
for kind in eng utf8 engr1 dup dna; do
  timeout 400 python scripts/gpu_time_build.py $kind >> $O 2>> gpurun_out/r2j/err.log
done
python -c "
import json
for l in open('$O'):
    d=json.loads(l); print(d['kind'], d['sa_ms'], 'fused', d.get('fused_sa_lcp_ms'), d.get('fused_lcp_kernels_ms'), 'sep', d.get('separate_lcp_ms_cold'), d.get('fused_lcp_equals_separate'), d['stats']['text_rounds'], d['stats']['rank_rounds'], d.get('sha256_sa'), d.get('sha256_lcp'))
"
tail -3 gpurun_out/r2j/err.log
