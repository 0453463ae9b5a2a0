#!/bin/bash
# full-size parity pins: complete SA and LCP of configs 3 and 5 compared element by element with the oracle
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r2d
rm -f gpurun_out/big/results.jsonl
(time SFX_FULL_ORACLE=1 timeout 1500 python tests/fullsize_configs.py c3 c5) > gpurun_out/r2d/full_oracle.log 2>&1
cp gpurun_out/big/results.jsonl gpurun_out/r2d/full_oracle.jsonl
rm -f gpurun_out/big/results.jsonl
(time timeout 600 python tests/fullsize_configs.py dup dna1g) > gpurun_out/r2d/others.log 2>&1
cp gpurun_out/big/results.jsonl gpurun_out/r2d/others.jsonl
python -c "
import json
for f in ('full_oracle','others'):
  for l in open('gpurun_out/r2d/%s.jsonl' % f):
    d=json.loads(l); print(d['config'][:40], d['sa_ms'], d.get('lcp_ms'), d['verified'], d.get('full_oracle'))
"
