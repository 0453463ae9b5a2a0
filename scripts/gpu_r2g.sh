#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r2g
O=gpurun_out/r2g/builds.jsonl; : > $O
for kind in eng utf8 engr1; do
  timeout 400 python scripts/gpu_time_build.py $kind >> $O 2>> gpurun_out/r2g/err.log
done
timeout 300 python bench.py --configs '' --cpu-sample 0 > gpurun_out/r2g/bench.json 2>> gpurun_out/r2g/err.log
python -c "
import json
for l in open('$O'):
    d=json.loads(l); print(d['kind'], d['sa_ms'], d.get('sha256_sa')); print('   ', d['kernel_ms'])
d=json.loads(open('gpurun_out/r2g/bench.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['lcp'])
"
