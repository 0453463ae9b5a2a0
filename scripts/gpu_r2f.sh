#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r2f
(time timeout 1200 python -m pytest tests -m gpu -x -q --durations=5) > gpurun_out/r2f/pytest.log 2>&1
tail -12 gpurun_out/r2f/pytest.log
O=gpurun_out/r2f/builds.jsonl; : > $O
for kind in eng utf8 dup engr1; do
  timeout 400 python scripts/gpu_time_build.py $kind >> $O 2>> gpurun_out/r2f/err.log
done
python -c "
import json
for l in open('$O'):
    d=json.loads(l); print(d['kind'], d['sa_ms'], d['stats']['rounds'], d['stats']['tile_sorted'], d['stats']['large_sorted'], d['stats']['elements_sorted'], d.get('sha256_sa')); print('   ', d['kernel_ms'])
"
tail -5 gpurun_out/r2f/err.log
