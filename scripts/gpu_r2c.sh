#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r2c
O=gpurun_out/r2c/sweep.jsonl; : > $O
for kind in eng utf8; do
for g in 0 1 2 3; do for pm in 64 32; do
  SFX_TILE_GEOM=$g SFX_TILE_PAIR=$pm timeout 300 python scripts/gpu_time_build.py $kind >> $O 2>> gpurun_out/r2c/err.log
done; done; done
SFX_TILE_GEOM=2 timeout 300 python scripts/gpu_time_build.py dup >> $O 2>> gpurun_out/r2c/err.log
SFX_TILE_GEOM=0 timeout 300 python scripts/gpu_time_build.py dup >> $O 2>> gpurun_out/r2c/err.log
python -c "
import json
for l in open('$O'):
    d=json.loads(l); print(d['kind'], d['env'], d['sa_ms'], d['kernel_ms'].get('tile_sort'), d['stats']['large_sorted'], d.get('sha256_sa'))
"
