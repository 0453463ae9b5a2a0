#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r2p
O=gpurun_out/r2p/sweep.jsonl; : > $O
export TIME_SHA=0
for g in 3 4 5; do
  SFX_TILE_GEOM=$g timeout 300 python scripts/gpu_time_build.py eng >> $O 2>> gpurun_out/r2p/err.log
  SFX_TILE_GEOM=$g timeout 300 python scripts/gpu_time_build.py utf8 >> $O 2>> gpurun_out/r2p/err.log
done
SFX_TILE_GEOM=4 timeout 300 python scripts/gpu_time_build.py dup >> $O 2>> gpurun_out/r2p/err.log
timeout 200 python scripts/gpu_widen.py > gpurun_out/r2p/widen.json 2>> gpurun_out/r2p/err.log
python -c "
import json
for l in open('$O'):
    d=json.loads(l); k=d['kernel_ms']; print(d['kind'], d['env'], d['sa_ms'], 'tile', k.get('tile_sort'), 'segpass', k.get('seg_radix_pass'), 'gather', k.get('seg_gather'), d['stats']['large_sorted'])
"
cat gpurun_out/r2p/widen.json
