#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r2h
O=gpurun_out/r2h/sweep.jsonl; : > $O
export TIME_SHA=0
SFX_SWITCH=text timeout 300 python scripts/gpu_time_build.py eng >> $O 2>> gpurun_out/r2h/err.log
SFX_SWITCH=rank timeout 300 python scripts/gpu_time_build.py eng >> $O 2>> gpurun_out/r2h/err.log
SFX_TILE_GEOM=2 timeout 300 python scripts/gpu_time_build.py eng >> $O 2>> gpurun_out/r2h/err.log
SFX_TILE_GEOM=0 timeout 300 python scripts/gpu_time_build.py eng >> $O 2>> gpurun_out/r2h/err.log
SFX_SWITCH=text timeout 300 python scripts/gpu_time_build.py utf8 >> $O 2>> gpurun_out/r2h/err.log
SFX_TILE_GEOM=2 timeout 300 python scripts/gpu_time_build.py utf8 >> $O 2>> gpurun_out/r2h/err.log
SFX_SWITCH=rank timeout 300 python scripts/gpu_time_build.py engr1 >> $O 2>> gpurun_out/r2h/err.log
python -c "
import json
for l in open('$O'):
    d=json.loads(l); print(d['kind'], d['env'], d['sa_ms'], d['stats']['text_rounds'], d['stats']['rank_rounds'], d['stats']['large_sorted']); print('   ', d['kernel_ms'])
"
