#!/bin/bash
# configs with rank rounds (and the partitioned-scatter test) after a kernel change: times, pins, top kernels
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "partitioned_scatter" 2>&1 | tail -2
timeout 700 python bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-microbench --configs ${CONFIGS:-c5,dup} --config-budget 400 2>/dev/null | python -c "
import json,sys
sys.path.insert(0,'scripts')
import _benchout
d=_benchout.legacy(*_benchout.load(sys.stdin))
print(d['value'], d['ms_per_step'], d['verified'])
for c in d['configs']:
    print(c['config'][:40], c.get('sa_ms'), c.get('lcp_ms'), c.get('fused_sa_lcp',{}).get('ms'), c.get('bit_exact_vs_pins'), {k:v for k,v in list(c['top_kernels_ms'].items())[:7]})"
