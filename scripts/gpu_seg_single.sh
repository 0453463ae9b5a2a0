#!/bin/bash
# large buckets that fit one tile sorted in LDS (k_seg_single): the 1 GB configs with the kernel on / off
mkdir -p gpurun_out
for v in "SFX_SEG_SINGLE=1"; do
  echo "== $v"
  env $v timeout 600 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-microbench --no-verify --configs c3,c5,dup --config-budget 400 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for c in d['configs']:
    print(c['config'][:40], c.get('sa_ms'), c.get('fused_sa_lcp',{}).get('ms'), c.get('bit_exact_vs_pins'), c['build']['large_sorted'], {k:v for k,v in c['top_kernels_ms'].items() if 'seg' in k or 'tile' in k})"
done
