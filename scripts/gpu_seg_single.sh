#!/bin/bash
# the 1 GB configs (and the headline) after a kernel change: times, pins, top kernels
mkdir -p gpurun_out
timeout 700 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-microbench --configs c3,c5,dup,c5r1 --config-budget 400 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print(d['value'], d['ms_per_step'], d['verified'], d['lcp']['fused_sa_lcp']['ms_per_step'], {a:b for a,b in k.items() if b > 0.04})
for c in d['configs']:
    print(c['config'][:40], c.get('sa_ms'), c.get('lcp_ms'), c.get('fused_sa_lcp',{}).get('ms'), c.get('bit_exact_vs_pins'), {k:v for k,v in list(c['top_kernels_ms'].items())[:6]})"
