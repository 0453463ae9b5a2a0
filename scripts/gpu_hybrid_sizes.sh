#!/bin/bash
# hybrid initial sort: headline timing with kernel split (and, with arguments, other sizes on / off)
mkdir -p gpurun_out
for n in ${@:-100000000}; do
  for v in "SFX_HYBRID=1" ${HYBRID_OFF:+"SFX_HYBRID=0"}; do
    echo "== n=$n $v"
    env $v timeout 200 python bench.py --size $n --steps 10 --warmup 2 --configs '' --cpu-sample 0 --no-microbench 2>/dev/null | python -c "
import json,sys
sys.path.insert(0,'scripts')
import _benchout
d=_benchout.legacy(*_benchout.load(sys.stdin)); k=d['roofline']['kernel_ms']
print(d['value'], d['ms_per_step'], d['verified'], d['lcp']['fused_sa_lcp']['ms_per_step'])
print({a:b for a,b in k.items() if b > 0.012})"
  done
done
