#!/bin/bash
# hybrid initial sort: where it starts to pay (smaller texts), on / off
mkdir -p gpurun_out
for n in 20000000 34000000 50000000; do
  for v in "SFX_HYBRID_MIN=1000" "SFX_HYBRID=0"; do
    echo "== n=$n $v"
    env $v timeout 200 python bench.py --size $n --steps 10 --warmup 2 --configs '' --cpu-sample 0 --no-microbench 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print(d['value'], d['ms_per_step'], d['verified'], k.get('bucket_sort_lds'), k.get('radix_scatter_u32'))"
  done
done
