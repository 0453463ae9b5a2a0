#!/bin/bash
# hybrid initial sort: ranking variant of the LDS sort, then PMC passes + kernel stats, then the validation run
mkdir -p gpurun_out
for v in "SFX_HYBRID_RANK=1" "SFX_HYBRID_RANK=0"; do
  echo "== $v"
  env $v timeout 200 python bench.py --steps 10 --warmup 2 --configs '' --cpu-sample 0 --no-microbench 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print(d['value'], d['ms_per_step'], d['verified'], d['lcp']['fused_sa_lcp']['ms_per_step'], k.get('bucket_sort_lds'))"
done
