#!/bin/bash
# hybrid initial sort: timing of the headline (and 200 MB), parity of the hybrid tests
mkdir -p gpurun_out
for n in 100000000 200000000; do
  echo "== n=$n"
  timeout 200 python bench.py --size $n --steps 10 --warmup 2 --configs '' --cpu-sample 0 --no-microbench 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print(d['value'], d['ms_per_step'], d['verified'], d['lcp']['fused_sa_lcp']['ms_per_step'], k.get('bucket_sort_lds'), d['roofline']['frac'])"
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hybrid or dna_20mb or device_resident" 2>&1 | tail -3
