#!/bin/bash
# radix scatter geometry sweep on the headline workload
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/variants; mkdir -p $OUT
for v in 1 2 3 4; do
  SFX_RADIX_VARIANT=$v timeout 300 python bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-verify > $OUT/v$v.json 2> $OUT/v$v.err
  echo "variant $v rc=$?"; python - <<PY
import json,sys
sys.path.insert(0,"scripts")
import _benchout
d=_benchout.legacy(*_benchout.load(open("$OUT/v$v.json")))
print("  MB/s", d["value"], "ms", d["ms_per_step"], {k:v for k,v in d["roofline"]["kernel_ms"].items()})
PY
done
