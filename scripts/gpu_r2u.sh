#!/bin/bash
# query probe at the two occupancies of the phase-1 kernel (index only)
mkdir -p gpurun_out
export PROBE_INDEX_ONLY=1
for occ in 6 8; do
  echo "== SFX_QUERY_OCC=$occ"
  SFX_QUERY_OCC=$occ timeout 900 python scripts/gpu_query_probe.py 1000000000 2>/dev/null | cut -c1-400 | tee gpurun_out/r2u_occ$occ.jsonl
done
