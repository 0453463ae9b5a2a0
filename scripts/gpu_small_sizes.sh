#!/bin/bash
# key-width choice on mid-size natural-language text (development library): gpu_small_sizes.sh OUT
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; mkdir -p $OUT
export SFX_DEV_LIB=$PWD/suffix_amd/libsuffix_hip_dev.so TIME_SHA=0
for n in 300000 1000000 4000000 8000000; do
  for v in "" "SFX_FORCE_KEY64=1 SFX_HT_MIN=1" "SFX_FORCE_KEY64=1 SFX_HT=0"; do
    env $v timeout 120 python scripts/gpu_time_build.py eng $n 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['n'], '$v', d['sa_ms'], d['stats']['key_bits'], d['stats']['rounds'], d['stats']['active_after_initial'])"
  done
done | tee $OUT/small_sizes.txt
