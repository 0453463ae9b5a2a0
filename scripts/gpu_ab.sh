#!/bin/bash
# A/B runs through the development library (hooks compiled in: `make -C suffix_amd/csrc dev`): one run per (text kind, variant).
#   gpu_ab.sh OUTNAME "kind1 kind2" "VAR=a,VAR2=b" "-" ...         full-size builds (scripts/gpu_time_build.py; "-" = no hook set)
#   AB_SCRIPT=scripts/gpu_lcp_prof.py gpu_ab.sh ...                 another per-kind script
#   AB_BENCH=50 gpu_ab.sh OUTNAME - "SFX_PARTITION_WAVES=8" ...     the headline bench at 50 steps instead (kinds ignored)
# -> gpurun_out/OUTNAME/ab.jsonl (+ ab.err)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; mkdir -p $OUT; shift
KINDS=$1; shift
export SFX_DEV_LIB=$PWD/suffix_amd/libsuffix_hip_dev.so TMPDIR=/tmp
SCRIPT=${AB_SCRIPT:-scripts/gpu_time_build.py}
for kind in $KINDS; do
  for v in "$@"; do
    envs=$(echo "$v" | tr ',' ' ')
    [ "$v" = "-" ] && envs=""
    if [ -n "${AB_BENCH:-}" ]; then
      env $envs timeout 300 python bench.py --dev-lib $SFX_DEV_LIB --steps $AB_BENCH --warmup 3 --configs "" --cpu-sample 0 --no-microbench 2>> $OUT/ab.err | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print(json.dumps({'variant': '$v', 'ms_per_step': r['ms_per_step'], 'MBps': r['value'], 'verified': r['verified'], 'kernels': r['roofline']['kernels'][:4]}))" >> $OUT/ab.jsonl
    else
      env $envs timeout 300 python $SCRIPT $kind >> $OUT/ab.jsonl 2>> $OUT/ab.err || echo "{\"kind\": \"$kind\", \"variant\": \"$v\", \"failed\": true}" >> $OUT/ab.jsonl
    fi
  done
done
cut -c1-1200 $OUT/ab.jsonl
tail -5 $OUT/ab.err
