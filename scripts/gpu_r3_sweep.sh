#!/bin/bash
# Round-3 development sweep: one full-size build per (text kind, SFX_* variant) through the development library
# (hooks compiled in).  usage: gpu_r3_sweep.sh OUTNAME "kind1 kind2" "VAR=a,VAR2=b" "..." ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; mkdir -p $OUT; shift
KINDS=$1; shift
export SFX_LIB=$PWD/suffix_amd/libsuffix_hip_dev.so
for kind in $KINDS; do
  for v in "$@"; do
    envs=$(echo "$v" | tr ',' ' ')
    [ "$v" = "-" ] && envs=""
    env $envs timeout 300 python scripts/gpu_time_build.py $kind >> $OUT/sweep.jsonl 2>> $OUT/sweep.err || echo "{\"kind\": \"$kind\", \"variant\": \"$v\", \"failed\": true}" >> $OUT/sweep.jsonl
  done
done
cat $OUT/sweep.jsonl | cut -c1-1200
tail -5 $OUT/sweep.err
