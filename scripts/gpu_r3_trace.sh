#!/bin/bash
# per-round trace of one full-size build (development library): gpu_r3_trace.sh OUT kind [ENV=..]...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; mkdir -p $OUT; kind=$2; shift; shift
export SFX_LIB=$PWD/suffix_amd/libsuffix_hip_dev.so
env "$@" SFX_TRACE=1 TIME_SHA=0 timeout 300 python scripts/gpu_time_build.py $kind > $OUT/trace_$kind.json 2> $OUT/trace_$kind.err
grep "^round" $OUT/trace_$kind.err | tail -12
