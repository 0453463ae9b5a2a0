#!/bin/bash
# rocprofv3 kernel trace of one full-size build: per-launch durations of the top kernels.  gpu_r3_ktrace.sh OUT kind [ENV=..]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD; OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT; kind=$2; shift; shift
export SFX_LIB=$ROOT/suffix_amd/libsuffix_hip_dev.so TMPDIR=/tmp TIME_SHA=0
cd /tmp
env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$kind -o t -- python $ROOT/scripts/gpu_time_build.py $kind > $OUT/ktrace_$kind.log 2>&1
cd $ROOT
f=$(find $OUT/prof_$kind -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'P' > $OUT/ktrace_$kind.txt
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
# keep the last third (the profiled build is the 3rd of 3)
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
n=len(rows)//3
rows=rows[-n:]
t0=int(rows[0]["Start_Timestamp"])
for r in rows:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
    if d>=0.3: print(f'{(int(r["Start_Timestamp"])-t0)/1e6:9.2f} ms  +{d:7.3f} ms  {r["Kernel_Name"][:70]}  grid={r.get("Grid_Size","?")}')
P
cat $OUT/ktrace_$kind.txt | head -80
s=$(find $OUT/prof_$kind -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp $s $OUT/kernel_stats_$kind.csv
find $OUT/prof_$kind -name "*.csv" -size +2M -delete
