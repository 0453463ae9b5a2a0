#!/bin/bash
# One full-size build through the development library, traced:
#   gpu_trace.sh rounds  OUT kind [ENV=..]...   one line per refinement round (SFX_TRACE=1: members, kept, rank changes)
#   gpu_trace.sh kernels OUT kind [ENV=..]...   rocprofv3 kernel trace: start and duration of every launch >= 0.3 ms, kernel stats
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mode=$1; ROOT=$PWD; OUT=$ROOT/gpurun_out/$2; mkdir -p $OUT; kind=$3; shift; shift; shift
export SFX_DEV_LIB=$ROOT/suffix_amd/libsuffix_hip_dev.so TMPDIR=/tmp TIME_SHA=0
if [ "$mode" = rounds ]; then
  env "$@" SFX_TRACE=1 timeout 300 python scripts/gpu_time_build.py $kind > $OUT/trace_$kind.json 2> $OUT/trace_$kind.err
  grep "^round" $OUT/trace_$kind.err | tail -12
  exit 0
fi
cd /tmp
env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$kind -o t -- python $ROOT/scripts/gpu_time_build.py $kind > $OUT/ktrace_$kind.log 2>&1
cd $ROOT
f=$(find $OUT/prof_$kind -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'P' > $OUT/ktrace_$kind.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-(len(rows) // 3):]            # the profiled build is the last of three
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d >= 0.3:
        print(f'{(int(r["Start_Timestamp"]) - t0) / 1e6:9.2f} ms  +{d:7.3f} ms  {r["Kernel_Name"][:70]}  grid={r.get("Grid_Size", "?")}')
P
head -80 $OUT/ktrace_$kind.txt
s=$(find $OUT/prof_$kind -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp $s $OUT/kernel_stats_$kind.csv
find $OUT/prof_$kind -name "*.csv" -size +2M -delete
