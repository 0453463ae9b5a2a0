#!/bin/bash
# rocprofv3 --kernel-trace --stats over config 3's build (1 GB English-like text, SA only, 3 builds)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD; OUT=$ROOT/gpurun_out/c3stats; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o c3 -- python $ROOT/scripts/gpu_time_build.py eng 1000000000 > $OUT/run.log 2>&1; echo "rc=$?"
cd $ROOT
f=$(find $OUT -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv && head -14 $OUT/kernel_stats.csv | cut -c1-200
tail -3 $OUT/run.log
find $OUT -name "*.csv" -size +3M -delete
