#!/bin/bash
# round-2 profiles: PMC passes + kernel stats of the headline bench, kernel stats of the config-3 build
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$PWD
bash scripts/gpu_pmc.sh > gpurun_out/pmc_run.log 2>&1
mkdir -p gpurun_out/r2i; export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r2i/c3 -o c3 -- python $ROOT/scripts/gpu_time_build.py eng > $ROOT/gpurun_out/r2i/c3.log 2>&1; echo "c3 stats rc=$?"
cd $ROOT
f=$(find gpurun_out/r2i/c3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r2i/c3_kernel_stats.csv && head -12 gpurun_out/r2i/c3_kernel_stats.csv
find gpurun_out/r2i -name "*.csv" -size +5M -delete
tail -3 gpurun_out/pmc_run.log | cut -c1-600
