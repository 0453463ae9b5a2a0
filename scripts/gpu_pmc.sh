#!/bin/bash
# PMC passes (separate runs, --kernel-trace only -- never combined with other trace domains)
# over one build of the headline workload, plus the rocprofv3 --stats kernel summary.
# Outputs under gpurun_out/pmc/; copy pmc_summary.csv, pmc_latest.json and kernel_stats.csv
# into profiles/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD; OUT=$ROOT/gpurun_out/pmc; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-verify --no-microbench --calibrate --configs ''"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $CMD > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $CMD > $OUT/write.log 2>&1; echo "write rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --output-format csv -d $OUT/sq -o p -- $CMD > $OUT/sq.log 2>&1; echo "sq rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -o p -- $CMD > $OUT/tcc.log 2>&1; echo "tcc rc=$?"
# kernel-trace statistics of the bench command itself (no counters)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-verify --no-microbench --configs '' > $OUT/stats.log 2>&1; echo "stats rc=$?"
cd $ROOT
SFX_COMMIT=$(cat suffix_amd/_build_commit.txt 2>/dev/null || echo unknown) python scripts/pmc_summary.py --json $OUT/pmc_latest.json $OUT/fetch $OUT/write $OUT/sq $OUT/tcc > $OUT/pmc_summary.csv
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv && head -20 $OUT/kernel_stats.csv
find $OUT -name "*.csv" -size +5M -delete
cat $OUT/pmc_latest.json
