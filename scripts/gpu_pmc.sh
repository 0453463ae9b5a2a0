#!/bin/bash
# PMC passes (separate runs, --kernel-trace only) on one build of the headline workload
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD; OUT=$ROOT/gpurun_out/pmc; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-verify"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $CMD > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $CMD > $OUT/write.log 2>&1; echo "write rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --output-format csv -d $OUT/sq -o p -- $CMD > $OUT/sq.log 2>&1; echo "sq rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -o p -- $CMD > $OUT/tcc.log 2>&1; echo "tcc rc=$?"
cd $ROOT
python scripts/pmc_summary.py $OUT/fetch $OUT/write $OUT/sq $OUT/tcc > $OUT/pmc_summary.csv
find $OUT -name "*.csv" -size +5M -delete
cat $OUT/pmc_summary.csv | head -150
