#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/run2; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python scripts/gpu_microbench.py > $OUT/microbench.jsonl 2> $OUT/microbench.err; echo "microbench rc=$?"
cat $OUT/microbench.jsonl; tail -3 $OUT/microbench.err
# PMC on the chunked KPT16 variant
ROOT=$PWD
CMD="python $ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-verify"
export SFX_RADIX_SWEEP=0 SFX_RADIX_KPT=16 SFX_RADIX_RANK=1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/$OUT/fetch -o p -- $CMD > $ROOT/$OUT/fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ROOT/$OUT/write -o p -- $CMD > $ROOT/$OUT/write.log 2>&1; echo "write rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --output-format csv -d $ROOT/$OUT/sq -o p -- $CMD > $ROOT/$OUT/sq.log 2>&1; echo "sq rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum --output-format csv -d $ROOT/$OUT/tcc -o p -- $CMD > $ROOT/$OUT/tcc.log 2>&1; echo "tcc rc=$?"
cd $ROOT
python scripts/pmc_summary.py $OUT/fetch $OUT/write $OUT/sq $OUT/tcc > $OUT/pmc_summary.csv
find $OUT -name "*.csv" -size +5M -delete
grep -E "radix_pass|groups|hist_all" $OUT/pmc_summary.csv
