#!/bin/bash
# SQ counters of the headline's kernels (100 MB of DNA): SFX_DEV_LIB / SFX_TIE_ROUTE select the variant; $1 = output tag
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD; TAG=${1:-dna}; OUT=$ROOT/gpurun_out/sq_$TAG; mkdir -p $OUT; export TMPDIR=/tmp TIME_SHA=0
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/sq -o p -- python $ROOT/scripts/gpu_time_build.py dna 100000000 > $OUT/sq.log 2>&1; echo "sq rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq2 -o p -- python $ROOT/scripts/gpu_time_build.py dna 100000000 > $OUT/sq2.log 2>&1; echo "sq2 rc=$?"
cd $ROOT
python scripts/pmc_summary.py $OUT/sq $OUT/sq2 > $OUT/sq_summary.csv
find $OUT -name "*.csv" -size +5M -delete
python scripts/sq_table.py $OUT/sq_summary.csv
