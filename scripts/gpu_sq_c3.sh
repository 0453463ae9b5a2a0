#!/bin/bash
# SQ counters of config 3's kernels (what are the deep rounds waiting for?): one rocprofv3 --pmc pass over a 1 GB build
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD; OUT=$ROOT/gpurun_out/sq_c3; mkdir -p $OUT; export TMPDIR=/tmp TIME_SHA=0
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/sq -o p -- python $ROOT/scripts/gpu_time_build.py ${1:-eng} > $OUT/sq.log 2>&1; echo "sq rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq2 -o p -- python $ROOT/scripts/gpu_time_build.py ${1:-eng} > $OUT/sq2.log 2>&1; echo "sq2 rc=$?"
cd $ROOT
python scripts/pmc_summary.py $OUT/sq $OUT/sq2 > $OUT/sq_summary.csv
find $OUT -name "*.csv" -size +5M -delete
python scripts/sq_table.py $OUT/sq_summary.csv
