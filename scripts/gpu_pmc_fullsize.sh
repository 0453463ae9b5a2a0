#!/bin/bash
# HBM traffic (PMC FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes with --kernel-trace only) of the kernels of a
# FULL-SIZE build, with two calibrations in the same process: the 1 GiB streaming copy (k_mb_copy) and 2^28 random
# 4-byte reads from a 4 GiB array (k_mb_gather<unsigned int>: what a key gather looks like to the memory system).
#   [PMC_TAG=r6] gpu_pmc_fullsize.sh c3 eng     -> gpurun_out/pmc_full/r6_pmc_fullsize.json (key "c3") + r6_pmc_summary_c3.csv:
#   copy BOTH to profiles/ (the JSON entry is recomputable from the CSV: tests/test_bench_logic.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD; OUT=$ROOT/gpurun_out/pmc_full; mkdir -p $OUT; export TMPDIR=/tmp
key=$1; kind=$2; TAG=${PMC_TAG:-r6}
CMD="python $ROOT/scripts/gpu_time_build.py $kind"
export TIME_SHA=0 PMC_CALIBRATE=1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$key -o p -- $CMD > $OUT/fetch_$key.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$key -o p -- $CMD > $OUT/write_$key.log 2>&1; echo "write rc=$?"
cd $ROOT
SFX_COMMIT=$(cat suffix_amd/_build_commit.txt 2>/dev/null || echo unknown) python scripts/pmc_summary.py --fullsize $OUT/${TAG}_pmc_fullsize.json $key 3 $OUT/fetch_$key $OUT/write_$key > $OUT/${TAG}_pmc_summary_$key.csv
find $OUT -name "*.csv" -size +5M -delete
python3 -c "import json;d=json.load(open('$OUT/${TAG}_pmc_fullsize.json'));print(json.dumps(d['$key'])[:3000])"
