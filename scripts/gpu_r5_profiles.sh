#!/bin/bash
# round-5 evidence in one call: PMC passes + kernel stats of the headline (gpu_pmc.sh), rocprofv3 kernel stats of config 3,
# PMC traffic of the three 1 GB builds (gpu_pmc_fullsize.sh), then a randomized parity run (gpu_fuzz.py).  Outputs under
# gpurun_out/{pmc,c3stats,pmc_full}: the summaries are copied into profiles/ (r5_*).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash scripts/gpu_pmc.sh > gpurun_out/r5_pmc.log 2>&1; tail -3 gpurun_out/r5_pmc.log | cut -c1-300
bash scripts/gpu_c3_stats.sh > gpurun_out/r5_c3stats.log 2>&1; tail -2 gpurun_out/r5_c3stats.log | cut -c1-300
for kv in "c3 eng" "c5 utf8" "dup dup"; do set -- $kv; bash scripts/gpu_pmc_fullsize.sh $1 $2 > gpurun_out/r5_pmcfull_$1.log 2>&1; tail -1 gpurun_out/r5_pmcfull_$1.log | cut -c1-200; done
timeout 600 python scripts/gpu_fuzz.py 150 3000000 505 > gpurun_out/r5_fuzz_large.log 2>&1; tail -2 gpurun_out/r5_fuzz_large.log
timeout 300 python scripts/gpu_fuzz.py 300 20000 506 40 > gpurun_out/r5_fuzz_small.log 2>&1; tail -2 gpurun_out/r5_fuzz_small.log
