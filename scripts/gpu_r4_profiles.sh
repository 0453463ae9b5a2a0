#!/bin/bash
# round-4 evidence in one call: PMC passes + kernel stats of the headline (gpu_pmc.sh), rocprofv3 kernel stats of config 3,
# PMC traffic of the three 1 GB builds (gpu_pmc_fullsize.sh).  Outputs under gpurun_out/{pmc,c3stats,pmc_full}: copy the
# summaries into profiles/ (r4_*).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash scripts/gpu_pmc.sh > gpurun_out/r4_pmc.log 2>&1; tail -3 gpurun_out/r4_pmc.log | cut -c1-300
bash scripts/gpu_c3_stats.sh > gpurun_out/r4_c3stats.log 2>&1; tail -2 gpurun_out/r4_c3stats.log | cut -c1-300
for kv in "c3 eng" "c5 utf8" "dup dup"; do set -- $kv; bash scripts/gpu_pmc_fullsize.sh $1 $2 > gpurun_out/r4_pmcfull_$1.log 2>&1; tail -1 gpurun_out/r4_pmcfull_$1.log | cut -c1-200; done
