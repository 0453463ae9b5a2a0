#!/bin/bash
# A round's profiler evidence in one call: PMC passes + kernel stats of the headline (gpu_pmc.sh), rocprofv3 kernel stats of config 3
# (gpu_c3_stats.sh), PMC traffic of the five 1 GB builds (gpu_pmc_fullsize.sh).  Outputs under gpurun_out/{pmc,c3stats,pmc_full}:
# copy the summaries into profiles/ (rN_*) and pmc_latest.json.    gpu_profiles.sh [TAG=r6] [configs="c3 c5 dup c3r1 c5r1"]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r6}; CFGS=${2:-"c3 c5 dup c3r1 c5r1"}
bash scripts/gpu_pmc.sh > gpurun_out/${TAG}_pmc.log 2>&1; tail -3 gpurun_out/${TAG}_pmc.log | cut -c1-300
bash scripts/gpu_c3_stats.sh > gpurun_out/${TAG}_c3stats.log 2>&1; tail -2 gpurun_out/${TAG}_c3stats.log | cut -c1-300
declare -A KIND=([c3]=eng [c5]=utf8 [dup]=dup [c3r1]=engr1 [c5r1]=utf8r1)
for c in $CFGS; do PMC_TAG=$TAG bash scripts/gpu_pmc_fullsize.sh $c ${KIND[$c]} > gpurun_out/${TAG}_pmcfull_$c.log 2>&1; tail -1 gpurun_out/${TAG}_pmcfull_$c.log | cut -c1-200; done
