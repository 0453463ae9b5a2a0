#!/bin/bash
# Rehearsal of the N>1 bench path on a 1-GPU box: every rank on cuda:0, gloo instead of RCCL
# (RCCL refuses two ranks on one device).  Exercises bench.py's multi-rank code and
# suffix_amd/dist.py on device memory; the timing is NOT a multi-GPU number.
#   gpurun --timeout 600 -- 'bash scripts/gpu_rehearse_ranks.sh [ranks] [bytes per rank]'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N="${1:-2}"
SIZE="${2:-100000000}"
export SFX_BENCH_SHARE_GPU=1
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 \
    --master-port 29511 bench.py --gpus "$N" --steps 3 --warmup 1 --size "$SIZE" 2>&1 | grep -E '^\{|Error|error|Traceback' | tail -5
