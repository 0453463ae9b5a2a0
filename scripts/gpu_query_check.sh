#!/bin/bash
# 16-byte keys in the B+tree of the resident index + two-phase batches: parity of the query paths, the config-5 probe,
# then config 5 at full size against the oracle (all 10^6 queries)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "directory or batched" 2>&1 | tail -3
timeout 900 python scripts/gpu_query_probe.py 1000000000 > gpurun_out/r2t_query_probe.jsonl 2> gpurun_out/r2t_query_probe.err
cat gpurun_out/r2t_query_probe.jsonl | cut -c1-600; tail -3 gpurun_out/r2t_query_probe.err
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config5" 2>&1 | tail -3
