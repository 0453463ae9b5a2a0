#!/bin/bash
# round-2 validation: GPU parity suite, default bench (headline + configs array), virtual-rank profile
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/validate
(time timeout 1500 python -m pytest tests -m gpu -x -q --durations=6) > gpurun_out/validate/pytest.log 2>&1
tail -14 gpurun_out/validate/pytest.log
(time timeout 900 python bench.py) > gpurun_out/validate/bench.json 2> gpurun_out/validate/bench.err
tail -4 gpurun_out/validate/bench.err
timeout 600 python scripts/gpu_virtual_ranks.py > gpurun_out/validate/virtual_ranks.jsonl 2>> gpurun_out/validate/err.log
timeout 300 python scripts/gpu_host_path.py > gpurun_out/validate/host_path.log 2>&1
python -c "
import json,sys
sys.path.insert(0,'scripts')
import _benchout
d=_benchout.legacy(*_benchout.load(open('gpurun_out/validate/bench.json')))
print(d['value'], d['ms_per_step'], d['lcp'], d['verified'], d['roofline']['frac'], d['roofline'].get('traffic_commit'), d['roofline'].get('this_commit'))
for c in d['configs']: print({k:c.get(k) for k in ('config','sa_ms','lcp_ms','sa_MB/s','fused_sa_lcp','bit_exact_vs_pins','error','queries')})
"
cat gpurun_out/validate/virtual_ranks.jsonl | cut -c1-400; cat gpurun_out/validate/host_path.log | tail -5
