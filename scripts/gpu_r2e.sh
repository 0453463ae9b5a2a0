#!/bin/bash
# parity suite incl. full-size tests + default bench (with the configs array)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r2e
(time timeout 1500 python -m pytest tests -m gpu -x -q --durations=8) > gpurun_out/r2e/pytest.log 2>&1
(time timeout 900 python bench.py) > gpurun_out/r2e/bench.json 2> gpurun_out/r2e/bench.err
tail -15 gpurun_out/r2e/pytest.log
python -c "
import json
d=json.loads(open('gpurun_out/r2e/bench.json').read())
print(d['value'], d['ms_per_step'], d['lcp'], d['verified'])
for c in d['configs']: print({k:c.get(k) for k in ('config','sa_ms','lcp_ms','sa_MB/s','bit_exact_vs_pins','pin_checks','error','queries','gen_s')})
"
tail -3 gpurun_out/r2e/bench.err
