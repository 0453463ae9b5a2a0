#!/bin/bash
# round 2, call A: parity suite + full-size configs (new generators, round-1 inputs, high-LCP) + bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r2a
(time timeout 900 python -m pytest tests -m gpu -x -q) > gpurun_out/r2a/pytest.log 2>&1
rm -f gpurun_out/big/results.jsonl
(time timeout 1200 python tests/fullsize_configs.py c3 c5 dup dna1g c3r1 c5r1) > gpurun_out/r2a/fullsize.log 2>&1
cp gpurun_out/big/results.jsonl gpurun_out/r2a/fullsize.jsonl
(time timeout 300 python bench.py) > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
tail -3 gpurun_out/r2a/pytest.log
grep -h "sa_ms" gpurun_out/r2a/fullsize.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(d['config'][:40], d['sa_ms'], d.get('lcp_ms'), d['verified'], d['build']['rounds'], d['build'].get('text_rounds'), d['build'].get('rank_rounds'))
"
