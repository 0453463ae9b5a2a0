#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r2b
rm -f gpurun_out/big/results.jsonl
(time timeout 900 python tests/fullsize_configs.py c3 c5 dup c3r1 c5r1) > gpurun_out/r2b/fullsize.log 2>&1
cp gpurun_out/big/results.jsonl gpurun_out/r2b/fullsize.jsonl
rm -f gpurun_out/big/results.jsonl
(time SFX_SWITCH=text timeout 600 python tests/fullsize_configs.py c3 c5) > gpurun_out/r2b/fullsize_text.log 2>&1
cp gpurun_out/big/results.jsonl gpurun_out/r2b/fullsize_text.jsonl
rm -f gpurun_out/big/results.jsonl
(time SFX_SWITCH=rank timeout 600 python tests/fullsize_configs.py c3 c3r1) > gpurun_out/r2b/fullsize_rank.log 2>&1
cp gpurun_out/big/results.jsonl gpurun_out/r2b/fullsize_rank.jsonl
for f in fullsize fullsize_text fullsize_rank; do echo "== $f"; python -c "
import sys, json
for l in open('gpurun_out/r2b/$f.jsonl'):
    d = json.loads(l)
    print(d['config'][:40], d['sa_ms'], d.get('lcp_ms'), d['verified'], d['build']['rounds'], d['build'].get('text_rounds'), d['build'].get('rank_rounds'))
"; done
