#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_prior_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider -s > gpurun_out/pytest_m.log 2>&1
echo "pytest prior: $(tail -1 gpurun_out/pytest_m.log)"
grep -E "FAILED|Error|assert|max\|d\||drift" gpurun_out/pytest_m.log | head -20
timeout 300 python tools/bench_prior.py 2>&1 | tail -3
