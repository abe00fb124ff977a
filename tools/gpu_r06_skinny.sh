#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_skinny_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15
echo "[t=$SECONDS s]"
timeout 200 python tools/bench_skinny.py > gpurun_out/r06_skinny.txt 2>&1; cat gpurun_out/r06_skinny.txt
echo "[t=$SECONDS s]"
timeout 300 python -m pytest tests/test_prior_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15
for v in 1 0; do echo "== K22_PRIOR_SKINNY=$v"; K22_PRIOR_SKINNY=$v timeout 200 python tools/bench_prior.py 2>&1 | grep -E "^prior"; done
echo "[done t=$SECONDS s]"
