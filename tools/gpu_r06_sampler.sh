#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -p no:cacheprovider -k "sampler or percentile or threshold" 2>&1 | tail -4
timeout 100 python tools/bench_sampler.py 2>&1 | tail -1
echo "[done t=$SECONDS s]"
