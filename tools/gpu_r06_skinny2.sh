#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_skinny_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5
timeout 200 python tools/bench_skinny.py --only 3,2 2>&1 | grep -v amdgpu.ids
python tools/skinny_trace.py 2>&1 | grep -v amdgpu.ids | grep -A7 "c_fc"
timeout 300 python -m pytest tests/test_prior_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5
for v in 1 2 3 4; do echo "== K22_PRIOR_QKV_SPLIT=$v"; K22_PRIOR_QKV_SPLIT=$v timeout 200 python tools/bench_prior.py 2>&1 | grep -E "^prior"; done
echo "[done t=$SECONDS s]"
