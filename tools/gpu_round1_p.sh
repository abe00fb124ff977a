#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_prior_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider > gpurun_out/pytest_p.log 2>&1
echo "pytest kernels+prior: $(tail -1 gpurun_out/pytest_p.log)"
grep -E "FAILED|Error|assert" gpurun_out/pytest_p.log | head -20
timeout 300 python tools/bench_kernels.py --configs h256x1,g256x1,h128x1,g128x1,h256x2,g256x2,h128x2,g128x2,g256x4,g128x4,g256x8,g128x8 > gpurun_out/bench_kernels_p.log 2>&1
echo "== conv"; tail -40 gpurun_out/bench_kernels_p.log
timeout 300 python tools/bench_prior.py 2>&1 | grep -v "^ \|taps" | tail -3
