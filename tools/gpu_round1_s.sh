#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "conv or gemm" > gpurun_out/pytest_s.log 2>&1
echo "pytest conv+gemm: $(tail -1 gpurun_out/pytest_s.log)"
grep -E "FAILED|Error|assert" gpurun_out/pytest_s.log | head -20
timeout 300 python tools/bench_kernels.py --configs h256x1,k256x1,h128x1,k128x1,h256x2,k256x2,h128x2,k128x2,h128x4,k128x4,k256x4,k256x8 > gpurun_out/bench_kernels_s.log 2>&1
echo "== conv"; tail -40 gpurun_out/bench_kernels_s.log
