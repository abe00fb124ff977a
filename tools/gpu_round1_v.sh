#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "conv" > gpurun_out/pytest_v.log 2>&1
echo "pytest conv: $(tail -1 gpurun_out/pytest_v.log)"
grep -E "FAILED|Error|assert" gpurun_out/pytest_v.log | head -20
timeout 300 python tools/bench_kernels.py --configs h256x1,h128x1,h256x2,h128x2,h128x4,h256x4 > gpurun_out/bench_kernels_v.log 2>&1
echo "== conv"; tail -32 gpurun_out/bench_kernels_v.log
for s in 384,384,96 768,768,48; do timeout 120 python tools/conv_trace.py --shape $s 2>&1 | grep -v amdgpu.ids | grep -v "first 12"; done
