#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 300 -p no:cacheprovider > gpurun_out/pytest_e.log 2>&1
echo "pytest kernels: $(tail -1 gpurun_out/pytest_e.log)"
grep -E "FAILED|Error|assert" gpurun_out/pytest_e.log | head -20
timeout 300 python tools/bench_kernels.py --stages 2 --configs auto,128x128,128x64x4,h256x1,h256x2,h256x4,h256x8,h128x1,h128x2,h128x4 > gpurun_out/bench_kernels_e.log 2>&1
echo "== conv"; tail -34 gpurun_out/bench_kernels_e.log
timeout 600 python -m pytest tests/test_unet_gpu.py -m gpu -q -x --timeout 300 -p no:cacheprovider > gpurun_out/pytest_e2.log 2>&1
echo "pytest unet: $(tail -1 gpurun_out/pytest_e2.log)"
grep -E "FAILED|Error|assert" gpurun_out/pytest_e2.log | head -20
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_e.log 2>&1; tail -1 gpurun_out/bench_e.log | cut -c1-1500
