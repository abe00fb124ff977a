#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 -p no:cacheprovider > gpurun_out/pytest_gpu_b.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu_b.log
tail -15 gpurun_out/pytest_gpu_b.log
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_b.log 2>&1; tail -1 gpurun_out/bench_b.log
timeout 300 python tools/bench_kernels.py --configs auto,128x128,128x64,64x64 > gpurun_out/bench_kernels_b.log 2>&1; tail -40 gpurun_out/bench_kernels_b.log
