#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 300 -p no:cacheprovider > gpurun_out/pytest_f.log 2>&1
echo "pytest kernels: $(tail -1 gpurun_out/pytest_f.log)"
grep -E "FAILED|Error|assert" gpurun_out/pytest_f.log | head -20
timeout 300 python tools/bench_kernels.py --configs auto,128x64x8,64x64x8,h256x1,h256x2,h256x4,h256x8,h128x1,h128x2,h128x4,h128x8 > gpurun_out/bench_kernels_f.log 2>&1
echo "== conv"; tail -36 gpurun_out/bench_kernels_f.log
timeout 900 python -m pytest tests/test_unet_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider > gpurun_out/pytest_f2.log 2>&1
echo "pytest unet: $(tail -1 gpurun_out/pytest_f2.log)"
grep -E "FAILED|Error|assert" gpurun_out/pytest_f2.log | head -20
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --tuning-report gpurun_out/tuning_f.txt > gpurun_out/bench_f.log 2>&1; tail -1 gpurun_out/bench_f.log | cut -c1-1500
cat gpurun_out/tuning_f.txt
K22_AUTOTUNE=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_f_noat.log 2>&1; tail -1 gpurun_out/bench_f_noat.log | cut -c1-1200
