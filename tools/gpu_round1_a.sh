#!/bin/bash
# First GPU pass: kernel + end-to-end parity, smoke, short bench, rocprof kernel stats.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
nproc >> gpurun_out/gpu.txt
timeout 840 python -m pytest tests -m gpu -q -rA --timeout 300 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -60 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -5 gpurun_out/smoke.log
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_a.log 2>&1; echo "bench exit $?" >> gpurun_out/bench_a.log; tail -3 gpurun_out/bench_a.log
timeout 240 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-graph > gpurun_out/bench_nograph.log 2>&1; tail -2 gpurun_out/bench_nograph.log
