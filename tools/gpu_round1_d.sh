#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 -p no:cacheprovider > gpurun_out/pytest_d.log 2>&1
echo "pytest: $(tail -1 gpurun_out/pytest_d.log)"
grep -E "FAILED|Error|assert" gpurun_out/pytest_d.log | head -20
for x in 1 0; do
  timeout 200 python tools/bench_kernels.py --stages 2 --xcd $x --configs auto,128x128,128x64,128x64x8,128x64x4,64x64x8 > gpurun_out/bench_kernels_d_xcd$x.log 2>&1
  echo "== conv xcd $x"; tail -32 gpurun_out/bench_kernels_d_xcd$x.log
done
timeout 200 python tools/bench_kernels.py --gemm --stages 2 --configs auto,128x128,128x64,64x64,128x64x2,64x64x2 > gpurun_out/bench_gemm_d.log 2>&1
echo "== gemm"; tail -20 gpurun_out/bench_gemm_d.log
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_d.log 2>&1; tail -1 gpurun_out/bench_d.log | cut -c1-1500
