#!/bin/bash
# A/B of two builds of the library on one box (K22_LIB_PATH): small-GEMM timings, gemm8 tests, bench step.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
B=$PWD/kandinsky-2_amd/libk22hip_base.so
echo "== bench_gemm_k NEW"; timeout 300 python tools/bench_gemm_k.py 2>&1 | tail -40
echo "== bench_gemm_k BASE"; K22_LIB_PATH=$B timeout 300 python tools/bench_gemm_k.py 2>&1 | tail -40
echo "[t=$SECONDS s]"
echo "== gemm8 tests NEW"; timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_x2_gpu.py tests/test_x3_gpu.py -m gpu -q -x -p no:cacheprovider -k "gemm" 2>&1 | tail -5
echo "[t=$SECONDS s]"
for rep in 1 2 3; do for lib in libk22hip.so libk22hip_base.so; do
  v=$(K22_LIB_PATH=$PWD/kandinsky-2_amd/$lib timeout 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-parity --no-e2e --no-box --no-traffic --no-profile 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*')
  echo "$lib rep $rep: $v"
done; done
echo "[done t=$SECONDS s]"
