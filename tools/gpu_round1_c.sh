#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for st in 2 3 4; do
  K22_IGEMM_STAGES=$st timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 120 -p no:cacheprovider -k "gemm or conv" > gpurun_out/pytest_glds_$st.log 2>&1
  echo "stages=$st: $(tail -1 gpurun_out/pytest_glds_$st.log)"
done
for st in 0 2 3 4; do
  timeout 200 python tools/bench_kernels.py --stages $st --configs auto,128x128,128x64,128x64x8,128x64x16,64x64x16 > gpurun_out/bench_kernels_c_$st.log 2>&1
  echo "== stages $st"; tail -30 gpurun_out/bench_kernels_c_$st.log
done
for st in 2 3; do
  K22_IGEMM_STAGES=$st timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_c_$st.log 2>&1; tail -1 gpurun_out/bench_c_$st.log | cut -c1-1400
done
