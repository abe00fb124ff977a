#!/bin/bash
# round 2, call D: producer/consumer conv kernel (algo 11): unit parity, then per-shape timing against the lock-step variants
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider -k "(halo or partial_sums or fused_skip) and (11 or 12)" > gpurun_out/pytest_d.log 2>&1
echo "pytest algo 11+12: $(tail -1 gpurun_out/pytest_d.log)"
grep -E "^FAILED|^E  " gpurun_out/pytest_d.log | head -10
timeout 600 python tools/bench_kernels.py --reps 10 --configs p256,t256,u256,a128,t128 > gpurun_out/bench_kernels_d.log 2>&1
cat gpurun_out/bench_kernels_d.log | cut -c1-150
