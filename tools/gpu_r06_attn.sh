#!/bin/bash
# round 6: attention kernel A/B (kernel tests, tools/bench_attn.py, step bench with K22_ATT_PIPE=0 / 1)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_x2_gpu.py tests/test_x3_gpu.py tests/test_prior_gpu.py tests/test_encoders_gpu.py -m gpu -x -q -p no:cacheprovider -k "attention or attn or prior or tower or encoder" 2>&1 | tail -8
echo "[t=$SECONDS s]"
for p in 0 1; do K22_ATT_PIPE=$p timeout 120 python tools/bench_attn.py 50 2>&1 | tail -3; done
for rep in 1 2; do for p in 0 1; do
  v=$(K22_ATT_PIPE=$p timeout 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-parity --no-e2e --no-box --no-traffic --no-profile 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*')
  echo "K22_ATT_PIPE=$p rep $rep: $v"
done; done
echo "[done t=$SECONDS s]"
