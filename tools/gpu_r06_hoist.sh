#!/bin/bash
# time / FiLM rows of all steps hoisted to the loop start (K22_HOIST_TIME): parity tests, then same-box A/B/A/B of the bench step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
echo "[t=$SECONDS s]"
timeout 600 python -m pytest tests/test_full_size_gpu.py tests/test_pipeline_gpu.py tests/test_unet22_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
echo "[t=$SECONDS s]"
for rep in 1 2 3; do for h in 1 0; do
  v=$(K22_HOIST_TIME=$h timeout 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-parity --no-e2e --no-box --no-traffic --no-profile 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*')
  echo "hoist=$h rep $rep: $v"
done; done
for h in 1 0; do
  v=$(K22_HOIST_TIME=$h timeout 200 python bench.py --dtype f16x2 --steps 50 --warmup 5 --no-cpu-baseline --no-parity --no-e2e --no-box --no-traffic --no-profile 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*')
  echo "f16x2 hoist=$h: $v"
done
echo "[t=$SECONDS s]"
echo "== tiny-model tests without the tuner"; K22_AUTOTUNE=0 timeout 600 python -m pytest tests/test_unet_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5
echo "[done t=$SECONDS s]"
