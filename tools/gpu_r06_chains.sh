#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_unet_gpu.py -m gpu -x -q -p no:cacheprovider -k "two_chains" 2>&1 | tail -8
echo "[t=$SECONDS s]"
for rep in 1 2 3; do for ch in 1 2; do
  v=$(K22_CHAINS=$ch timeout 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-parity --no-e2e --no-box --no-traffic --no-profile 2>gpurun_out/chains_err_$ch.log | tail -1 | grep -o '"value": [0-9.]*')
  echo "K22_CHAINS=$ch rep $rep: $v"
done; done
echo "[t=$SECONDS s]"
for ch in 1 2; do
  v=$(K22_CHAINS=$ch timeout 200 python bench.py --dtype f16x2 --steps 50 --warmup 5 --no-cpu-baseline --no-parity --no-e2e --no-box --no-traffic --no-profile 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*')
  echo "f16x2 K22_CHAINS=$ch: $v"
done
tail -3 gpurun_out/chains_err_2.log
echo "[done t=$SECONDS s]"
