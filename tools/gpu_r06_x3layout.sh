#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_x3_gpu.py tests/test_x2_gpu.py tests/test_unet_gpu.py::test_plms_step_kernel_matches_the_oracle_arithmetic -m gpu -x -q -p no:cacheprovider 2>&1 | tail -8
echo "[t=$SECONDS s]"
timeout 600 python -m pytest tests/test_full_size_gpu.py -m gpu -x -q -p no:cacheprovider -k "split or asymmetric" 2>&1 | tail -8
echo "[t=$SECONDS s]"
for dt in f16x2 f16x3; do
  timeout 200 python bench.py --dtype $dt --chains 1 --steps 50 --warmup 5 --no-cpu-baseline --no-parity --no-e2e --no-box --no-traffic 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$dt', d['value'], d['roofline']['by_class_ms'])"
done
echo "[done t=$SECONDS s]"
