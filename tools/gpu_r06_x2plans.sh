#!/bin/bash
# round 6: the f16x2 precision plans re-priced on the 8-group chunk kernels: steps/s + C2 50-step final latent per plan
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for plan in 0 1 2 3; do
  K22_X2_PLAN=$plan timeout 300 python bench.py --dtype f16x2 --steps 50 --warmup 5 --no-cpu-baseline --no-e2e --no-box --no-traffic --parity-timed-only > gpurun_out/x2plan_$plan.log 2>&1
  tail -1 gpurun_out/x2plan_$plan.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=(d.get('parity_paths') or {}).get('f16x2') or {}
print('K22_X2_PLAN=$plan: %.2f steps/s (graph replay), final latent max-abs %s rms %s, conv3x3 %.2f ms' % (d['value'], p.get('final_latent_max_abs'), p.get('final_latent_rms'), d['roofline']['by_class_ms']['conv3x3']))"
done
K22_CHAINS=2 K22_X2_PLAN=0 timeout 300 python bench.py --dtype f16x2 --chains 2 --steps 50 --warmup 5 --no-cpu-baseline --no-e2e --no-box --no-traffic --no-parity --no-profile 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*'
echo "[done t=$SECONDS s]"
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_x2_gpu.py tests/test_unet_gpu.py -m gpu -x -q -p no:cacheprovider -k "attention or attn or tiny" 2>&1 | tail -3
echo "[tests done t=$SECONDS s]"
