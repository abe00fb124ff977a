#!/bin/bash
mkdir -p gpurun_out
{
echo "== sanity: K22_GN_FUSED=0 bench"
K22_GN_FUSED=0 timeout 600 python bench.py --steps 50 --warmup 10 --no-parity --no-e2e --no-cpu-baseline 2>&1 | tail -3 | cut -c1-300
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -v -k "groupnorm" 2>&1 | grep -v PASSED | tail -15
python tools/bench_gn.py 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_unet_gpu.py -x -q -k "golden or tiny" 2>&1 | tail -3
for m in 0 1 1 0; do
  echo "== K22_GN_FUSED=$m"
  K22_GN_FUSED=$m timeout 600 python bench.py --steps 50 --warmup 10 --no-parity --no-e2e --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['by_class_ms'])
"
done
} > gpurun_out/gn_fused.txt 2>&1
cat gpurun_out/gn_fused.txt
