#!/bin/bash
# round 4: fused GroupNorm-apply in the specialised conv kernel - parity, then step time with / without
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gn_fused_conv_gpu.py -x -q 2>&1 | tail -12 > gpurun_out/gnf_pytest.txt
cat gpurun_out/gnf_pytest.txt
for f in 1 0; do
  K22_FUSE_GN=$f timeout 600 python tools/x3_check.py bf16 2>&1 | grep dtype | cut -c1-700 | sed "s/^/fuse_gn=$f /" | tee -a gpurun_out/gnf_ab.txt
done
for f in 1 0; do
  K22_FUSE_GN=$f timeout 600 python tools/x3_check.py f16x3 2>&1 | grep dtype | cut -c1-700 | sed "s/^/fuse_gn=$f /" | tee -a gpurun_out/gnf_ab.txt
done
