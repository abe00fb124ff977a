#!/bin/bash
# A/B of the 16-bit MoVQ decode with SiLU from the native exp2 / rcp (shipped) vs the IEEE form everywhere (libk22hip_ieee.so:
# elementwise.hip + movq_kernels.hip built with -DK22_SILU_IEEE_EVERYWHERE and linked with the other objects into kandinsky-2_amd/libk22hip_ieee.so;
# at the time of the measurement the shipped build used silu_fast on the 16-bit paths - it no longer does, see common.h)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
P=kandinsky-2_amd
for v in fast ieee; do
  if [ $v = ieee ]; then cp $P/libk22hip.so /tmp/keep.so; cp $P/libk22hip_ieee.so $P/libk22hip.so; fi
  timeout 900 python -m pytest tests/test_movq_gpu.py -m gpu -q -s -p no:cacheprovider -k "golden" 2>&1 | grep -E "movq_|passed|failed" | grep -v "^tests" > gpurun_out/silu_ab_$v.txt
  echo "== $v"; cat gpurun_out/silu_ab_$v.txt | cut -c1-200
done
cp /tmp/keep.so $P/libk22hip.so
timeout 600 python -m pytest tests/test_x3_gpu.py -m gpu -q -p no:cacheprovider -k groupnorm 2>&1 | tail -3
