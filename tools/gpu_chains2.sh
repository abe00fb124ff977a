#!/bin/bash
mkdir -p gpurun_out
for c in 2 1; do K22_CHAINS=$c timeout 900 python tools/chains_determinism.py bf16,fp16,fp32 2>&1 | grep DET | tee -a gpurun_out/chains2.txt; done
for c in 2 1 2 1; do K22_CHAINS=$c timeout 600 python tools/x3_check.py bf16 2>&1 | grep dtype | cut -c1-300 | sed "s/^/chains=$c /" | tee -a gpurun_out/chains2.txt; done
