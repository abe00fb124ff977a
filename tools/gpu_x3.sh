#!/bin/bash
# round 4: first GPU pass of the split-precision engine
mkdir -p gpurun_out
export K22_TUNE_REPS=3
timeout 900 python -m pytest tests/test_x3_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/x3_pytest.txt
cat gpurun_out/x3_pytest.txt
timeout 900 python tools/x3_check.py f16x3 --tuning > gpurun_out/x3_check.txt 2>&1
tail -5 gpurun_out/x3_check.txt | cut -c1-1500
