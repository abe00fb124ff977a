#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_prior_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "attention or prior" > gpurun_out/pytest_x.log 2>&1
echo "pytest: $(tail -1 gpurun_out/pytest_x.log)"
grep -E "FAILED|Error|assert" gpurun_out/pytest_x.log | head -10
python tools/bench_attn.py 20 2>&1 | grep "T="
