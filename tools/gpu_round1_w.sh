#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_unet_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "rank_nonzero" -s > gpurun_out/pytest_w.log 2>&1
echo "pytest: $(tail -1 gpurun_out/pytest_w.log)"
grep -E "FAILED|Error|assert|DDIM" gpurun_out/pytest_w.log | head -20
