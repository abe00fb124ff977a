#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_movq_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider > gpurun_out/pytest_l.log 2>&1
echo "pytest movq: $(tail -1 gpurun_out/pytest_l.log)"
grep -E "FAILED|Error|assert|max\|d\|" gpurun_out/pytest_l.log | head -20
timeout 300 python tools/bench_movq.py --size 768 2>&1 | tail -2
timeout 300 python tools/bench_movq.py --size 1024 --bs 4 2>&1 | tail -2
