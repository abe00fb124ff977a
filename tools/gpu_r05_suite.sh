#!/bin/bash
# full GPU suite + smoke only (validation pass)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export K22_PARITY_REPORT=$PWD/gpurun_out/parity_suite.json
timeout 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > gpurun_out/pytest_suite.log 2>&1
echo "pytest: $(grep -E ' passed| failed' gpurun_out/pytest_suite.log | tail -1)   [t=$SECONDS s]"
grep -E "^FAILED|^ERROR" gpurun_out/pytest_suite.log | head -30
grep -A18 "slowest" gpurun_out/pytest_suite.log
unset K22_PARITY_REPORT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -E "smoke|Error|assert"
