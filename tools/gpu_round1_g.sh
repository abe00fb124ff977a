#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider > gpurun_out/pytest_g.log 2>&1
echo "pytest: $(tail -1 gpurun_out/pytest_g.log)"
grep -E "FAILED|Error|assert" gpurun_out/pytest_g.log | head -20
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --tuning-report gpurun_out/tuning_g.txt > gpurun_out/bench_g.log 2>&1; tail -1 gpurun_out/bench_g.log | cut -c1-1500
cat gpurun_out/tuning_g.txt
bash tools/gpu_profile.sh r01_v4 10
