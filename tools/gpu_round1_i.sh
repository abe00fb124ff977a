#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider > gpurun_out/pytest_i.log 2>&1
echo "pytest: $(tail -1 gpurun_out/pytest_i.log)"
grep -E "FAILED|Error|assert" gpurun_out/pytest_i.log | head -20
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --tuning-report gpurun_out/tuning_i.txt > gpurun_out/bench_i.log 2>&1; tail -1 gpurun_out/bench_i.log | cut -c1-1800
timeout 300 python tools/cpu_threads_probe.py 2>&1 | tail -8
