#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-y}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider > gpurun_out/pytest_$TAG.log 2>&1
echo "pytest all: $(tail -1 gpurun_out/pytest_$TAG.log)"
grep -E "FAILED|Error|assert" gpurun_out/pytest_$TAG.log | head -20
export K22_TUNE_CACHE=$PWD/gpurun_out/tune_cache_$TAG.txt
rm -f $K22_TUNE_CACHE
timeout 900 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --tuning-report gpurun_out/tuning_$TAG.txt > gpurun_out/bench_$TAG.log 2>&1
tail -1 gpurun_out/bench_$TAG.log | cut -c1-1800
grep -E "^ +1 " gpurun_out/tuning_$TAG.txt
