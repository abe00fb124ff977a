#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider > gpurun_out/pytest_t.log 2>&1
echo "pytest all: $(tail -1 gpurun_out/pytest_t.log)"
grep -E "FAILED|Error|assert" gpurun_out/pytest_t.log | head -20
export K22_TUNE_CACHE=$PWD/gpurun_out/tune_cache_t.txt
rm -f $K22_TUNE_CACHE
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --tuning-report gpurun_out/tuning_t.txt > gpurun_out/bench_t.log 2>&1
python - <<PY
import json
l=[x for x in open('gpurun_out/bench_t.log') if x.startswith('{')][-1]
d=json.loads(l); print('bench C2:', d['value'], d['ms_per_step'], d['roofline']['by_class_ms'], d['roofline']['achieved'])
PY
bash tools/gpu_profile.sh r01_v7 10
