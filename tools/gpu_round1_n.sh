#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider > gpurun_out/pytest_n.log 2>&1
echo "pytest all: $(tail -1 gpurun_out/pytest_n.log)"
grep -E "FAILED|Error|assert" gpurun_out/pytest_n.log | head -20
timeout 300 python tools/bench_prior.py 2>&1 | tail -12
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_n.log 2>&1
python - <<PY
import json
l=[x for x in open('gpurun_out/bench_n.log') if x.startswith('{')][-1]
d=json.loads(l); print('bench:', d['value'], d['ms_per_step'], d['roofline']['by_class_ms'], d['roofline']['achieved'])
PY
