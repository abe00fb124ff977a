#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
export K22_TUNE_CACHE=$PWD/gpurun_out/tune_cache_j.txt
rm -f $K22_TUNE_CACHE
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -6
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider > gpurun_out/pytest_j.log 2>&1
echo "pytest: $(tail -1 gpurun_out/pytest_j.log)"
grep -E "FAILED|Error|assert" gpurun_out/pytest_j.log | head -20
for i in 1 2 3; do
  timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --tuning-report gpurun_out/tuning_j.txt > gpurun_out/bench_j$i.log 2>&1
  python - <<PY
import json
l=[x for x in open('gpurun_out/bench_j$i.log') if x.startswith('{')][-1]
d=json.loads(l); print('run $i fused-skip:', d['value'], d['ms_per_step'], d['roofline']['by_class_ms'])
PY
done
K22_FUSE_SKIP=0 K22_TUNE_CACHE=$PWD/gpurun_out/tune_cache_j2.txt timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_j4.log 2>&1
python - <<PY
import json
l=[x for x in open('gpurun_out/bench_j4.log') if x.startswith('{')][-1]
d=json.loads(l); print('run 4 unfused-skip:', d['value'], d['ms_per_step'], d['roofline']['by_class_ms'])
PY
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -6
