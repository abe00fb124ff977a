#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "from_producer_group_sums" > gpurun_out/pytest_x.log 2>&1
echo "pytest: $(tail -1 gpurun_out/pytest_x.log)"
cp tools/tune_cache_dev.txt /tmp/tc.txt
export K22_TUNE_CACHE=/tmp/tc.txt
for v in 1 0 1 0; do
K22_GN_ONEPASS=$v timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/bench_x$v.log 2>&1
tail -1 gpurun_out/bench_x$v.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('onepass=$v', d['value'], d['ms_per_step'], d['roofline']['by_class_ms'])"
done
