#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider > gpurun_out/pytest_u.log 2>&1
echo "pytest all: $(tail -1 gpurun_out/pytest_u.log)"
grep -E "FAILED|Error|assert" gpurun_out/pytest_u.log | head -20
export K22_TUNE_CACHE=$PWD/gpurun_out/tune_cache_u.txt
rm -f $K22_TUNE_CACHE
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_u.log 2>&1
python - <<PY
import json
l=[x for x in open('gpurun_out/bench_u.log') if x.startswith('{')][-1]
d=json.loads(l); print('bench C2:', d['value'], d['ms_per_step'], d['roofline']['by_class_ms'], d['roofline']['achieved'])
PY
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-profile 2>&1 | tail -1 | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 300 python tools/bench_prior.py 2>&1 | grep -v "^ \|taps" | tail -3
