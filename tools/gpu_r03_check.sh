#!/bin/bash
# round 3 checkpoint: full GPU suite (fp16 mode, race fix, scheduler variants) + the default bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03a}
mkdir -p gpurun_out
export TMPDIR=/tmp
export K22_PARITY_REPORT=$PWD/gpurun_out/parity_$TAG.json
timeout 1500 python -m pytest tests -m gpu -q -s -x -p no:cacheprovider --timeout 900 > gpurun_out/pytest_$TAG.log 2>&1
echo "pytest: $(grep -E ' passed| failed' gpurun_out/pytest_$TAG.log | tail -1)"
grep -E "^FAILED|^ERROR|^E  " gpurun_out/pytest_$TAG.log | head -30
grep -E "fp32:|bf16:|fp16:|bfloat16|float16|float32|max\|d\||drift|uint8" gpurun_out/pytest_$TAG.log | grep -v "^tests" | head -200 > gpurun_out/parity_lines_$TAG.txt
grep -E "fp16" gpurun_out/parity_lines_$TAG.txt | head -40
unset K22_PARITY_REPORT
timeout 900 python bench.py --steps 50 --warmup 5 --tuning-report gpurun_out/tuning_$TAG.txt > gpurun_out/bench_$TAG.log 2> gpurun_out/bench_$TAG.err; tail -1 gpurun_out/bench_$TAG.log | cut -c1-200
tail -1 gpurun_out/bench_$TAG.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('by_class', d['roofline']['by_class_ms'], 'frac', d['roofline']['frac'])
print('parity', json.dumps(d.get('parity_paths'))[:900])
print('e2e', json.dumps(d.get('e2e'))[:700])
print('cpu', d['cpu_baseline']['value'] if d.get('cpu_baseline') else None)
"
tail -5 gpurun_out/bench_$TAG.err
