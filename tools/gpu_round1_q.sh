#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "conv" > gpurun_out/pytest_q.log 2>&1
echo "pytest conv: $(tail -1 gpurun_out/pytest_q.log)"
grep -E "FAILED|Error|assert" gpurun_out/pytest_q.log | head -20
timeout 300 python tools/bench_kernels.py --configs h256x1,h128x1,h256x2,h128x2,h128x4,g128x4,g256x8 > gpurun_out/bench_kernels_q.log 2>&1
echo "== conv"; tail -34 gpurun_out/bench_kernels_q.log
for i in 1 2; do
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --tuning-report gpurun_out/tuning_q.txt > gpurun_out/bench_q$i.log 2>&1
python - <<PY
import json
l=[x for x in open('gpurun_out/bench_q$i.log') if x.startswith('{')][-1]
d=json.loads(l); print('bench C2:', d['value'], d['ms_per_step'], d['roofline']['by_class_ms'], d['roofline']['achieved'])
PY
done
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --size 1024 --bs 4 > gpurun_out/bench_q_c3.log 2>&1
python - <<PY
import json
l=[x for x in open('gpurun_out/bench_q_c3.log') if x.startswith('{')][-1]
d=json.loads(l); print('bench C3 per-GPU shape (1024^2 bs4):', d['value'], d['ms_per_step'], d['images_per_sec'], d['roofline']['by_class_ms'], d['roofline']['achieved'])
PY
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --size 768 --bs 4 --inpaint > gpurun_out/bench_q_c4.log 2>&1
python - <<PY
import json
l=[x for x in open('gpurun_out/bench_q_c4.log') if x.startswith('{')][-1]
d=json.loads(l); print('bench C4 (inpaint 768^2 bs4):', d['value'], d['ms_per_step'], d['images_per_sec'], d['roofline']['by_class_ms'], d['roofline']['achieved'])
PY
