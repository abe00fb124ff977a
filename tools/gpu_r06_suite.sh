#!/bin/bash
# full GPU suite + smoke (validation pass); durations of every test kept for the per-file budget
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export K22_PARITY_REPORT=$PWD/gpurun_out/parity_r06.json
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=0 -s > gpurun_out/pytest_r06.log 2>&1
echo "pytest: $(grep -E ' passed| failed' gpurun_out/pytest_r06.log | tail -1)   [t=$SECONDS s]"
grep -E "^FAILED|^ERROR" gpurun_out/pytest_r06.log | head -30
grep -E "fp32:|bf16:|fp16:|f16x3|f16x2|bfloat16|float16|float32|max\|d\||drift|uint8|c3_loop" gpurun_out/pytest_r06.log | grep -v "^tests" | head -400 > gpurun_out/parity_lines_r06.txt
python - <<'PY'
import re, collections
per = collections.Counter(); n = collections.Counter()
for l in open('gpurun_out/pytest_r06.log'):
    m = re.match(r"\s*([\d.]+)s (call|setup|teardown)\s+(tests/[\w_]+\.py)", l)
    if m: per[m.group(3)] += float(m.group(1)); n[m.group(3)] += 1
for k, v in per.most_common(): print(f"{v:8.1f} s  {n[k]:5d} entries  {k}")
PY
unset K22_PARITY_REPORT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -E "smoke|Error|assert" | tee gpurun_out/smoke_r06.txt
echo "[done t=$SECONDS s]"
