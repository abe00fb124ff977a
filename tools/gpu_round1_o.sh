#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_prior_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider > gpurun_out/pytest_o.log 2>&1
echo "pytest prior: $(tail -1 gpurun_out/pytest_o.log)"
timeout 300 python tools/bench_prior.py 2>&1 | grep -v "^ \|taps" | tail -4
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/rocprof_prior -- python $OLDPWD/tools/bench_prior.py > /dev/null 2>&1 )
F=$(find gpurun_out/rocprof_prior -name "*kernel_stats.csv" | head -1)
python - "$F" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print(f"{int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:9.3f} ms {float(r['AverageNs'])/1e3:8.2f} us  {re.sub(r'\s+',' ',r['Name'])[:90]}")
PY
bash tools/gpu_pmc_conv.sh sq1
