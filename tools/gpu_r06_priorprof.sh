#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
TAG=${1:-r06b}
OUT=$PWD/gpurun_out/rocprof_prior_$TAG
rm -rf $OUT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $OLDPWD/tools/bench_prior.py > $OLDPWD/gpurun_out/bench_prior_prof_$TAG.log 2>&1 )
grep -E "^prior" gpurun_out/bench_prior_prof_$TAG.log
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
python tools/rocprof_summary.py "$F" "python tools/bench_prior.py" > gpurun_out/rocprof_prior_${TAG}_summary.txt
head -16 gpurun_out/rocprof_prior_${TAG}_summary.txt | cut -c1-200
# gaps: time between consecutive kernels of the graph replay
python - "$(find $OUT -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 1000 kernels = steady-state sampling loop
rows = rows[-1200:]
gaps, durs = [], []
for a, b in zip(rows, rows[1:]):
    gaps.append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
    durs.append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
import statistics
print("steady-state: kernels", len(rows), "mean duration us", sum(durs) / len(durs) / 1e3, "mean gap us", statistics.mean(g for g in gaps if g < 20000) / 1e3, "median gap", statistics.median(gaps) / 1e3)
PY
find $OUT -name "*kernel_trace.csv" -size +20M -delete
