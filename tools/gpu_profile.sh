#!/bin/bash
# rocprofv3 kernel-trace summary of bench.py (run on the GPU box through gpurun).  $1 = tag
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-prof}
STEPS=${2:-10}
EXTRA=${3:-}          # extra bench.py flags, e.g. "--dtype f16x3"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/rocprof_$TAG
rm -rf $OUT
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $OLDPWD/bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --no-parity --no-e2e --no-box $EXTRA > $OLDPWD/gpurun_out/bench_prof_$TAG.log 2>&1 )
tail -1 gpurun_out/bench_prof_$TAG.log | cut -c1-600
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
echo "stats file: $F"
python - "$F" "$STEPS" "$EXTRA" > gpurun_out/rocprof_${TAG}_summary.txt <<'PY'
import csv, sys, re
f, steps, extra = sys.argv[1], int(sys.argv[2]), (sys.argv[3] if len(sys.argv) > 3 else '')
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# rocprofv3 --kernel-trace --stats of: python bench.py --steps {steps} --warmup 3 --no-cpu-baseline {extra}")
print(f"# total kernel time {tot/1e6:.3f} ms over all launches (warm-up, profile pass and timed steps)")
print(f"{'calls':>8} {'total_ms':>10} {'avg_us':>9} {'pct':>6}  kernel")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    name = re.sub(r"\s+", " ", r["Name"])[:150]
    print(f"{int(r['Calls']):8d} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:9.2f} {float(r['Percentage']):6.2f}  {name}")
PY
cat gpurun_out/rocprof_${TAG}_summary.txt | cut -c1-220
# keep the merged-back payload small
find $OUT -name "*kernel_trace.csv" -size +20M -delete
