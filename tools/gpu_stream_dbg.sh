#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
F="1536,1536,12"
for dbg in 0 1 2 3; do
  for cold in 1 0; do
    echo "== dbg $dbg cold $cold"
    K22_STREAM_DBG=$dbg timeout 120 python tools/bench_kernels.py --reps 30 --cold $cold --filter "$F" --configs t256x8,m160x5,m160x3 2>&1 | grep -E "^ *1536"
  done
done
OUT=$PWD/gpurun_out/rocprof_streamdbg; rm -rf $OUT
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $OLDPWD/tools/bench_kernels.py --reps 30 --filter "$F" --configs t256x8,m160x5,m160x3 > /dev/null 2>&1 )
SF=$(find $OUT -name "*kernel_stats.csv" | head -1)
[ -n "$SF" ] && python tools/rocprof_summary.py "$SF" "bench_kernels 1536,1536,12" | head -12 | cut -c1-160
find $OUT -name "*kernel_trace.csv" -size +5M -delete
