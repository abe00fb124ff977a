#!/bin/bash
# round 6, first diagnostic pass: baseline bench line of the round-5 binary on this box + kernel trace of the prior forward
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-parity --no-e2e > gpurun_out/bench_r06_base.log 2>&1; tail -1 gpurun_out/bench_r06_base.log | cut -c1-400
echo "[t=$SECONDS s]"
timeout 200 python tools/bench_prior.py 2>&1 | grep -E "prior|ms" | head -5
OUT=$PWD/gpurun_out/rocprof_prior_r06a
rm -rf $OUT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $OLDPWD/tools/bench_prior.py > $OLDPWD/gpurun_out/bench_prior_prof.log 2>&1 )
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
python tools/rocprof_summary.py "$F" "python tools/bench_prior.py" > gpurun_out/rocprof_prior_r06a_summary.txt
head -40 gpurun_out/rocprof_prior_r06a_summary.txt | cut -c1-200
find $OUT -name "*kernel_trace.csv" -size +20M -delete
echo "[done t=$SECONDS s]"
