#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/bench_kernels.py --configs h256,a256,k256,h128,k128 --reps 10 > gpurun_out/bk_x.log 2>&1
cat gpurun_out/bk_x.log | tail -31
