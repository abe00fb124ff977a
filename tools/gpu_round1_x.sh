#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== prior"; timeout 300 python tools/bench_prior.py 2>&1 | grep -v amdgpu.ids | tail -4
echo "== movq"; timeout 300 python tools/bench_movq.py 2>&1 | grep -v amdgpu.ids | tail -4
echo "== C3 per-GPU shape (1024^2, bs 4)"; timeout 600 python bench.py --size 1024 --bs 4 --steps 10 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | tail -1 | cut -c1-200
echo "== C4 inpaint 768^2 bs 4"; timeout 600 python bench.py --inpaint --bs 4 --steps 10 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | tail -1 | cut -c1-200
