#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
export K22_TUNE_CACHE=$PWD/gpurun_out/tune_cache_h.txt
rm -f $K22_TUNE_CACHE
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "groupnorm or conv" > gpurun_out/pytest_h.log 2>&1
echo "pytest: $(tail -1 gpurun_out/pytest_h.log)"
grep -E "FAILED|Error|assert" gpurun_out/pytest_h.log | head -20
timeout 600 python bench.py --steps 50 --warmup 5 --tuning-report gpurun_out/tuning_h.txt > gpurun_out/bench_h.log 2>&1; tail -1 gpurun_out/bench_h.log | cut -c1-2500
wc -l $K22_TUNE_CACHE
bash tools/gpu_profile.sh r01_v5 10
K22_AUTOTUNE_CACHE= bash tools/gpu_pmc.sh r01_v5
