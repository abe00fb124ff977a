#!/bin/bash
# round 2, call C: changed paths (stem, sampler threshold, FiLM side branch, pipeline / comm tests), smoke numbers, side-branch A/B, rocprof
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_parallel_gpu.py tests/test_unet_gpu.py tests/test_unet22_gpu.py tests/test_full_size_gpu.py -m gpu -q -s -p no:cacheprovider -k "not c4 and not c3" > gpurun_out/pytest_c.log 2>&1
echo "pytest: $(tail -1 gpurun_out/pytest_c.log)"
grep -E "^FAILED|^ERROR|^E  " gpurun_out/pytest_c.log | head -30
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "sampler or percentile" 2>&1 | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
echo "--- side branch off / on / off / on"
for sb in 0 1 0 1; do
  K22_SIDE_BRANCH=$sb timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*, "unit": "steps/s"' | sed "s/^/side=$sb /"
done
bash tools/gpu_profile.sh r02c 10 > gpurun_out/profile_c.log 2>&1
head -32 gpurun_out/rocprof_r02c_summary.txt | cut -c1-170
