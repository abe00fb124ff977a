#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2 3; do for lib in libk22hip.so libk22hip_nopk.so; do
  v=$(K22_LIB_PATH=$PWD/kandinsky-2_amd/$lib timeout 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-parity --no-e2e --no-box --no-traffic --no-profile 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*')
  echo "$lib rep $rep: $v"
done; done
echo "[t=$SECONDS s]"
echo "== two-stream probe, default build"; timeout 200 python tools/lds_victim_probe.py 2>&1 | tail -12
echo "== two-stream probe, NOPK build"; K22_LIB_PATH=$PWD/kandinsky-2_amd/libk22hip_nopk.so timeout 200 python tools/lds_victim_probe.py 2>&1 | tail -12
echo "[t=$SECONDS s]"
echo "== NOPK: sampler + movq tests"; K22_LIB_PATH=$PWD/kandinsky-2_amd/libk22hip_nopk.so timeout 500 python -m pytest tests/test_kernels_gpu.py tests/test_movq_gpu.py tests/test_pipeline_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -15
echo "[done t=$SECONDS s]"
