set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/gn
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "groupnorm" 2>&1 | tail -5 > gpurun_out/gn/tests_gn.txt
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_unet22_gpu.py -q -x -m gpu 2>&1 | tail -8 > gpurun_out/gn/tests_unet.txt
for m in 0 2304 9216 0 2304; do
  K22_GN_FUSED_MAXHW=$m timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/gn/bench_$m.$RANDOM.json
done
