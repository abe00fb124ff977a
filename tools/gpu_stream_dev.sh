#!/bin/bash
# development pass for the weight-streaming kernel: unit parity, then per-shape timing against the tile table's picks
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
TAG=${1:-s1}
timeout 900 python -m pytest tests/test_stream_gpu.py -q -m gpu -p no:cacheprovider --timeout 300 > gpurun_out/stream_tests_$TAG.log 2>&1
tail -25 gpurun_out/stream_tests_$TAG.log
timeout 400 python tools/bench_kernels.py --reps 20 --filter "1536,1536,12;3072,1536,12;1152,1536,12;1152,1152,24;2304,1152,24;768,1152,24" \
  --configs t256x8,t256x10,h256x10,t256x4,t128x2,m160x3,m160x4,m160x5,m160x6,m160x8,m288x2,m288x3,m288x4,m288x6 > gpurun_out/stream_bench_conv_$TAG.txt 2>&1
cat gpurun_out/stream_bench_conv_$TAG.txt
timeout 400 python tools/bench_kernels.py --gemm --reps 20 --filter "288,4608,1536;288,1536,1536;1152,3456,1152;1152,1152,1152" --extra "162,2048,2048;162,8192,2048;162,2048,8192;162,6144,2048" \
  --configs auto,128x0x1,128x0x3,128x64x1,m160x1,m160x2,m160x3,m160x4,m288x1,m288x2,m288x3 > gpurun_out/stream_bench_gemm_$TAG.txt 2>&1
cat gpurun_out/stream_bench_gemm_$TAG.txt
