#!/bin/bash
mkdir -p gpurun_out
for d in 0 1 2 3; do
  echo "== K22_STREAM_DBG=$d"
  K22_STREAM_DBG=$d timeout 300 python tools/bench_kernels.py --filter "1536,1536,12;1152,1152,24;3072,1536,12" --configs f160x5,f160x3,f160x10 2>&1 | grep -E "^ +[0-9]+ +[0-9]+ +[0-9]+ "
done > gpurun_out/stream_v2_dbg.txt 2>&1
cat gpurun_out/stream_v2_dbg.txt
