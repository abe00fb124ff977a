#!/bin/bash
# stream kernel v2: parity (both weight layouts) + timing against the halo / gemm8 kernels at the small-M shapes
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_stream_gpu.py -x -q 2>&1 | tail -3 > gpurun_out/stream_v2_pytest.txt
timeout 600 python tools/bench_kernels.py --filter "1536,1536,12;3072,1536,12;2304,1536,12;1152,1152,24;2304,1152,24;1920,1152,24;1152,1536,12" \
   --configs auto,t256x8,f160x4,f160x5,f160x6,f288x2,f288x3 > gpurun_out/stream_v2_conv.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --filter "1536,1536,12" --configs f160x5,t256x8 --reps 50 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY' > gpurun_out/stream_v2_kernels.txt
import csv, glob
f = glob.glob('/tmp/prof_s/**/*kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    print(r['Name'][:90], r['Calls'], r['AverageNs'], r['MinNs'])
PY
cat gpurun_out/stream_v2_pytest.txt gpurun_out/stream_v2_conv.txt gpurun_out/stream_v2_kernels.txt
