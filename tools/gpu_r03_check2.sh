#!/bin/bash
# round 3 checkpoint 2: new tests (CLIP-bigG tower, whole-loop graph), bench with the loop graph, MoVQ dtype timings + rocprof of MoVQ
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03b}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_encoders_gpu.py tests/test_unet_gpu.py tests/test_pipeline_gpu.py "tests/test_full_size_gpu.py::test_full_size_p_sampler_fp32_gate" -m gpu -q -s -x -p no:cacheprovider --timeout 900 -k "clip_vision or whole_loop or fp32_gate or generate" > gpurun_out/pytest_$TAG.log 2>&1
echo "pytest: $(grep -E ' passed| failed' gpurun_out/pytest_$TAG.log | tail -1)"
grep -E "^FAILED|^ERROR|^E  " gpurun_out/pytest_$TAG.log | head -30
grep -E "enc_clipvision" gpurun_out/pytest_$TAG.log | head
for extra in "" "--no-loop-graph"; do
  timeout 900 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-parity $extra > gpurun_out/bench_${TAG}$extra.log 2> gpurun_out/bench_${TAG}$extra.err
  tail -1 gpurun_out/bench_${TAG}$extra.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$extra value', d['value'], 'ms', d['ms_per_step'], 'loop', d['config'].get('loop_graph'), 'e2e', d['e2e'] and (d['e2e']['images_per_sec'], d['e2e']['phases_ms']))"
  tail -3 gpurun_out/bench_${TAG}$extra.err | grep -v amdgpu.ids
done
for dt in bf16 fp16 fp32; do timeout 300 python tools/bench_movq.py --dtype $dt 2>&1 | grep -E "decode|encode"; done
OUT=$PWD/gpurun_out/rocprof_${TAG}_movq; rm -rf $OUT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $OLDPWD/tools/bench_movq.py > /dev/null 2>&1 )
SF=$(find $OUT -name "*kernel_stats.csv" | head -1)
[ -n "$SF" ] && python tools/rocprof_summary.py "$SF" "python tools/bench_movq.py (bf16, 768 px)" > gpurun_out/rocprof_${TAG}_movq_summary.txt
find $OUT -name "*kernel_trace.csv" -size +5M -delete
head -22 gpurun_out/rocprof_${TAG}_movq_summary.txt | cut -c1-150
