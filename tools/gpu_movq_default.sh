#!/bin/bash
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_movq_gpu.py -x -q -s -k "real_sizes" 2>&1 | grep -E "reference's own|passed|failed|Error" | tail -8
timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_unet22_gpu.py -x -q -k "generate or pipeline or cache_dir or img2img or inpaint" 2>&1 | tail -3
timeout 600 python bench.py --steps 50 --warmup 5 > gpurun_out/bench_r03h.log 2> gpurun_out/bench_r03h.err; tail -1 gpurun_out/bench_r03h.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], json.dumps(d['e2e']), json.dumps({k: v for k, v in d['parity_paths'].items() if k != 'reference'})[:400])
"
} > gpurun_out/movq_default.txt 2>&1
cat gpurun_out/movq_default.txt
