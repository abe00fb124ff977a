#!/bin/bash
mkdir -p gpurun_out
{
timeout 1500 python tools/make_tile_table.py --small-conv gpurun_out/tiles_gfx950.txt 2>&1 | grep -v amdgpu.ids | tail -25
export K22_TILE_TABLE=$PWD/gpurun_out/tiles_gfx950.txt
grep -c " 20 " gpurun_out/tiles_gfx950.txt
for m in 1 0 1 0; do
  echo "== K22_STREAM=$m"
  K22_STREAM=$m timeout 600 python bench.py --steps 50 --warmup 10 --no-parity --no-e2e --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['by_class_ms'], d['roofline']['frac'], d['config'].get('tile_configs_measured_in_this_process'))
"
done
timeout 900 python -m pytest tests/test_unet_gpu.py -x -q -k "golden or tiny" 2>&1 | tail -3
} > gpurun_out/stream_engine.txt 2>&1
cat gpurun_out/stream_engine.txt
