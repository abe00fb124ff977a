#!/bin/bash
# round 2, call A: tile table -> C2 parity numbers with it -> bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/make_tile_table.py gpurun_out/tiles_gfx950.txt > gpurun_out/tiles.log 2>&1
tail -4 gpurun_out/tiles.log
export K22_TILE_TABLE=$PWD/gpurun_out/tiles_gfx950.txt
export K22_PARITY_REPORT=$PWD/gpurun_out/parity_a.json
timeout 900 python -m pytest tests/test_full_size_gpu.py -q -x -s -k "c2 or tuner" -p no:cacheprovider > gpurun_out/pytest_full_a.log 2>&1
grep -E "fp32:|bf16:|passed|failed|Error|error" gpurun_out/pytest_full_a.log | head -40
timeout 600 python bench.py --steps 50 --warmup 5 --tuning-report gpurun_out/tuning_a.txt > gpurun_out/bench_a.log 2>&1
tail -1 gpurun_out/bench_a.log | cut -c1-1800
timeout 600 python -m pytest tests/test_unet_gpu.py -q -x -p no:cacheprovider > gpurun_out/pytest_unet_a.log 2>&1
tail -3 gpurun_out/pytest_unet_a.log
