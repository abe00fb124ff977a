#!/bin/bash
# round 4: split-precision tile-table lines + the full-size gates + smoke
mkdir -p gpurun_out
timeout 1200 python tools/make_tile_table.py --x3-only gpurun_out/tiles_with_x3.txt > gpurun_out/x3_table.log 2>&1
tail -3 gpurun_out/x3_table.log
export K22_TILE_TABLE=$PWD/gpurun_out/tiles_with_x3.txt
timeout 1500 python -m pytest tests/test_full_size_gpu.py -x -q -s -k "split_precision or c3_forward" 2>&1 | grep -v "^$" | tail -40 > gpurun_out/x3_fullsize.txt
cat gpurun_out/x3_fullsize.txt | cut -c1-220
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 | tee gpurun_out/x3_smoke.txt
