#!/bin/bash
# re-measures the tile-table lines of the split-precision engines (dtype codes 3 = f16x3, 4 = f16x2) after the chunk-format change
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
awk '$1 != "3" && $1 != "4"' kandinsky-2_amd/tiles_gfx950.txt > /tmp/t0.txt
wc -l /tmp/t0.txt
K22_TILE_TABLE=/tmp/t0.txt timeout 1200 python tools/make_tile_table.py --x3-only /tmp/t1.txt 2>&1 | tail -2
echo "[t=$SECONDS s]"
K22_TILE_TABLE=/tmp/t1.txt timeout 1200 python tools/make_tile_table.py --x2-only /tmp/t2.txt 2>&1 | tail -2
echo "[t=$SECONDS s]"
cp /tmp/t2.txt gpurun_out/tiles_gfx950_r06.txt; wc -l gpurun_out/tiles_gfx950_r06.txt
for dt in f16x2 f16x3; do for tb in kandinsky-2_amd/tiles_gfx950.txt /tmp/t2.txt; do
  K22_TILE_TABLE=$tb timeout 200 python bench.py --dtype $dt --chains 1 --steps 50 --warmup 5 --no-cpu-baseline --no-parity --no-e2e --no-box --no-traffic 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$dt $tb', d['value'], d['roofline']['by_class_ms'])"
done; done
echo "[done t=$SECONDS s]"
