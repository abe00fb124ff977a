#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/retune_gemm8_spec.py kandinsky-2_amd/tiles_gfx950.txt gpurun_out/tiles_gfx950_spec.txt 2>&1 | grep -v amdgpu.ids | tail -90
echo "[t=$SECONDS s]"
NEW=$PWD/gpurun_out/tiles_gfx950_spec.txt
F="--steps 50 --warmup 5 --no-cpu-baseline --no-parity --no-e2e --no-box --no-traffic"
for rep in 1 2 3; do
  for tb in old new; do
    if [ $tb = new ]; then export K22_TILE_TABLE=$NEW; else unset K22_TILE_TABLE; fi
    v=$(timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['by_class_ms']['gemm'], d['config'].get('tile_configs_measured_in_this_process'))")
    echo "C2 $tb rep $rep: $v"
  done
done
for tb in old new old new; do
  if [ $tb = new ]; then export K22_TILE_TABLE=$NEW; else unset K22_TILE_TABLE; fi
  v=$(timeout 300 python bench.py --size 1024 --bs 4 --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-e2e --no-box --no-traffic 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['by_class_ms']['gemm'], d['config'].get('tile_configs_measured_in_this_process'))")
  echo "C3 $tb: $v"
done
unset K22_TILE_TABLE
echo "[done t=$SECONDS s]"
