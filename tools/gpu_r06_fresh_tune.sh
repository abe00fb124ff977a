#!/bin/bash
# is the shipped tile table still the best set on the current kernels?  bench step with the table against a fresh in-process tune, same box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
F="--steps 50 --warmup 5 --no-cpu-baseline --no-parity --no-e2e --no-box --no-traffic"
for rep in 1 2; do
  v=$(timeout 300 python bench.py $F --tuning-report gpurun_out/tuning_table.txt 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['by_class_ms'], d['config'].get('tile_configs_measured_in_this_process'))")
  echo "table rep $rep: $v"
  v=$(K22_TILE_TABLE=0 timeout 600 python bench.py $F --tuning-report gpurun_out/tuning_fresh$rep.txt 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['by_class_ms'], d['config'].get('tile_configs_measured_in_this_process'))")
  echo "fresh rep $rep: $v"
done
diff gpurun_out/tuning_table.txt gpurun_out/tuning_fresh1.txt | head -80
echo "[done t=$SECONDS s]"
