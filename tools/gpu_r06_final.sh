#!/bin/bash
# round 6 final pass, most important first; every item is skipped once the time budget ($2 seconds, default 1500) is nearly spent:
# full GPU suite -> smoke -> bench lines (default, the driver's arguments, other engines / configs) -> rocprof kernel trace -> PMC traffic / matrix-pipe passes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r06}
BUDGET=${2:-2000}
mkdir -p gpurun_out
export TMPDIR=/tmp
room() { [ $((BUDGET - SECONDS)) -gt $1 ]; }     # room N: at least N seconds of budget left
export K22_PARITY_REPORT=$PWD/gpurun_out/parity_$TAG.json
if [ -z "$SKIP_SUITE" ]; then
timeout 1200 python -m pytest tests -m gpu -q -s -p no:cacheprovider --durations=15 > gpurun_out/pytest_$TAG.log 2>&1
echo "pytest: $(grep -E ' passed| failed' gpurun_out/pytest_$TAG.log | tail -1)   [t=$SECONDS s]"
grep -E "^FAILED|^ERROR" gpurun_out/pytest_$TAG.log | head -20
grep -E "fp32:|bf16:|fp16:|f16x3|f16x2|bfloat16|float16|float32|max\|d\||drift|uint8|c3_loop" gpurun_out/pytest_$TAG.log | grep -v "^tests" | head -400 > gpurun_out/parity_lines_$TAG.txt
grep -A18 "slowest" gpurun_out/pytest_$TAG.log > gpurun_out/pytest_durations_$TAG.txt
fi
unset K22_PARITY_REPORT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -E "smoke|Error|assert" | tee gpurun_out/smoke_$TAG.txt
echo "[t=$SECONDS s]"
timeout 900 python bench.py --steps 50 --warmup 5 --tuning-report gpurun_out/tuning_$TAG.txt > gpurun_out/bench_$TAG.log 2> gpurun_out/bench_$TAG.err; tail -1 gpurun_out/bench_$TAG.log | cut -c1-200
tail -1 gpurun_out/bench_$TAG.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('by_class', d['roofline']['by_class_ms'], 'frac', d['roofline']['frac'], d['roofline']['by_class_frac'])
print('parity', json.dumps({k:v for k,v in (d.get('parity_paths') or {}).items() if k!='reference'}))
print('gate', d.get('gate_holding')); print('box', d.get('box'))
e=d.get('e2e') or {}; print('e2e', e.get('images_per_sec'), e.get('phases_ms'), json.dumps(e.get('other_engines')))
print('cpu', d['cpu_baseline']['value'] if d.get('cpu_baseline') else None)"
echo "[t=$SECONDS s]"
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-e2e --no-traffic > gpurun_out/bench_${TAG}_driverargs.log 2>&1; tail -1 gpurun_out/bench_${TAG}_driverargs.log | cut -c1-260
run_cfg() {
  local cfg="$1"; local tag=$(echo $cfg | tr -d ' -')
  timeout 240 python bench.py --steps 50 --warmup 3 --no-cpu-baseline --no-parity --no-e2e --no-traffic $cfg > gpurun_out/bench_${TAG}_$tag.log 2>&1
  echo "$cfg: $(tail -1 gpurun_out/bench_${TAG}_$tag.log | grep -o '"value": [0-9.]*') $(tail -1 gpurun_out/bench_${TAG}_$tag.log | grep -o '"tile_configs_measured_in_this_process": [0-9]*') $(tail -1 gpurun_out/bench_${TAG}_$tag.log | grep -o '"frac": [0-9.]*') $(tail -1 gpurun_out/bench_${TAG}_$tag.log | grep -o '"by_class_ms[^}]*}')"
}
room 60 && run_cfg "--dtype f16x2"
room 60 && run_cfg "--dtype f16x3"
room 60 && run_cfg "--dtype fp16"
room 90 && { bash tools/gpu_profile.sh $TAG 10 "--no-traffic" > gpurun_out/profile_$TAG.log 2>&1; head -24 gpurun_out/rocprof_${TAG}_summary.txt | cut -c1-150; }
room 90 && { bash tools/gpu_profile.sh ${TAG}_f16x2 10 "--dtype f16x2 --no-traffic" > gpurun_out/profile_${TAG}_f16x2.log 2>&1; head -16 gpurun_out/rocprof_${TAG}_f16x2_summary.txt | cut -c1-150; }
echo "[t=$SECONDS s]"
room 60 && run_cfg "--inpaint --bs 4"
room 60 && run_cfg "--size 1024 --bs 4"
room 150 && { bash tools/gpu_pmc.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1; head -14 gpurun_out/pmc_${TAG}_summary.txt | cut -c1-170; }
room 150 && { bash tools/gpu_pmc_mfma.sh ${TAG}_mfma > gpurun_out/pmc_${TAG}_mfma.log 2>&1; head -20 gpurun_out/pmc_${TAG}_mfma_summary.txt | cut -c1-170; }
room 60 && run_cfg "--head 2.2"
room 60 && run_cfg "--controlnet --bs 2"
room 120 && { bash tools/gpu_pmc_mfma.sh ${TAG}_mfma_f16x2 "--dtype f16x2" > gpurun_out/pmc_${TAG}_mfma_f16x2.log 2>&1; head -14 gpurun_out/pmc_${TAG}_mfma_f16x2_summary.txt | cut -c1-170; }
room 30 && { timeout 120 tools/micro/attn_probe > gpurun_out/attn_probe_$TAG.txt 2>&1; grep -E "workgroups|shipped|pipe_kernel|differ" gpurun_out/attn_probe_$TAG.txt | head -16; }
echo "[done t=$SECONDS s]"
