#!/bin/bash
# round 4 final pass, most important first, every item skipped once the time budget ($2 seconds, default 1080) is nearly spent:
# full GPU suite -> smoke -> bench lines -> rocprof kernel trace -> secondary configs -> once-per-image engines -> PMC traffic
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r04f}
BUDGET=${2:-1080}
mkdir -p gpurun_out
export TMPDIR=/tmp
room() { [ $((BUDGET - SECONDS)) -gt $1 ]; }     # room N: at least N seconds of budget left
export K22_PARITY_REPORT=$PWD/gpurun_out/parity_$TAG.json
timeout 860 python -m pytest tests -m gpu -q -s -p no:cacheprovider --timeout 600 > gpurun_out/pytest_$TAG.log 2>&1
echo "pytest: $(grep -E ' passed| failed' gpurun_out/pytest_$TAG.log | tail -1)   [t=$SECONDS s]"
grep -E "^FAILED|^ERROR" gpurun_out/pytest_$TAG.log | head -20
grep -E "fp32:|bf16:|fp16:|f16x3|bfloat16|float16|float32|max\|d\||drift|uint8" gpurun_out/pytest_$TAG.log | grep -v "^tests" | head -300 > gpurun_out/parity_lines_$TAG.txt
unset K22_PARITY_REPORT
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee gpurun_out/smoke_$TAG.txt
timeout 400 python bench.py --steps 50 --warmup 5 --tuning-report gpurun_out/tuning_$TAG.txt > gpurun_out/bench_$TAG.log 2> gpurun_out/bench_$TAG.err; tail -1 gpurun_out/bench_$TAG.log | cut -c1-180
tail -1 gpurun_out/bench_$TAG.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('by_class', d['roofline']['by_class_ms'], 'frac', d['roofline']['frac'], 'gn', d['roofline']['groupnorm_frac_hbm'])
print('parity', json.dumps({k:v for k,v in (d.get('parity_paths') or {}).items() if k!='reference'}))
e=d.get('e2e') or {}; print('e2e', e.get('images_per_sec'), e.get('phases_ms'), json.dumps(e.get('other_engines')))
print('cpu', d['cpu_baseline']['value'] if d.get('cpu_baseline') else None)"
echo "[t=$SECONDS s]"
run_cfg() {
  local cfg="$1"; local tag=$(echo $cfg | tr -d ' -')
  timeout 200 python bench.py --steps 50 --warmup 3 --no-cpu-baseline --no-parity --no-e2e $cfg > gpurun_out/bench_${TAG}_$tag.log 2>&1
  echo "$cfg: $(tail -1 gpurun_out/bench_${TAG}_$tag.log | grep -o '"value": [0-9.]*') $(tail -1 gpurun_out/bench_${TAG}_$tag.log | grep -o '"tile_configs_measured_in_this_process": [0-9]*') $(tail -1 gpurun_out/bench_${TAG}_$tag.log | grep -o '"frac": [0-9.]*') $(tail -1 gpurun_out/bench_${TAG}_$tag.log | grep -o '"by_class_ms[^}]*}')"
}
room 60 && run_cfg "--dtype f16x3"
room 60 && run_cfg "--dtype fp16"
room 90 && { bash tools/gpu_profile.sh $TAG 10 > gpurun_out/profile_$TAG.log 2>&1; head -22 gpurun_out/rocprof_${TAG}_summary.txt | cut -c1-150; }
room 60 && run_cfg "--inpaint --bs 4"
room 60 && run_cfg "--size 1024 --bs 4"
# large plain GEMMs on the engine's GEMM kernels (the Winograd price model of DESIGN 9 R4-7: 16 x 4608 x 768 x 768 as one M = 73728 problem)
room 50 && { timeout 120 python tools/bench_kernels.py --gemm --filter "1,1,1" --extra "73728,768,768;18432,768,768;73728,384,384" --configs "256x0x1,128x0x1,128x128x1,128x64x1" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/big_gemm_$TAG.txt | tail -8; }
echo "[t=$SECONDS s]"
room 60 && run_cfg "--head 2.2"
room 60 && run_cfg "--controlnet --bs 2"
room 60 && { K22_CHAINS=2 timeout 200 python bench.py --steps 50 --warmup 3 --no-cpu-baseline --no-parity --no-e2e > gpurun_out/bench_${TAG}_chains2.log 2>&1; echo "K22_CHAINS=2: $(tail -1 gpurun_out/bench_${TAG}_chains2.log | grep -o '"value": [0-9.]*')"; }
room 60 && { timeout 120 python tools/bench_prior.py 2>&1 | grep -E "prior forward|steady" | head -3; }
room 60 && for dt in bf16 fp16; do timeout 100 python tools/bench_movq.py --dtype $dt 2>&1 | grep -E "decode"; done
room 90 && { bash tools/gpu_profile.sh ${TAG}_f16x3 10 "--dtype f16x3" > gpurun_out/profile_${TAG}_f16x3.log 2>&1; head -14 gpurun_out/rocprof_${TAG}_f16x3_summary.txt | cut -c1-150; }
room 150 && { bash tools/gpu_pmc.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1; head -12 gpurun_out/pmc_${TAG}_summary.txt | cut -c1-170; }
echo "[done t=$SECONDS s]"
