#!/bin/bash
# round 4 final pass: full GPU suite -> smoke -> bench lines -> once-per-image engines -> rocprof kernel trace -> PMC traffic
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r04f}
mkdir -p gpurun_out
export TMPDIR=/tmp
export K22_PARITY_REPORT=$PWD/gpurun_out/parity_$TAG.json
timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider --timeout 900 > gpurun_out/pytest_$TAG.log 2>&1
echo "pytest: $(grep -E ' passed| failed' gpurun_out/pytest_$TAG.log | tail -1)"
grep -E "^FAILED|^ERROR|^E  " gpurun_out/pytest_$TAG.log | head -30
grep -E "fp32:|bf16:|fp16:|f16x3:|f16x3|bfloat16|float16|float32|max\|d\||drift|uint8" gpurun_out/pytest_$TAG.log | grep -v "^tests" | head -300 > gpurun_out/parity_lines_$TAG.txt
unset K22_PARITY_REPORT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee gpurun_out/smoke_$TAG.txt
timeout 900 python bench.py --steps 50 --warmup 5 --tuning-report gpurun_out/tuning_$TAG.txt > gpurun_out/bench_$TAG.log 2> gpurun_out/bench_$TAG.err; tail -1 gpurun_out/bench_$TAG.log | cut -c1-180
tail -1 gpurun_out/bench_$TAG.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('by_class', d['roofline']['by_class_ms'], 'frac', d['roofline']['frac'], 'gn', d['roofline']['groupnorm_frac_hbm'])
print('parity', json.dumps({k:v for k,v in (d.get('parity_paths') or {}).items() if k!='reference'}))
print('e2e', json.dumps(d.get('e2e'))[:420])
print('cpu', d['cpu_baseline']['value'] if d.get('cpu_baseline') else None)"
for cfg in "--dtype f16x3" "--dtype fp16" "--inpaint --bs 4" "--size 1024 --bs 4" "--head 2.2" "--controlnet --bs 2"; do
  tag=$(echo $cfg | tr -d ' -')
  timeout 600 python bench.py --steps 50 --warmup 3 --no-cpu-baseline --no-parity --no-e2e $cfg > gpurun_out/bench_${TAG}_$tag.log 2>&1
  echo "$cfg: $(tail -1 gpurun_out/bench_${TAG}_$tag.log | grep -o '"value": [0-9.]*') $(tail -1 gpurun_out/bench_${TAG}_$tag.log | grep -o '"tile_configs_measured_in_this_process": [0-9]*') $(tail -1 gpurun_out/bench_${TAG}_$tag.log | grep -o '"frac": [0-9.]*') $(tail -1 gpurun_out/bench_${TAG}_$tag.log | grep -o '"by_class_ms[^}]*}')"
done
K22_CHAINS=2 timeout 600 python bench.py --steps 50 --warmup 3 --no-cpu-baseline --no-parity --no-e2e > gpurun_out/bench_${TAG}_chains2.log 2>&1
echo "K22_CHAINS=2: $(tail -1 gpurun_out/bench_${TAG}_chains2.log | grep -o '"value": [0-9.]*')"
timeout 300 python tools/bench_prior.py 2>&1 | grep -E "prior forward|steady" | head -3
for dt in bf16 fp16; do timeout 300 python tools/bench_movq.py --dtype $dt 2>&1 | grep -E "decode"; done
bash tools/gpu_profile.sh $TAG 10 > gpurun_out/profile_$TAG.log 2>&1
head -26 gpurun_out/rocprof_${TAG}_summary.txt | cut -c1-150
bash tools/gpu_pmc.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1
head -12 gpurun_out/pmc_${TAG}_summary.txt | cut -c1-170
