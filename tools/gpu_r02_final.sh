#!/bin/bash
# round 2 final pass: tile table -> full GPU suite -> smoke -> bench lines -> rocprof kernel trace -> PMC traffic
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r02f}
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "${SKIP_TABLE:-0}" != "1" ]; then
  timeout 1200 python tools/make_tile_table.py gpurun_out/tiles_gfx950.txt > gpurun_out/tiles_$TAG.log 2>&1
  tail -2 gpurun_out/tiles_$TAG.log
  cp gpurun_out/tiles_gfx950.txt kandinsky-2_amd/tiles_gfx950.txt
fi
export K22_PARITY_REPORT=$PWD/gpurun_out/parity_$TAG.json
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider --timeout 900 > gpurun_out/pytest_$TAG.log 2>&1
echo "pytest: $(grep -E ' passed| failed' gpurun_out/pytest_$TAG.log | tail -1)"
grep -E "^FAILED|^ERROR|^E  " gpurun_out/pytest_$TAG.log | head -30
grep -E "fp32:|bf16:|bfloat16|float32|max\|d\||drift|uint8" gpurun_out/pytest_$TAG.log | grep -v "^tests" | head -150 > gpurun_out/parity_lines_$TAG.txt
unset K22_PARITY_REPORT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
timeout 600 python bench.py --steps 50 --warmup 5 --tuning-report gpurun_out/tuning_$TAG.txt > gpurun_out/bench_$TAG.log 2>&1; tail -1 gpurun_out/bench_$TAG.log | cut -c1-220; tail -1 gpurun_out/bench_$TAG.log | grep -o '"by_class_ms[^}]*}'; tail -1 gpurun_out/bench_$TAG.log | grep -o '"cpu_baseline.*'
for cfg in "--head 2.2" "--controlnet --bs 2" "--inpaint --bs 4" "--size 1024 --bs 4"; do
  tag=$(echo $cfg | tr -d ' -')
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $cfg > gpurun_out/bench_${TAG}_$tag.log 2>&1
  echo "$cfg: $(tail -1 gpurun_out/bench_${TAG}_$tag.log | grep -o '"value": [0-9.]*') $(tail -1 gpurun_out/bench_${TAG}_$tag.log | grep -o '"tile_configs_measured_in_this_process": [0-9]*') $(tail -1 gpurun_out/bench_${TAG}_$tag.log | grep -o '"by_class_ms[^}]*}')"
done
timeout 300 python tools/bench_prior.py 2>&1 | grep -E "prior forward|steady" | head -3
timeout 300 python tools/bench_movq.py 2>&1 | tail -3
timeout 300 python tools/bench_encoders.py 2>&1 | tail -1 | tee gpurun_out/bench_encoders_$TAG.json
EOUT=$PWD/gpurun_out/rocprof_${TAG}_enc; rm -rf $EOUT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $EOUT -- python $OLDPWD/tools/bench_encoders.py --reps 10 > /dev/null 2>&1 )
EF=$(find $EOUT -name "*kernel_stats.csv" | head -1)
[ -n "$EF" ] && python tools/rocprof_summary.py "$EF" "python tools/bench_encoders.py --reps 10" > gpurun_out/rocprof_${TAG}_encoders_summary.txt
find $EOUT -name "*kernel_trace.csv" -size +20M -delete
head -12 gpurun_out/rocprof_${TAG}_encoders_summary.txt | cut -c1-150
bash tools/gpu_profile.sh $TAG 10 > gpurun_out/profile_$TAG.log 2>&1
head -24 gpurun_out/rocprof_${TAG}_summary.txt | cut -c1-150
bash tools/gpu_pmc.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1
head -14 gpurun_out/pmc_${TAG}_summary.txt | cut -c1-170
