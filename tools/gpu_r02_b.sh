#!/bin/bash
# round 2, call B: full GPU suite with the parity report, attention variants, bench lines of every config
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
export K22_PARITY_REPORT=$PWD/gpurun_out/parity_b.json
export K22_TUNE_CACHE=$PWD/gpurun_out/tune_cache_b.txt
rm -f $K22_TUNE_CACHE
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider --timeout 900 > gpurun_out/pytest_b.log 2>&1
echo "pytest: $(tail -1 gpurun_out/pytest_b.log)"
grep -E "^FAILED|^ERROR|Error:|assert " gpurun_out/pytest_b.log | head -30
grep -E "fp32:|bf16:|bfloat16|max\|d\||drift|uint8" gpurun_out/pytest_b.log | grep -v "^tests" | head -120 > gpurun_out/parity_lines_b.txt
echo "--- attention default / pipelined"
timeout 120 python tools/bench_attn.py 30 0 2>&1 | tail -4
timeout 120 python tools/bench_attn.py 30 1 2>&1 | tail -4
echo "--- bench"
timeout 600 python bench.py --steps 50 --warmup 5 --tuning-report gpurun_out/tuning_b.txt > gpurun_out/bench_b.log 2>&1; tail -1 gpurun_out/bench_b.log | cut -c1-300; tail -1 gpurun_out/bench_b.log | grep -o '"by_class_ms[^}]*}'
for cfg in "--head 2.2" "--controlnet --bs 2" "--inpaint --bs 4" "--size 1024 --bs 4"; do
  tag=$(echo $cfg | tr -d ' -')
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $cfg > gpurun_out/bench_b_$tag.log 2>&1
  echo "$cfg: $(tail -1 gpurun_out/bench_b_$tag.log | cut -c1-160) $(tail -1 gpurun_out/bench_b_$tag.log | grep -o '"by_class_ms[^}]*}')"
done
