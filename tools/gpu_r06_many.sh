#!/bin/bash
# round 6: generate_text2img_many (pipeline over prompts) test + the e2e pass of the bench line + attention kernel baseline
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q -p no:cacheprovider -k "many" 2>&1 | tail -8
echo "[t=$SECONDS s]"
timeout 120 python tools/bench_attn.py 50 2>&1 | tail -4
echo "[t=$SECONDS s]"
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-parity --no-traffic --no-profile > gpurun_out/bench_many.log 2> gpurun_out/bench_many.err; tail -3 gpurun_out/bench_many.err
tail -1 gpurun_out/bench_many.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value', d['value'], 'two_chains', d.get('two_chains'))
e=d.get('e2e') or {}; print('e2e', e.get('images_per_sec'), e.get('phases_ms'), json.dumps(e.get('pipelined_over_prompts')))"
echo "[done t=$SECONDS s]"
