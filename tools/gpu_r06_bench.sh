#!/bin/bash
# round 6: the default bench line (all side measurements incl. live PMC traffic), the driver's arguments, rocprofv3 kernel trace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r06}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --steps 50 --warmup 5 --tuning-report gpurun_out/tuning_$TAG.txt > gpurun_out/bench_$TAG.log 2> gpurun_out/bench_$TAG.err; tail -1 gpurun_out/bench_$TAG.log | cut -c1-200
tail -5 gpurun_out/bench_$TAG.err
tail -1 gpurun_out/bench_$TAG.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('by_class', r['by_class_ms'], 'frac', r['frac'], r['by_class_frac'])
print('traffic', r['traffic'], r.get('traffic_detail'), r['traffic_source'][:120])
print('parity', json.dumps({k:v for k,v in (d.get('parity_paths') or {}).items() if k!='reference'}))
print('gate', d.get('gate_holding')); print('box', d.get('box'))
e=d.get('e2e') or {}; print('e2e', e.get('images_per_sec'), e.get('phases_ms'), e.get('prior_weight_stream_tb_per_s'), json.dumps(e.get('other_engines')))
print('cpu', d['cpu_baseline']['value'] if d.get('cpu_baseline') else None)"
echo "[t=$SECONDS s]"
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-e2e --no-traffic > gpurun_out/bench_${TAG}_driverargs.log 2>&1; tail -1 gpurun_out/bench_${TAG}_driverargs.log | cut -c1-260
echo "[t=$SECONDS s]"
bash tools/gpu_profile.sh $TAG 10 "--no-traffic" > gpurun_out/profile_$TAG.log 2>&1; head -28 gpurun_out/rocprof_${TAG}_summary.txt | cut -c1-150
echo "[done t=$SECONDS s]"
