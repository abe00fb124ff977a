#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for rep in 1 2 3; do for lib in libk22hip.so libk22hip_nt.so; do
  v=$(K22_LIB_PATH=$PWD/kandinsky-2_amd/$lib timeout 200 python bench.py --chains 1 --steps 50 --warmup 5 --no-cpu-baseline --no-parity --no-e2e --no-box --no-traffic 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['by_class_ms'])")
  echo "$lib rep $rep: $v"
done; done
for lib in libk22hip.so libk22hip_nt.so; do echo "== $lib prior"; K22_LIB_PATH=$PWD/kandinsky-2_amd/$lib timeout 200 python tools/bench_prior.py 2>&1 | grep -E "^prior forward"; done
