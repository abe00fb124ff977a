#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for d in 0 1 2 4 6 7 8 16; do echo "== K22_SK_DBG=$d"; K22_SK_DBG=$d timeout 100 python tools/bench_skinny.py --only 3,2 --quick 2>&1 | grep -E "splitk"; done
echo "== warm (no flush)"; timeout 100 python tools/bench_skinny.py --only 3,2 --quick --warm 2>&1 | grep -E "splitk"
