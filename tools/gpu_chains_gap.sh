#!/bin/bash
# R4-3 follow-up: eager two-chain forwards of the split-precision engine with the prologues before the fork (shipped) and inside it (SWAP=3)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export K22_CHAINS=2 K22_AUTOTUNE=0
( echo "== prologues before the fork (shipped form of the eager path)"; timeout 200 python tools/chains_gap_probe.py f16x3 10 0
  echo "== K22_CHAINS_SWAP=3: prologues inside the fork (round 4's first form)"; K22_CHAINS_SWAP=3 timeout 200 python tools/chains_gap_probe.py f16x3 6 0 ) 2>&1 | grep -v "amdgpu.ids" | grep -E "^==|^RESULT|^dtype" | tee gpurun_out/chains_prologue.txt | cut -c1-200
