#!/bin/bash
# R4-3 follow-up: which slot of the second-enqueued kid's workspace is the first to differ between a good and a bad forward?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export K22_CHAINS=2 K22_CHAINS_X3=1 K22_AUTOTUNE=0 K22_DBG_SLOTS=1 WS_DIFF=1 K22_CHAINS_SWAP=1
timeout 600 python tools/chains_gap_probe.py f16x3 2 0 2>&1 | grep -v "amdgpu.ids" > gpurun_out/chains_wsdiff.txt
grep -v "^k22 slot" gpurun_out/chains_wsdiff.txt | cut -c1-300
grep "^k22 slot" gpurun_out/chains_wsdiff.txt | head -24
