#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for s in 384,384,96 768,768,96 768,768,48 1152,1152,48; do timeout 120 python tools/conv_trace.py --shape $s 2>&1 | grep -v amdgpu.ids; done
