#!/bin/bash
# last validation of round 5 (library built without packed-fp32 instructions): two-stream probe on the shipped library, bench lines, suite, smoke
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export CANDS="none,x3 conv generic (1, 128x64),bf16 conv generic (1, 128x64),x3 gemm generic"
echo "== shipped library of the final commit (every object built with -packed-fp32-ops)" > gpurun_out/probe_pk_final.txt
VICTIMS=1 timeout 300 python tools/lds_victim_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/probe_pk_final.txt
cat gpurun_out/probe_pk_final.txt
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/last_default.json 2> gpurun_out/last_default.err
timeout 200 python bench.py --dtype f16x2 --no-cpu-baseline --no-e2e --parity-timed-only > gpurun_out/last_x2.json 2> gpurun_out/last_x2.err
for f in gpurun_out/last_default.json gpurun_out/last_x2.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1])); pp=j.get("parity_paths") or {}
    print(sys.argv[1], j["dtype"], j["value"], (j.get("roofline") or {}).get("by_class_ms"), {k:(v.get("steps_per_s"),v.get("final_latent_max_abs")) for k,v in pp.items() if isinstance(v,dict)}, (j.get("box") or {}).get("calibration_gemm_tflops"), (j.get("e2e") or {}).get("images_per_sec"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
bash tools/gpu_r05_suite.sh
