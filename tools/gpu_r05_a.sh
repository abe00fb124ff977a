#!/bin/bash
# round 5, first GPU pass: the asymmetric split engine (K22_F16X2) - kernel parity, smoke, plan A/B, and this box's default line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export K22_TUNE_REPS=2
python -m pytest tests/test_x2_gpu.py tests/test_x3_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/a_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/a_smoke.txt 2>&1
for plan in 1 3 0 2; do
  K22_X2_PLAN=$plan timeout 600 python bench.py --dtype f16x2 --no-e2e --no-cpu-baseline --parity-timed-only > gpurun_out/a_x2_plan$plan.json 2> gpurun_out/a_x2_plan$plan.err
done
timeout 600 python bench.py --dtype f16x3 --no-e2e --no-cpu-baseline --parity-timed-only > gpurun_out/a_x3.json 2> gpurun_out/a_x3.err
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/a_default.json 2> gpurun_out/a_default.err
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-e2e > gpurun_out/a_steps20.json 2> gpurun_out/a_steps20.err
tail -3 gpurun_out/a_pytest.txt; tail -8 gpurun_out/a_smoke.txt
for f in gpurun_out/a_x2_plan*.json gpurun_out/a_x3.json gpurun_out/a_default.json gpurun_out/a_steps20.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1]))
    pp=j.get("parity_paths") or {}
    print(sys.argv[1], j["dtype"], j["value"], (j.get("roofline") or {}).get("by_class_ms"), {k:(v.get("steps_per_s"),v.get("final_latent_max_abs"),v.get("final_latent_rms")) for k,v in pp.items() if isinstance(v,dict)}, j.get("gate_holding"), j.get("box"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
