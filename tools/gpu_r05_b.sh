#!/bin/bash
# round 5, second GPU pass: pipelined fused-skip loop -> kernel parity, re-measured tile lines (fused-skip convs, x2's own), benches, full suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export K22_TUNE_REPS=2
python -m pytest tests/test_kernels_gpu.py tests/test_x3_gpu.py tests/test_x2_gpu.py tests/test_stream_gpu.py -x -q -m gpu -k "skip" 2>&1 | tail -15 > gpurun_out/b_pytest_skip.txt
tail -3 gpurun_out/b_pytest_skip.txt
if ! grep -q " passed" gpurun_out/b_pytest_skip.txt || grep -q "failed" gpurun_out/b_pytest_skip.txt; then echo "fused-skip kernel tests FAILED: stopping"; exit 3; fi
# the same box, before re-tuning: the bf16 line with the OLD table lines (hal6 for the fused-skip convs) but the new skip loop
timeout 300 python bench.py --no-cpu-baseline --no-parity --no-e2e > gpurun_out/b_bf16_oldtable.json 2> gpurun_out/b_bf16_oldtable.err
unset K22_TUNE_REPS
( time python tools/make_tile_table.py --fused-skip gpurun_out/tiles_a.txt ) > gpurun_out/b_tune_skip.log 2>&1
( time K22_TILE_TABLE=gpurun_out/tiles_a.txt python tools/make_tile_table.py --x2-only gpurun_out/tiles_b.txt ) > gpurun_out/b_tune_x2.log 2>&1
tail -2 gpurun_out/b_tune_skip.log gpurun_out/b_tune_x2.log
if [ -s gpurun_out/tiles_b.txt ]; then cp gpurun_out/tiles_b.txt kandinsky-2_amd/tiles_gfx950.txt; fi
export K22_TUNE_REPS=2
timeout 900 python bench.py --no-cpu-baseline --tuning-report gpurun_out/b_tuning.txt > gpurun_out/b_default.json 2> gpurun_out/b_default.err
timeout 300 python bench.py --dtype f16x2 --no-cpu-baseline --no-e2e --parity-timed-only --tuning-report gpurun_out/b_tuning_x2.txt > gpurun_out/b_x2.json 2> gpurun_out/b_x2.err
timeout 300 python bench.py --dtype f16x3 --no-cpu-baseline --no-e2e --parity-timed-only > gpurun_out/b_x3.json 2> gpurun_out/b_x3.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/b_smoke.txt 2>&1; tail -7 gpurun_out/b_smoke.txt
K22_PARITY_REPORT=gpurun_out/b_parity.json timeout 1300 python -m pytest tests -x -q -m gpu --durations=25 > gpurun_out/b_pytest.txt 2>&1
tail -35 gpurun_out/b_pytest.txt
for f in gpurun_out/b_bf16_oldtable.json gpurun_out/b_default.json gpurun_out/b_x2.json gpurun_out/b_x3.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1]))
    pp=j.get("parity_paths") or {}
    print(sys.argv[1], j["dtype"], j["value"], (j.get("roofline") or {}).get("by_class_ms"), {k:(v.get("steps_per_s"),v.get("final_latent_max_abs"),v.get("final_latent_rms")) for k,v in pp.items() if isinstance(v,dict)}, j.get("box"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
