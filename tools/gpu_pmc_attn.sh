#!/bin/bash
# matrix-pipe / VALU occupancy counters of the attention kernel (tools/bench_attn.py), two PMC passes (no tracing domains besides kernel-trace)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  OUT=$REPO/gpurun_out/pmc_attn_$i
  rm -rf $OUT
  ( cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT -- python $REPO/tools/bench_attn.py 5 > $REPO/gpurun_out/pmc_attn_$i.log 2>&1 )
  tail -3 $REPO/gpurun_out/pmc_attn_$i.log | cut -c1-200
done
python - "$REPO/gpurun_out" > gpurun_out/pmc_attn_summary.txt <<'PY'
import csv, sys, glob, collections, re
root = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/pmc_attn_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\s+", " ", r.get("Kernel_Name", ""))
        if "attention_kernel" not in name:
            continue
        grid = r.get("Grid_Size", "?")
        res[grid][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# rocprofv3 --kernel-trace --pmc <set> (two passes) of: python tools/bench_attn.py 5   - attention_kernel<bf16>, per launch (mean)")
print("# SQ_VALU_MFMA_BUSY_CYCLES: cycles summed over the 1024 SIMDs (= 32 per 32x32x16 bf16 MFMA); GRBM_GUI_ACTIVE: active cycles summed over the 8 XCDs")
for grid, d in sorted(res.items(), key=lambda kv: -float(kv[0]) if kv[0].isdigit() else 0):
    m = {k: sum(v) / len(v) for k, v in d.items()}
    line = f"grid {grid:>9}: " + "  ".join(f"{k}={v:.3e}" for k, v in sorted(m.items()))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("GRBM_GUI_ACTIVE"):
        line += f"  => matrix pipe busy {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f} of SIMD-cycles ({m['GRBM_GUI_ACTIVE'] / 8:.0f} cycles per launch)"
    if "SQ_WAVE_CYCLES" in m and "SQ_ACTIVE_INST_VALU" in m:
        line += f"  VALU-issue {m['SQ_ACTIVE_INST_VALU'] / m['SQ_WAVE_CYCLES']:.3f}, waiting-on-issue {m.get('SQ_WAIT_INST_ANY', 0) / m['SQ_WAVE_CYCLES']:.3f}, parked {m.get('SQ_WAIT_ANY', 0) / m['SQ_WAVE_CYCLES']:.3f} of wave-cycles"
    print(line)
PY
cat gpurun_out/pmc_attn_summary.txt | cut -c1-400
find gpurun_out/pmc_attn_* -name "*.csv" -size +4M -delete
