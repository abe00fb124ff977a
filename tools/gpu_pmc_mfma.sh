#!/bin/bash
# Matrix-pipe / VALU / wait counters of EVERY kernel of a bench step (VERDICT r3 missing #3: MFMA utilisation of the conv / linear kernels), two
# PMC passes with --kernel-trace only.  $1 = tag, $2 = extra bench.py arguments (e.g. "--dtype f16x3").  Run on the GPU box through gpurun.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-mfma}
EXTRA=${2:-}
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  OUT=$REPO/gpurun_out/pmc_${TAG}_$i
  rm -rf $OUT
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-parity --no-e2e --no-box --no-loop-graph $EXTRA > $REPO/gpurun_out/pmc_${TAG}_$i.log 2>&1 )
  tail -1 $REPO/gpurun_out/pmc_${TAG}_$i.log | cut -c1-160
done
python - "$REPO/gpurun_out" "$TAG" "$EXTRA" > gpurun_out/pmc_${TAG}_summary.txt <<'PY'
import csv, sys, glob, collections, re
root, tag, extra = sys.argv[1], sys.argv[2], sys.argv[3]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/pmc_{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\s+", " ", r.get("Kernel_Name", ""))
        name = re.sub(r"\(.*$", "", name)
        name = re.sub(r"^void ", "", name)[:64]
        res[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"# rocprofv3 --kernel-trace --pmc <set> (two passes) of: python bench.py --steps 2 --warmup 1 --no-loop-graph {extra}   (per launch, mean over the launches of a kernel)")
print("# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): share of all SIMD-cycles of the launch with the matrix pipe busy")
print("# valu / stall / parked / lds = SQ_ACTIVE_INST_VALU / SQ_WAIT_INST_ANY / SQ_WAIT_ANY / SQ_ACTIVE_INST_LDS over SQ_WAVE_CYCLES (wave-cycles issuing VALU incl. MFMA, stalled on issue, parked on s_waitcnt / barrier, issuing LDS); bank = SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS")
print(f"{'kernel':64s} {'calls':>6} {'cycles':>9} {'mfma_busy':>9} {'valu':>6} {'stall':>6} {'parked':>6} {'lds':>6} {'bank':>6}")
rows = []
for name, d in res.items():
    m = {k: sum(v) / len(v) for k, v in d.items()}
    n = max(len(v) for v in d.values())
    cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8
    tot = cyc * n
    busy = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024) if cyc and "SQ_VALU_MFMA_BUSY_CYCLES" in m else float("nan")
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    f = lambda k: (m.get(k, 0.0) / wc) if wc else float("nan")
    bank = (m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_ACTIVE_INST_LDS"]) if m.get("SQ_ACTIVE_INST_LDS") else float("nan")
    rows.append((tot, f"{name:64s} {n:6d} {cyc:9.0f} {busy:9.3f} {f('SQ_ACTIVE_INST_VALU'):6.3f} {f('SQ_WAIT_INST_ANY'):6.3f} {f('SQ_WAIT_ANY'):6.3f} {f('SQ_ACTIVE_INST_LDS'):6.3f} {bank:6.3f}"))
for _, line in sorted(rows, key=lambda r: -r[0])[:28]:
    print(line)
PY
head -34 gpurun_out/pmc_${TAG}_summary.txt | cut -c1-170
find gpurun_out/pmc_${TAG}_* -name "*.csv" -size +6M -delete
