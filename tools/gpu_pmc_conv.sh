#!/bin/bash
# SQ counters of the conv kernels on three representative shapes (run on the GPU box)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
TAG=${1:-sq}
OUT=$REPO/gpurun_out/pmc_$TAG
rm -rf $OUT
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT -- python $REPO/tools/bench_kernels.py --reps 2 --configs ${PMC_CONFIGS:-h256x1,h128x1} --filter "384,384,96;768,768,48;768,768,96" > $REPO/gpurun_out/pmc_${TAG}.log 2>&1 )
tail -5 $REPO/gpurun_out/pmc_${TAG}.log | cut -c1-200
python - "$OUT" > gpurun_out/pmc_${TAG}_summary.txt <<'PY'
import csv, sys, glob, collections, re
root = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\s+", " ", r.get("Kernel_Name", ""))[:70]
        if "conv3_halo" not in name: continue
        name = re.sub(r"void |unsigned short", "", name)
        key = (name, r.get("Grid_Size", r.get("Grid_Size_X", "")))
        rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
ctrs = ["SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_VALU_MFMA_BUSY_CYCLES","SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE"]
print("kernel / grid | " + " | ".join(c.replace("SQ_","") for c in ctrs))
for key, d in rows.items():
    vals = [sum(d.get(c,[0]))/max(1,len(d.get(c,[0]))) for c in ctrs]
    wc = vals[0] or 1
    print(f"{key[0]} g={key[1]} | " + " | ".join(f"{v:.3e}" for v in vals))
    print(f"    wait_any/wave_cycles={vals[2]/wc:.2f} wait_inst/wave_cycles={vals[3]/wc:.2f} active/wave_cycles={vals[4]/wc:.2f} mfma_busy/(busy_cycles*4)={vals[5]/max(1,vals[1])/4:.2f} lds_conflict/lds_active={vals[6]/max(1,vals[7]):.3f}")
PY
cat gpurun_out/pmc_${TAG}_summary.txt | cut -c1-250
find $OUT -name "*.csv" -size +8M -delete
