#!/bin/bash
# HBM traffic of the conv kernels from PMC counters (separate passes for FETCH_SIZE and WRITE_SIZE), run on the GPU box.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-pmc}
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
for CTR in FETCH_SIZE WRITE_SIZE; do
  OUT=$REPO/gpurun_out/pmc_${TAG}_$CTR
  rm -rf $OUT
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d $OUT -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-parity --no-e2e --no-box > $REPO/gpurun_out/bench_pmc_${TAG}_$CTR.log 2>&1 )
  tail -2 $REPO/gpurun_out/bench_pmc_${TAG}_$CTR.log | cut -c1-300
  find $OUT -name "*.csv" | head
done
python - "$REPO/gpurun_out" "$TAG" > gpurun_out/pmc_${TAG}_summary.txt <<'PY'
import csv, sys, glob, collections, re
root, tag = sys.argv[1], sys.argv[2]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{root}/pmc_{tag}_{ctr}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\s+", " ", r.get("Kernel_Name", ""))[:60]
            if r.get("Counter_Name") == ctr:
                res[name][ctr].append(float(r["Counter_Value"]))
print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of: python bench.py --steps 3 --warmup 1")
print("# raw counter units as reported by rocprofv3 (KB); MI355X_MICROARCH.md: FETCH_SIZE under-reports wide streaming reads by 2x on gfx950")
print(f"{'kernel':60s} {'calls':>6} {'FETCH avg':>12} {'FETCH sum':>14} {'WRITE avg':>12} {'WRITE sum':>14}")
for name, d in sorted(res.items(), key=lambda kv: -sum(kv[1].get("FETCH_SIZE", [0]))):
    fz, wz = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
    n = max(len(fz), len(wz))
    print(f"{name:60s} {n:6d} {sum(fz)/max(1,len(fz)):12.1f} {sum(fz):14.1f} {sum(wz)/max(1,len(wz)):12.1f} {sum(wz):14.1f}")
PY
head -30 gpurun_out/pmc_${TAG}_summary.txt | cut -c1-200
find gpurun_out/pmc_${TAG}_* -name "*.csv" -size +8M -delete
