"""Turns a rocprofv3 `*kernel_stats.csv` into the per-kernel summary kept under profiles/:  python tools/rocprof_summary.py <csv> "<command>" """
import csv
import re
import sys

f, cmd = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# rocprofv3 --kernel-trace --stats of: {cmd}")
print(f"# total kernel time {tot / 1e6:.3f} ms over all launches")
print(f"{'calls':>8} {'total_ms':>10} {'avg_us':>9} {'pct':>6}  kernel")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    name = re.sub(r"\s+", " ", r["Name"])[:150]
    print(f"{int(r['Calls']):8d} {float(r['TotalDurationNs']) / 1e6:10.3f} {float(r['AverageNs']) / 1e3:9.2f} {float(r['Percentage']):6.2f}  {name}")
