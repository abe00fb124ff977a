#!/usr/bin/env python
"""profiles/<round>_pmc_summary.txt (tools/gpu_pmc.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, of
`python bench.py --steps 3 --warmup 1`) -> profiles/pmc_conv.json, the per-launch HBM traffic of the 3x3-convolution kernel class that
bench.py quotes in roofline.traffic (labelled there as "not live").

    python tools/pmc_to_json.py profiles/r02_pmc_summary.txt r02

FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: gfx950 tallies the 128-byte fabric requests of wide streaming reads at
64 bytes); WRITE_SIZE is used as reported; both are in KB.  The algorithmic bytes per launch are the bench's own figure:
(input + output activations + weights of every 3x3 convolution of one forward) / launches.
"""
import json
import re
import sys

src, rnd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r02")
fetch = write = 0.0
calls = 0
for line in open(src):
    if not (("conv3_halo" in line and "kernel<" in line) or "stream_kernel<" in line):   # the 3x3-convolution launches of a forward
        continue
    m = re.match(r"(.{60})\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", line)
    if not m:
        continue
    calls += int(m.group(2))
    fetch += float(m.group(4))
    write += float(m.group(6))
# bench.py --steps 3 --warmup 1 runs 1 (engine init) + 1 + 3 forwards + the split between instantiations is irrelevant here
launches_per_forward = 72
forwards = calls / launches_per_forward
out = {
    "round": rnd,
    "source": f"{src} (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, python bench.py --steps 3 --warmup 1: {calls} 3x3-convolution launches = {forwards:.1f} forwards)",
    "kernel": "conv3_halo_kernel / conv3_halo_spec_kernel / stream_kernel (all instantiations: the 72 3x3-convolution launches of a forward)",
    "launches_per_step": float(launches_per_forward),
    "fetch_bytes_per_step": fetch * 1024 * 2 / forwards,
    "write_bytes_per_step": write * 1024 / forwards,
    "correction": "FETCH_SIZE (KB) x 1024 x 2: gfx950 tallies 128-B fabric requests at 64 B for wide streaming reads (MI355X_MICROARCH.md, HBM section); "
                  "WRITE_SIZE (KB) x 1024 uncorrected; both count fabric (L2-miss) traffic, Infinity-Cache hits included",
    "hbm_bytes_per_launch": (fetch * 2048 + write * 1024) / calls,
    "algorithmic_bytes_per_launch": 2.84e9 / launches_per_forward,   # 1.03 GB activations + 1.80 GB weights per forward
}
json.dump(out, open("profiles/pmc_conv.json", "w"), indent=1)
print(json.dumps(out, indent=1))
