#!/usr/bin/env python
"""Per-tap cycle budget of the halo conv kernel from in-kernel s_memtime stamps (developer tool)."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kandinsky2_amd import _lib
ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="384,384,96", help="Cin,Cout,H")
a = ap.parse_args()
ci, co, h = (int(v) for v in a.shape.split(","))
B = 2
L = _lib.lib()
x = torch.randn(B, h + 2, h + 2, ci, device="cuda").bfloat16()
w = (torch.randn(co, 9 * ci, device="cuda") * (9 * ci) ** -0.5).bfloat16()
bias = torch.randn(co, device="cuda")
out = torch.empty(B, h, h, co, device="cuda", dtype=torch.bfloat16)
trace = torch.zeros(2, 1024, 4, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    _lib.check(L.k22_debug_conv_trace(x.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), B, h, h, ci, co, co, trace.data_ptr(), st))
torch.cuda.synchronize()
t = trace.cpu()
ntap = int((t[0, :, 3] != 0).sum())
print(f"shape {a.shape}: {ntap} taps traced")
for w_ in range(2):
    tw = t[w_, :ntap].double()
    wait_vm = (tw[:, 1] - tw[:, 0])
    wait_bar = (tw[:, 2] - tw[:, 1])
    body = (tw[:, 3] - tw[:, 2])
    gap = tw[1:, 0] - tw[:-1, 3]
    per = tw[1:, 0] - tw[:-1, 0]
    print(f" wave {'0' if w_ == 0 else '5'}: per-tap period mean {per.mean():.0f} (min {per.min():.0f} max {per.max():.0f}) | vmcnt wait {wait_vm.mean():.0f} | barrier wait {wait_bar.mean():.0f} | issue loads + ds_read + MFMA issue {body.mean():.0f} | loop overhead {gap.mean():.0f}   [s_memtime ticks]")
    print("   first 12 taps (vm, bar, body):", [(int(a_), int(b_), int(c_)) for a_, b_, c_ in zip(wait_vm[:12], wait_bar[:12], body[:12])])
    print(f"   total {tw[-1, 3] - tw[0, 0]:.0f} ticks for {ntap} taps")
