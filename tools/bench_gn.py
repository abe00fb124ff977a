#!/usr/bin/env python
"""GroupNorm32 + SiLU through the C ABI (memset + gn_stats + gn_apply2) on the UNet's shapes (developer tool; run under
rocprofv3 --kernel-trace --stats to see the per-kernel times)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kandinsky2_amd import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
B = 2
for (H, C) in [(96, 384), (96, 768), (48, 768), (48, 1536), (24, 1152), (12, 1536), (12, 3072)]:
    x = torch.randn(B, H, H, C, device="cuda").bfloat16()
    gamma, beta = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    out = torch.empty(B, H + 2, H + 2, C, device="cuda", dtype=torch.bfloat16)
    scratch = torch.empty(L.k22_groupnorm_scratch_bytes(B, C), dtype=torch.uint8, device="cuda")
    run = lambda: _lib.check(L.k22_groupnorm(x.data_ptr(), None, C, 0, B, H, H, gamma.data_ptr(), beta.data_ptr(), None, 0, 1e-5, 1, 0, 1,
                                             scratch.data_ptr(), out.data_ptr(), 0, st))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    mb = (x.numel() + out.numel()) * 2 / 1e6
    print(f"H={H} C={C}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us for memset+stats+apply2  ({mb:.1f} MB in+out of apply)")
