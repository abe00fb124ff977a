#!/usr/bin/env python
"""Attention kernel on the three UNet levels (developer tool): time per call and achieved TFLOP/s."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kandinsky2_amd import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
S, B = 87, 2
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for (T, C) in [(2304, 768), (576, 1152), (144, 1536)]:
    H = C // 64
    Tkp = (S + T + 63) // 64 * 64
    qkv = torch.randn(B * T, 3 * C, device="cuda").bfloat16()
    ctx = torch.randn(B * S, 2 * C, device="cuda").bfloat16()
    kall = torch.zeros(B, H, Tkp, 64, device="cuda", dtype=torch.bfloat16)
    vt = torch.zeros(B, H, 64, Tkp, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(B * T, C, device="cuda", dtype=torch.bfloat16)
    run = lambda: _lib.check(L.k22_attention(qkv.data_ptr(), ctx.data_ptr(), kall.data_ptr(), vt.data_ptr(), out.data_ptr(), B, H, T, S, 0, st))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    fl = 4.0 * B * H * T * (T + S) * 64
    print(f"T={T} heads={H}: {us:7.1f} us per kv_pack+attention  ({fl / us / 1e6:6.1f} TFLOP/s)")
