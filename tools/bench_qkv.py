#!/usr/bin/env python
"""qkv projection (IG_OUT_QKV) of the three attention levels: generic igemm tiles vs gemm8_kernel (developer tool)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kandinsky2_amd import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
S, B = 87, 2
for (T, C) in [(2304, 768), (576, 1152), (144, 1536)]:
    H = C // 64
    Tkp = (S + T + 63) // 64 * 64
    x = torch.randn(B * T, C, device="cuda").bfloat16()
    w = (torch.randn(3 * C, C, device="cuda") * C ** -0.5).bfloat16()
    bias = torch.randn(3 * C, device="cuda")
    q = torch.empty(B * T, C, device="cuda", dtype=torch.bfloat16)
    kall = torch.zeros(B, H, Tkp, 64, device="cuda", dtype=torch.bfloat16)
    vt = torch.zeros(B, H, 64, Tkp, device="cuda", dtype=torch.bfloat16)
    row = []
    for name, algo, bm, bn, stg in [("gen128x64", 0, 128, 64, -1), ("gen64x64", 0, 64, 64, -1), ("gem8-256", 10, 256, 0, -1), ("gem8-128", 10, 128, 0, -1), ("gem8-128s2", 10, 128, 0, 2)]:
        _lib.check(L.k22_set_option(b"gemm_algo", algo)); _lib.check(L.k22_set_option(b"igemm_stages", stg))
        run = lambda: _lib.check(L.k22_qkv_project(x.data_ptr(), w.data_ptr(), bias.data_ptr(), q.data_ptr(), kall.data_ptr(), vt.data_ptr(), B, H, T, S, C, bm, bn, 0, st))
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        row.append(f"{name} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
    print(f"T={T} C={C}: " + " | ".join(row))
_lib.check(L.k22_set_option(b"gemm_algo", 0)); _lib.check(L.k22_set_option(b"igemm_stages", -1))
