#!/usr/bin/env python
"""Where a skinny GEMM launch spends its time (K22_SKINNY_DEBUG build, K22_SK_DBG=32): s_memrealtime stamps (100 MHz) of every workgroup -
entry, first chunk landed, main loop done, fold barrier passed, stores issued, stores acknowledged - on one time axis."""
import os, sys
import torch
os.environ["K22_SK_DBG"] = "32"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from kandinsky2_amd import _lib
import helpers as hp
L = _lib.lib()
M, T = 162, torch.bfloat16
trace = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
_lib.check(L.k22_debug_set_stream_scratch(trace.data_ptr(), trace.numel() * 8))
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for name, N, K, epi, sk in [("c_qkv", 6144, 2048, 0, 1), ("c_fc", 8192, 2048, 1, 1), ("c_proj", 2048, 2048, 2, 4), ("mlp.c_proj", 2048, 8192, 2, 4)]:
    a = torch.randn(M, K, device="cuda").to(T); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(T); bias = torch.randn(N, device="cuda")
    af = torch.zeros(L.k22_afrag_bytes(M, K) // 2, dtype=T, device="cuda")
    _lib.check(L.k22_afrag_pack(a.data_ptr(), K, af.data_ptr(), M, K, 0, hp.stream()))
    wf = torch.empty_like(w)
    _lib.check(L.k22_stream_repack(w.data_ptr(), wf.data_ptr(), N, 1, K, 0, hp.stream()))
    out = torch.empty(max(M * N, L.k22_afrag_bytes(M, N) // 2), dtype=T, device="cuda"); partial = torch.empty(8 * M * N, dtype=torch.float32, device="cuda")
    for rep in range(3):
        flush.fill_(rep); trace.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(L.k22_skinny_gemm(af.data_ptr(), wf.data_ptr(), bias.data_ptr() if epi != 2 else None, out.data_ptr(), partial.data_ptr(), M, N, N, K, sk, epi, 2 if epi == 1 else 0, N, 3, 2, 0, hp.stream()))
        e1.record(); torch.cuda.synchronize()
    nwg = (N // 64) * 2 * sk
    t = trace[: nwg * 8].reshape(nwg, 8).cpu().double()
    t0 = t[:, 0].min()
    us = (t[:, :6] - t0) / 100.0
    names = ["entry", "chunk0 landed", "loop done", "fold barrier", "stores issued", "stores acked"]
    print(f"{name} N={N} K={K} splitk={sk}: {nwg} workgroups, event time {e0.elapsed_time(e1) * 1e3:.1f} us")
    for i, nm in enumerate(names):
        c = us[:, i]
        print(f"   {nm:14s} min {c.min():6.2f}  median {c.median():6.2f}  max {c.max():6.2f} us after the first workgroup's entry")
    xcc = t[:, 6].long()
    print("   workgroups per XCD:", torch.bincount(xcc, minlength=8).tolist())
