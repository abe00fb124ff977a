#!/usr/bin/env python
"""Where a small GEMM's time goes (developer tool): k22_gemm through gemm8 (gemm_algo = 10) and the generic igemm on the proj_out / qkv shapes
of a C2 step, at the real K and at K = 64 (ONE slab: launch + prologue + epilogue only), with / without bias + residual, back to back in one
stream (HIP events), and with the operands flushed from the caches between launches."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from kandinsky2_amd import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
T = torch.bfloat16
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")


def run(M, N, K, algo, bm, bn, splitk, epi, cold, reps=40, stages=-1):
    a = torch.randn(M, K, device="cuda").to(T)
    w = torch.randn((N + 63) // 64 * 64, K, device="cuda").to(T)
    bias = torch.randn(N, device="cuda") if epi else None
    res = torch.randn(M, N, device="cuda").to(T) if epi else None
    out = torch.empty(M, N, dtype=T, device="cuda")
    part = torch.empty(max(1, splitk) * M * N + 64, dtype=torch.float32, device="cuda")
    _lib.check(L.k22_set_option(b"gemm_algo", algo))
    _lib.check(L.k22_set_option(b"igemm_stages", stages))
    call = lambda: _lib.check(L.k22_gemm(a.data_ptr(), None, w.data_ptr(), _lib.ptr(bias), _lib.ptr(res), out.data_ptr(), part.data_ptr(),
                                         M, N, w.shape[0], K, 0, K, 0, N, N, 0, 0, splitk, bm, bn, _lib.K22_BF16, st))
    for _ in range(3): call()
    torch.cuda.synchronize()
    tot = 0.0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if cold:
        for _ in range(10):
            flush.add_(1)
            e0.record(); call(); e1.record(); torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        us = tot / 10 * 1e3
    else:
        e0.record()
        for _ in range(reps): call()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
    _lib.check(L.k22_set_option(b"gemm_algo", 0))
    _lib.check(L.k22_set_option(b"igemm_stages", -1))
    return us


SHAPES = [(4608, 768, 768, "proj_out 48x48"), (4608, 2304, 768, "qkv 48x48"), (1152, 1152, 1152, "proj_out 24x24"), (1152, 3456, 1152, "qkv 24x24"),
          (288, 1536, 1536, "proj_out 12x12"), (288, 4608, 1536, "qkv 12x12"),
          (32768, 768, 768, "C3 proj_out 64x64"), (32768, 2304, 768, "C3 qkv 64x64"), (8192, 1152, 1152, "C3 proj_out 32x32"), (8192, 3456, 1152, "C3 qkv 32x32"),
          (73728, 768, 768, "VERDICT r5 #7 shape")]
if "--big" in sys.argv:
    SHAPES = SHAPES[6:]
for (M, N, K, name) in SHAPES:
    gf = 2.0 * M * N * K / 1e9
    print(f"{name}: M={M} N={N} K={K}  ({gf:.1f} GFLOP)")
    for algo, bm, bn, stg, label in ((10, 128, 0, -1, "gemm8 BM=128"), (10, 128, 0, 2, "gemm8 128 2/CU"), (10, 128, 0, 3, "gemm8 spec 128"), (10, 128, 0, 4, "spec 128 2/CU"),
                                     (10, 256, 0, -1, "gemm8 BM=256"), (10, 256, 0, 3, "gemm8 spec 256"),
                                     (0, 128, 128, -1, "igemm 128x128"), (0, 128, 64, -1, "igemm 128x64"), (0, 64, 64, -1, "igemm 64x64")):
        try:
            full = run(M, N, K, algo, bm, bn, 1, True, False, stages=stg)
            one = run(M, N, 64, algo, bm, bn, 1, True, False, stages=stg)
            noepi = run(M, N, K, algo, bm, bn, 1, False, False, stages=stg)
            cold = run(M, N, K, algo, bm, bn, 1, True, True, stages=stg)
            print(f"  {label:14s} warm {full:6.1f} us ({gf / full / 1e3:5.2f} PFLOP/s) | K=64 {one:6.1f} | no bias/residual {noepi:6.1f} | cold {cold:6.1f}")
        except Exception as e:
            print(f"  {label:14s} failed: {str(e)[:80]}")
