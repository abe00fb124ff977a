#!/usr/bin/env python
"""Kernel-level timing of the skinny-M weight-streaming GEMM (csrc/skinny.hip) on the prior's four Linears (M = 162), every tile
configuration, cold weights (a 512-MB buffer is rewritten between launches to evict the Infinity Cache), HIP events on the launch stream.
Prints us per launch and TB/s of weights.  Usage: python tools/bench_skinny.py [--m 162]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from kandinsky2_amd import _lib
import helpers as hp

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=162)
ap.add_argument("--reps", type=int, default=8)
ap.add_argument("--only", default="")
ap.add_argument("--warm", action="store_true")
ap.add_argument("--quick", action="store_true")
args = ap.parse_args()
L = _lib.lib()
M = args.m
T = torch.bfloat16
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
SHAPES = [("c_qkv", 6144, 2048, 0, [1]), ("c_qkv part", 6144, 2048, 2, [2, 3, 4]), ("c_fc", 8192, 2048, 1, [1]), ("c_proj", 2048, 2048, 2, [1, 2, 4, 8]), ("mlp.c_proj", 2048, 8192, 2, [1, 2, 4, 8])]
CFGS = [(6, 1), (3, 4), (3, 2), (3, 1), (2, 2), (2, 1)]
if args.only:
    CFGS = [tuple(int(v) for v in args.only.split(","))]
if args.quick:
    SHAPES = [(n, N, K, e, [sks[0] if len(sks) == 1 else 4]) for n, N, K, e, sks in SHAPES]
print(f"# skinny GEMM, M = {M}, bf16, cold weights; us per launch (min of {args.reps}) | TB/s of weights")
for name, N, K, epi, sks in SHAPES:
    g = torch.Generator().manual_seed(1)
    a = (torch.randn(M, K, generator=g)).cuda().to(T)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).cuda().to(T)
    bias = torch.randn(N, generator=g).cuda()
    af = torch.zeros(L.k22_afrag_bytes(M, K) // 2, dtype=T, device="cuda")
    _lib.check(L.k22_afrag_pack(a.data_ptr(), K, af.data_ptr(), M, K, 0, hp.stream()))
    wf = torch.empty_like(w)
    _lib.check(L.k22_stream_repack(w.data_ptr(), wf.data_ptr(), N, 1, K, 0, hp.stream()))
    out = torch.empty(max(M * N, L.k22_afrag_bytes(M, N) // 2), dtype=T, device="cuda")
    partial = torch.empty(8 * M * N, dtype=torch.float32, device="cuda")
    for sk in sks:
        row = []
        for mt, nb in CFGS:
            if (-(-M // 32)) < mt and mt != 6:
                pass
            best = 1e9
            for rep in range(args.reps + 1):
                if not args.warm:
                    flush.fill_(rep & 255)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.check(L.k22_skinny_gemm(af.data_ptr(), wf.data_ptr(), bias.data_ptr() if epi != 2 else None, out.data_ptr(), partial.data_ptr(),
                                             M, N, N, K, sk, epi, 2 if epi == 1 else 0, N, mt, nb, 0, hp.stream()))
                e1.record()
                torch.cuda.synchronize()
                if rep:
                    best = min(best, e0.elapsed_time(e1) * 1e3)
            row.append(f"({mt},{nb}) {best:6.1f} us {N * K * 2 / best / 1e6:5.2f}")
        print(f"{name:10s} N={N:5d} K={K:5d} splitk={sk}: " + " | ".join(row))
# the fused finish + LayerNorm and the small attention at the prior's shape
x = torch.randn(M, 2048, device="cuda"); part = torch.randn(4, M, 2048, device="cuda"); b = torch.randn(2048, device="cuda")
y = torch.zeros(L.k22_afrag_bytes(M, 2048) // 2, dtype=T, device="cuda")
qkv = torch.randn(M, 6144, device="cuda").to(T); att = torch.zeros(L.k22_afrag_bytes(M, 2048) // 2, dtype=T, device="cuda")
valid = torch.ones(2, 77, device="cuda")
for what, fn in (("finish_ln splitk 4", lambda: L.k22_finish_ln(part.data_ptr(), 4, b.data_ptr(), x.data_ptr(), 2048, b.data_ptr(), b.data_ptr(), y.data_ptr(), M, 2048, 1e-5, 0, hp.stream())),
                 ("small_attention 2x32x81", lambda: L.k22_small_attention(qkv.data_ptr(), None, 0, None, att.data_ptr(), 1, 2, 32, 81, 1, valid.data_ptr(), 77, 0, hp.stream()))):
    if M != 162 or args.quick:
        break
    best = 1e9
    for rep in range(args.reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _lib.check(fn()); e1.record(); torch.cuda.synchronize()
        if rep:
            best = min(best, e0.elapsed_time(e1) * 1e3)
    print(f"{what}: {best:.1f} us")
