#!/usr/bin/env python
"""Per-shape micro-benchmark of the implicit-GEMM conv / GEMM kernels on the real layer shapes of one UNet
step (C2: B=2, latent 96x96).  Used to pick tile configurations; prints TFLOP/s per shape and the summed time.

    python tools/bench_kernels.py [--lat 96] [--B 2] [--dtype bf16] [--configs auto,128x128,128x64,256x128]
"""
import argparse
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kandinsky2_amd as k22  # noqa: E402
from kandinsky2_amd import _lib  # noqa: E402


def conv_shapes(arch, B, lat):
    """(Cin, Cout, H) -> count, for every 3x3 conv of one forward."""
    shapes = OrderedDict()
    res = {}
    h = lat
    # spatial size per block follows the walk: down blocks halve, up blocks double
    for b in arch.blocks:
        if b[0] != "res":
            continue
        _, pfx, cin, cout, ud = b
        ho = h // 2 if ud == 1 else (h * 2 if ud == 2 else h)
        for (ci, co) in ((cin, cout), (cout, cout)):
            key = (ci, co, ho)
            shapes[key] = shapes.get(key, 0) + 1
        h = ho
    return shapes


def gemm_shapes(arch, B, lat):
    """(M, N, K) -> count for the MFMA GEMMs of one forward: 1x1 skip convs, qkv and proj_out."""
    shapes = OrderedDict()
    h = lat
    for b in arch.blocks:
        if b[0] == "res":
            _, pfx, cin, cout, ud = b
            h = h // 2 if ud == 1 else (h * 2 if ud == 2 else h)
            if cin != cout and ud == 0:
                key = (B * h * h, cout, cin)
                shapes[key] = shapes.get(key, 0) + 1
        elif b[0] == "attn":
            c = b[2]
            for key in ((B * h * h, 3 * c, c), (B * h * h, c, c)):
                shapes[key] = shapes.get(key, 0) + 1
    return shapes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lat", type=int, default=96)
    ap.add_argument("--B", type=int, default=2)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--configs", default="auto")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--stages", type=int, default=-1, help="igemm_stages option (2..4 LDS-DMA pipeline depth)")
    ap.add_argument("--xcd", type=int, default=1, help="XCD-aware block renumbering on/off")
    ap.add_argument("--filter", default="", help="only these conv shapes: 'cin,cout,h;cin,cout,h'")
    ap.add_argument("--cold", type=int, default=1, help="rotate weight copies so weights come from HBM, not the Infinity Cache")
    ap.add_argument("--extra", default="", help="--gemm: additional 'M,N,K;M,N,K' shapes (e.g. the prior's Linears)")
    ap.add_argument("--gemm", action="store_true", help="benchmark the GEMM shapes (1x1 skip, qkv, proj) instead of the 3x3 convs")
    a = ap.parse_args()
    arch = k22.make_arch(k22.MODEL_CONFIG_2_1)
    dt = _lib.K22_BF16 if a.dtype == "bf16" else _lib.K22_F32
    T = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    L = _lib.lib()
    _lib.check(L.k22_set_option(b"igemm_stages", a.stages))
    _lib.check(L.k22_set_option(b"igemm_xcd_remap", a.xcd))
    print("igemm_stages =", a.stages, "xcd_remap =", a.xcd)
    if a.gemm:
        return bench_gemm(a, arch, L, dt, T)
    shapes = conv_shapes(arch, a.B, a.lat)
    if a.filter:
        want = {tuple(int(v) for v in f.split(",")) for f in a.filter.split(";")}
        shapes = OrderedDict((k, v) for k, v in shapes.items() if k in want)
    cfgs = []
    for c in a.configs.split(","):
        if c == "auto":
            cfgs.append((0, 0, 0, 0, False))
        elif c[0] in "hgklpaxystuvmf":   # h256 / h128x2 : halo kernel (g = 64-byte-row variant), BM, split-K; m160x5 = streaming kernel (f160x5: fragment-major weights)
            parts = c[1:].split("x")
            cfgs.append((int(parts[0]), 0, int(parts[1]) if len(parts) > 1 else 0, {"h": 2, "g": 3, "k": 4, "l": 5, "p": 6, "a": 7, "x": 8, "y": 9, "s": 11, "t": 12, "u": 13, "v": 14, "m": 20, "f": 20}[c[0]], c[0] == "f"))
        else:                            # 128x64x8 : generic implicit GEMM
            parts = c.split("x")
            cfgs.append((int(parts[0]), int(parts[1]), int(parts[2]) if len(parts) > 2 else 1, 1, False))
    st = torch.cuda.current_stream().cuda_stream
    tot = {c: 0.0 for c in cfgs}
    totfl = 0.0
    names = a.configs.split(",")
    print(f"{'Cin':>5} {'Cout':>5} {'H':>3} {'cnt':>3} {'GFLOP':>8} | " + " | ".join(f"{n:>10}" for n in names) + " |  best")
    best_tot = 0.0
    for (ci, co, h), cnt in shapes.items():
        x = torch.randn(a.B, h + 2, h + 2, ci, device="cuda").to(T)
        w = (torch.randn((co + 63) // 64 * 64, 9 * ci, device="cuda") * (9 * ci) ** -0.5).to(T)
        # rotate over enough weight copies to defeat the 256 MB Infinity Cache (in a real step the 2.5 GB of
        # weights stream from HBM once per forward)
        ncopy = max(1, min(24, int(400e6 // (w.numel() * w.element_size())) + 1)) if a.cold else 1
        ws = [w] + [w.clone() for _ in range(ncopy - 1)]
        wf = None
        if any(c[4] for c in cfgs):
            wf = [torch.empty_like(v) for v in ws]
            for v, f in zip(ws, wf):
                _lib.check(L.k22_stream_repack(v.data_ptr(), f.data_ptr(), v.shape[0], 9, ci, dt, st))
        bias = torch.randn(co, device="cuda")
        out = torch.empty(a.B, h, h, co, device="cuda", dtype=T)
        part = torch.empty(16 * a.B * h * h * co + 64, device="cuda")
        fl = 2.0 * a.B * h * h * co * 9 * ci
        totfl += fl * cnt
        row = []
        for c in cfgs:
            it = [0]

            _lib.check(L.k22_set_option(b"conv_algo", c[3]))

            def run():
                w = ws[it[0] % len(ws)]
                _lib.check(L.k22_debug_set_stream_frag(wf[it[0] % len(ws)].data_ptr() if c[4] else None, None))
                it[0] += 1
                _lib.check(L.k22_conv3x3(x.data_ptr(), w.data_ptr(), bias.data_ptr(), None, out.data_ptr(), part.data_ptr(),
                                         a.B, h, h, ci, co, w.shape[0], 0, 0, c[2], c[0], c[1], dt, st))
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.reps
            tot[c] += ms * cnt
            row.append(f"{fl / ms / 1e9:10.1f}")
        bi = max(range(len(row)), key=lambda i: float(row[i]))
        best_tot += fl * cnt / float(row[bi]) / 1e9
        print(f"{ci:5d} {co:5d} {h:3d} {cnt:3d} {fl / 1e9:8.1f} | " + " | ".join(row) + f" |  {names[bi]} {fl / float(row[bi]) / 1e3:7.1f} us")
    print(f"best-per-shape: conv total {best_tot:.3f} ms/step -> {totfl / best_tot / 1e9:.1f} TFLOP/s")
    for c in cfgs:
        print(f"config {c}: conv total {tot[c]:.3f} ms/step -> {totfl / tot[c] / 1e9:.1f} TFLOP/s over {totfl / 1e9:.0f} GFLOP")


def bench_gemm(a, arch, L, dt, T):
    shapes = gemm_shapes(arch, a.B, a.lat)
    cfgs = []
    for c in a.configs.split(","):
        if c == "auto":
            cfgs.append((0, 0, 0, 0, False))
        elif c[0] in "mf":         # m160x3 : weight-streaming kernel, bm = 160 / 288, split-K (f160x3: fragment-major weights)
            parts = c[1:].split("x")
            cfgs.append((int(parts[0]), 0, int(parts[1]) if len(parts) > 1 else 1, 20, c[0] == "f"))
        else:
            parts = c.split("x")   # 128x64x1 generic tile; 256x0x1 / 128x0x2 = gemm8_kernel (bn = 0)
            cfgs.append((int(parts[0]), int(parts[1]), int(parts[2]) if len(parts) > 2 else 1, 10 if int(parts[0]) and not int(parts[1]) else 0, False))
    st = torch.cuda.current_stream().cuda_stream
    tot = {c: 0.0 for c in cfgs}
    totfl = 0.0
    print(f"{'M':>6} {'N':>5} {'K':>5} {'cnt':>3} {'GFLOP':>8} | " + " | ".join(f"{n:>10}" for n in a.configs.split(",")) + " |  us(best)")
    if a.filter:
        want = {tuple(int(v) for v in f.split(",")) for f in a.filter.split(";")}
        shapes = OrderedDict((k, v) for k, v in shapes.items() if k in want)
    for extra in a.extra.split(";") if a.extra else []:
        shapes[tuple(int(v) for v in extra.split(","))] = 1
    for (M, N, K), cnt in shapes.items():
        x = torch.randn(M, K, device="cuda").to(T)
        w = (torch.randn((N + 63) // 64 * 64, K, device="cuda") * K ** -0.5).to(T)
        ncopy = max(1, min(48, int(400e6 // (w.numel() * w.element_size())) + 1)) if a.cold else 1
        ws = [w] + [w.clone() for _ in range(ncopy - 1)]
        wf = None
        if any(c[4] for c in cfgs):
            wf = [torch.empty_like(v) for v in ws]
            for v, f in zip(ws, wf):
                _lib.check(L.k22_stream_repack(v.data_ptr(), f.data_ptr(), v.shape[0], 1, K, dt, st))
        bias = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda", dtype=T)
        part = torch.empty(16 * M * N + 64, device="cuda")
        fl = 2.0 * M * N * K
        totfl += fl * cnt
        row = []
        for c in cfgs:
            _lib.check(L.k22_set_option(b"gemm_algo", c[3]))
            it = [0]

            def run():
                w = ws[it[0] % len(ws)]
                _lib.check(L.k22_debug_set_stream_frag(wf[it[0] % len(ws)].data_ptr() if c[4] else None, None))
                it[0] += 1
                _lib.check(L.k22_gemm(x.data_ptr(), None, w.data_ptr(), bias.data_ptr(), None, out.data_ptr(), part.data_ptr(),
                                      M, N, w.shape[0], K, 0, K, 0, N, N, 0, 0, c[2], c[0], c[1], dt, st))
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.reps
            tot[c] += ms * cnt
            row.append(f"{fl / ms / 1e9:10.1f}")
        print(f"{M:6d} {N:5d} {K:5d} {cnt:3d} {fl / 1e9:8.1f} | " + " | ".join(row) + f" |  {fl / max(float(r) for r in row) / 1e3:7.1f}")
    _lib.check(L.k22_set_option(b"gemm_algo", 0))
    _lib.check(L.k22_debug_set_stream_frag(None, None))
    for c in cfgs:
        print(f"config {c}: gemm total {tot[c]:.3f} ms/step -> {totfl / tot[c] / 1e9:.1f} TFLOP/s over {totfl / 1e9:.0f} GFLOP")


if __name__ == "__main__":
    main()
