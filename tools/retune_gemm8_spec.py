#!/usr/bin/env python
"""Round 6: gemm8_spec_kernel (p.stages 3 / 4) joined the candidates of the plain GEMMs.  It gives the lock-step kernel's bits for the same
(bm, splitk), so the shipped table is updated IN PLACE: for every 16-bit gemm8 line the four ring / specialisation forms are timed here with
bm and splitk held at the line's values (plain bias + residual epilogue; warm back-to-back and cold after a 512 MB flush), and only the line's
`stages` column changes, and only where the specialised form wins by >= 3 % of the mean of the two protocols.  Engines that use the new table
therefore produce the same bits as with the old one.

    python tools/retune_gemm8_spec.py kandinsky-2_amd/tiles_gfx950.txt gpurun_out/tiles_gfx950_spec.txt
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kandinsky2_amd import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
TD = {0: (torch.bfloat16, _lib.K22_BF16), 2: (torch.float16, _lib.K22_F16)}


def time_cfg(M, N, K, bm, splitk, stages, dt):
    T, code = TD[dt]
    a = torch.randn(M, K, device="cuda").to(T)
    w = torch.randn((N + 127) // 128 * 128, K, device="cuda").to(T)
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda").to(T)
    out = torch.empty(M, N, dtype=T, device="cuda")
    part = torch.empty(max(1, splitk) * M * N + 64, dtype=torch.float32, device="cuda")
    _lib.check(L.k22_set_option(b"gemm_algo", 10))
    _lib.check(L.k22_set_option(b"igemm_stages", stages if stages else -1))
    try:
        call = lambda: _lib.check(L.k22_gemm(a.data_ptr(), None, w.data_ptr(), bias.data_ptr(), res.data_ptr(), out.data_ptr(), part.data_ptr(),
                                             M, N, w.shape[0], K, 0, K, 0, N, N, 0, 0, splitk, bm, 0, code, st))
        for _ in range(3): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): call()
        e1.record(); torch.cuda.synchronize()
        warm = e0.elapsed_time(e1) / 30 * 1e3
        tot = 0.0
        for _ in range(8):
            flush.add_(1)
            e0.record(); call(); e1.record(); torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        return 0.5 * (warm + tot / 8 * 1e3)
    finally:
        _lib.check(L.k22_set_option(b"gemm_algo", 0))
        _lib.check(L.k22_set_option(b"igemm_stages", -1))


src, dst = sys.argv[1], sys.argv[2]
changed = seen = 0
with open(src) as f, open(dst, "w") as g:
    for line in f:
        v = line.split()
        if line.startswith("#") or len(v) < 16 or not (v[0] in ("0", "2") and v[1] == "1" and v[10] == "10"):
            g.write(line); continue
        dt, M, N, K, bm, splitk, s_old = int(v[0]), int(v[2]), int(v[3]), int(v[4]), int(v[11]), int(v[13]), int(v[14])
        cands = [0, 3] + ([2, 4] if bm == 128 else [])
        seen += 1
        try:
            t = {s: min(time_cfg(M, N, K, bm, splitk, s, dt) for _ in range(2)) for s in cands}
        except Exception as e:
            print("skip", line.strip(), str(e)[:60]); g.write(line); continue
        best = min(t, key=t.get)
        ref = t.get(s_old, t[0])
        if best in (3, 4) and t[best] <= 0.97 * ref:
            v[14] = str(best); changed += 1
            print(f"M={M:6d} N={N:5d} K={K:5d} bm={bm} splitk={splitk}: stages {s_old} -> {best}   " + "  ".join(f"{s}:{t[s]:.1f}" for s in cands), flush=True)
            g.write(" ".join(v) + "\n")
        else:
            g.write(line)
print(f"{seen} 16-bit gemm8 lines timed, {changed} switched to gemm8_spec_kernel")
