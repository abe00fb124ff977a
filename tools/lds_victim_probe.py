"""Round-4 debug: which kernel, running on ANOTHER stream, perturbs linear_smallm_kernel (the time-embedding MLP)?  DESIGN.md 9 R4-3.
A victim loop (k22_linear_smallm; every launch must reproduce the bits of a launch that ran alone) runs on one stream while one
candidate kernel loops on a second stream through the C ABI's unit entries.
    python tools/lds_victim_probe.py"""
import os, sys, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import kandinsky2_amd as k22  # noqa: F401  (puts the package on the path for _lib)
from kandinsky2_amd import _lib
from kandinsky2_amd.pack import to_x3
import helpers as hp

L = _lib.lib()
X3, BF16, F32 = _lib.K22_F16X3, _lib.K22_BF16, _lib.K22_F32
dev = "cuda"
g = torch.Generator().manual_seed(0)
def rnd(*s, scale=1.0): return (torch.randn(*s, generator=g) * scale).to(dev)

sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
# ---- victim -------------------------------------------------------------------------------------------------------------------------
Kv, Nv = 384, 1536
xv = rnd(1, Kv).repeat(4, 1).contiguous(); Wv = rnd(Nv, Kv, scale=Kv ** -0.5); bv = rnd(Nv)
outv = torch.zeros(4, Nv, device=dev); badv = torch.zeros((), device=dev, dtype=torch.int64)
rowbad = torch.zeros(4, device=dev, dtype=torch.int64); featbad = torch.zeros(Nv, device=dev, dtype=torch.int64); relerr = torch.zeros((), device=dev)
refv = torch.zeros(4, Nv, device=dev)
def victim(n):
    with torch.cuda.stream(sB):
        for _ in range(n):
            _lib.check(L.k22_linear_smallm(xv.data_ptr(), Wv.data_ptr(), bv.data_ptr(), None, outv.data_ptr(), 4, Nv, Kv, 0, 1, F32, sB.cuda_stream))
            ne = outv != refv
            badv.add_(ne.any()); rowbad.add_(ne.any(1)); featbad.add_(ne.any(0)); relerr.copy_(torch.maximum(relerr, ((outv - refv).abs() / (refv.abs() + 1e-3)).max()))

victim(1); torch.cuda.synchronize(); refv.copy_(outv); torch.cuda.synchronize()
# ---- offenders ------------------------------------------------------------------------------------------------------------------------
B, Cin, Cout, H, W = 2, 768, 768, 48, 48
x = rnd(B, Cin, H, W); w = rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5); bias = rnd(Cout)
def conv_setup(dt):
    T = torch.bfloat16 if dt == BF16 else torch.float32
    xp = hp.nhwc_padded(x, T); wp = hp.pack_conv3(w, T)
    if dt == X3:
        xq = torch.empty_like(xp); _lib.check(L.k22_x3_pack(xp.data_ptr(), xq.data_ptr(), xp.numel(), 1.0, torch.cuda.current_stream().cuda_stream)); xp = xq
        wp = to_x3(wp)
    out = torch.empty(B, H, W, Cout, device=dev, dtype=T)
    part = torch.empty(16 * B * H * W * Cout + 64, device=dev)
    return xp, wp, out, part
cx3, cbf, cf32 = conv_setup(X3), conv_setup(BF16), conv_setup(F32)
def conv(dt, algo, bm, bn=0, splitk=1, stages=-1):
    xp, wp, out, part = cx3 if dt == X3 else (cbf if dt == BF16 else cf32)
    def run():
        _lib.check(L.k22_set_option(b"igemm_stages", stages))
        _lib.check(L.k22_set_option(b"conv_algo", algo))
        _lib.check(L.k22_conv3x3(xp.data_ptr(), wp.data_ptr(), bias.data_ptr(), None, out.data_ptr(), part.data_ptr(), B, H, W, Cin, Cout, wp.shape[0], 0, 0, splitk, bm, bn, dt, sA.cuda_stream))
    return run
M, N, K = B * H * W, 2304, 768
Ag = rnd(M, K); Wg = rnd(N, K, scale=K ** -0.5); wgp = to_x3(hp.pad_rows(Wg)); outg = torch.empty(M, N, device=dev); partg = torch.empty(4 * M * N + 64, device=dev)
def gemm(algo, bm, bn):
    def run():
        _lib.check(L.k22_set_option(b"gemm_algo", algo))
        _lib.check(L.k22_gemm(Ag.data_ptr(), None, wgp.data_ptr(), None, None, outg.data_ptr(), partg.data_ptr(), M, N, wgp.shape[0], K, 0, K, 0, N, N, 0, 0, 1, bm, bn, X3, sA.cuda_stream))
    return run
Hh, T_, S = 12, H * W, 87
qkv = rnd(B * T_, 3 * 64 * Hh); ctx = rnd(B * S, 2 * 64 * Hh); Tkp = (S + T_ + 63) // 64 * 64
kall = torch.zeros(B, Hh, Tkp, 64, device=dev); vtall = torch.zeros(B, Hh, 64, Tkp, device=dev); outa = torch.empty(B * T_, 64 * Hh, device=dev)
def attn():
    _lib.check(L.k22_attention(qkv.data_ptr(), ctx.data_ptr(), kall.data_ptr(), vtall.data_ptr(), outa.data_ptr(), B, Hh, T_, S, X3, sA.cuda_stream))
a0 = x.permute(0, 2, 3, 1).contiguous(); gam, bet = 1 + 0.1 * rnd(Cin), 0.1 * rnd(Cin)
gout = torch.empty(B, H + 2, W + 2, Cin, device=dev); gscr = torch.empty(L.k22_groupnorm_scratch_bytes(B, Cin), dtype=torch.uint8, device=dev)
def gnorm():
    _lib.check(L.k22_groupnorm(a0.data_ptr(), None, Cin, 0, B, H, W, gam.data_ptr(), bet.data_ptr(), None, 0, 1e-5, 1, 0, 1, gscr.data_ptr(), gout.data_ptr(), X3, sA.cuda_stream))

cands = [("none", None), ("bf16 conv generic (1, 128x64)", conv(BF16, 1, 128, 64)), ("fp32 conv generic (1, 128x64)", conv(F32, 1, 128, 64)),
         ("x3 conv generic 128x64 stages 2", conv(X3, 1, 128, 64, 1, 2)), ("x3 conv generic 128x64 stages 3", conv(X3, 1, 128, 64, 1, 3)), ("x3 conv generic 128x64 stages 4", conv(X3, 1, 128, 64, 1, 4)),
         ("x3 conv generic 64x64", conv(X3, 1, 64, 64)), ("x3 conv generic 128x128", conv(X3, 1, 128, 128)), ("fp32 conv generic 128x128", conv(F32, 1, 128, 128)), ("bf16 conv generic 128x128", conv(BF16, 1, 128, 128)), ("bf16 conv spec (algo 11, bm 128)", conv(BF16, 11, 128)), ("x3 conv spec pipelined (11, 128)", conv(X3, 11, 128)),
         ("x3 conv spec (12, 256)", conv(X3, 12, 256)), ("x3 conv spec (11, 256)", conv(X3, 11, 256)), ("x3 conv lock-step (7, 256)", conv(X3, 7, 256)),
         ("x3 conv lock-step (2, 128)", conv(X3, 2, 128)), ("x3 conv generic (1, 128x64)", conv(X3, 1, 128, 64)), ("x3 conv spec split-K 2 (11, 128)", conv(X3, 11, 128, 0, 2)),
         ("x3 gemm8 raw A (10, 128)", gemm(10, 128, 0)), ("x3 gemm generic raw A (0, 128x64)", gemm(0, 128, 64)), ("x3 attention T=2304", attn), ("x3 groupnorm chunk store", gnorm)]
torch.cuda.synchronize()
L.k22_debug_lds_sentinel.restype = C.c_int
L.k22_debug_lds_sentinel.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
rec = torch.zeros(1 + 4 * 256, device=dev, dtype=torch.int32)
def sentinel(nbytes, name, fn):
    rec.zero_(); torch.cuda.synchronize()
    for it in range(60):
        if fn is not None:
            fn()
        _lib.check(L.k22_debug_lds_sentinel(nbytes, 512, 200, rec.data_ptr(), sB.cuda_stream))
    torch.cuda.synchronize()
    r = rec.cpu().numpy().astype("uint32"); n = int(r[0])
    line = f"SENTINEL {nbytes:6d} B  {name:40s} changed words: {n}"
    if n:
        recs = r[1: 1 + 4 * min(n, 255)].reshape(-1, 4)
        offs = recs[:, 1]
        vals = recs[:12, 2].copy().view("float32")
        line += f"  offsets {int(offs.min())}..{int(offs.max())}  first (wg, off, value-as-f32/hex, spin): " + ", ".join(f"({a},{b},{v:.4g}/{c:08x},{d})" for (a, b, c, d), v in zip(recs[:12].tolist(), vals.tolist()))
    print(line, flush=True)
if os.environ.get("SENTINEL"):
    for nb in (6144, 16384):
        for name, fn in cands:
            if any(f in name for f in ("none", "bf16 conv generic (1, 128x64)", "x3 conv generic (1, 128x64)", "fp32 conv generic (1, 128x64)", "x3 conv spec (12, 256)")):
                sentinel(nb, name, fn)
if os.environ.get("DETAIL"):
    fn = dict(cands)["x3 conv generic (1, 128x64)"]
    shown = 0
    for it in range(400):
        fn()
        with torch.cuda.stream(sB):
            _lib.check(L.k22_linear_smallm(xv.data_ptr(), Wv.data_ptr(), bv.data_ptr(), None, outv.data_ptr(), 4, Nv, Kv, 0, 1, F32, sB.cuda_stream))
            snap = outv.clone()
        sB.synchronize()
        ne = (snap != refv)
        if ne.any() and shown < 12:
            idx = ne[0].nonzero().flatten().tolist()
            print(f"launch {it}: rows hit {ne.any(1).tolist()}  features {idx[:40]}  got-ref {[round(float(snap[0, j] - refv[0, j]), 4) for j in idx[:8]]}", flush=True)
            shown += 1
    torch.cuda.synchronize()
if os.environ.get("VICTIMS"):
    # which victims are susceptible?  linear_smallm with 1 / 2 / 4 / 8 rows (different code shapes: the 4-row form uses packed fp32 math)
    fn = dict(cands)["x3 conv generic (1, 128x64)"]
    for Mv in (1, 2, 3, 4, 5, 8):
        xm = rnd(1, Kv).repeat(Mv, 1).contiguous(); om = torch.zeros(Mv, Nv, device=dev); rm = torch.zeros(Mv, Nv, device=dev)
        rb = torch.zeros(Mv, device=dev, dtype=torch.int64)
        with torch.cuda.stream(sB):
            _lib.check(L.k22_linear_smallm(xm.data_ptr(), Wv.data_ptr(), bv.data_ptr(), None, rm.data_ptr(), Mv, Nv, Kv, 0, 1, F32, sB.cuda_stream))
        torch.cuda.synchronize()
        for it in range(150):
            fn()
            with torch.cuda.stream(sB):
                for _ in range(6):
                    _lib.check(L.k22_linear_smallm(xm.data_ptr(), Wv.data_ptr(), bv.data_ptr(), None, om.data_ptr(), Mv, Nv, Kv, 0, 1, F32, sB.cuda_stream))
                    rb.add_((om != rm).any(1))
        torch.cuda.synchronize()
        print(f"victim linear_smallm M = {Mv}: launches (of 900) with a wrong element, per row: {rb.tolist()}", flush=True)
flt = os.environ.get("CANDS")
for name, fn in cands:
    if flt and not any(f in name for f in flt.split(",")):
        continue
    badv.zero_(); rowbad.zero_(); featbad.zero_(); relerr.zero_()
    try:
        for it in range(150):
            if fn is not None:
                fn()
            victim(6)
        torch.cuda.synchronize()
        print(f"{name:45s} victim launches with rows differing: {int(badv.item())} of {150 * 6}  per row {rowbad.tolist()}  features hit {int((featbad > 0).sum())} of {Nv} (first {featbad.nonzero().flatten()[:8].tolist()})  max rel err {relerr.item():.2e}", flush=True)
    except Exception as e:
        print(f"{name:45s} FAILED: {str(e)[:150]}", flush=True)
        torch.cuda.synchronize()
_lib.check(L.k22_set_option(b"conv_algo", 0)); _lib.check(L.k22_set_option(b"gemm_algo", 0)); _lib.check(L.k22_set_option(b"igemm_stages", -1))
