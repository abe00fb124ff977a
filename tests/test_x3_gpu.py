"""GPU parity of the SPLIT-PRECISION arithmetic (K22_F16X3, include/k22.h): every MFMA kernel that has an x3 instantiation, called
through the C ABI, against fp64 references of the same op on the same fp32 operands.

What is asserted is that three fp16 MFMAs per product (hi.hi + hi.lo + lo.hi on fp16 (hi, lo) operand pairs) give fp32-class results:
tolerance 1e-5 of the output scale (the fp16 engine misses it by 50x, the exact-fp32 MFMA engine sits at ~1e-6 under the same
metric), i.e. the arithmetic that carries BASELINE.json's 1e-3 final-latent gate (tests/test_full_size_gpu.py) at 16-bit MFMA rate.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import helpers as hp
from kandinsky2_amd import _lib
from kandinsky2_amd.pack import to_x3

pytestmark = pytest.mark.gpu
X3 = _lib.K22_F16X3
TOL = 1e-5


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def close64(out, ref64, what="", tol=TOL):
    scale = ref64.abs().max().item() + 1e-9
    err = (out.double() - ref64).abs().max().item()
    assert np.isfinite(err) and err <= tol * scale, f"{what}: max|d|={err:.3e} tol={tol * scale:.3e} scale={scale:.3f}"


def x3_pack(t, scale=1.0):
    """fp32 cuda tensor -> x3 chunks through the C ABI (k22_x3_pack); same shape, float32-typed bits"""
    t = t.contiguous()
    out = torch.empty_like(t)
    _lib.check(_lib.lib().k22_x3_pack(t.data_ptr(), out.data_ptr(), t.numel(), scale, hp.stream()))
    return out


def x3_unpack(t):
    """x3 chunks -> the value the MFMAs see (hi + lo), float64"""
    h = t.contiguous().view(torch.float16).double()
    g = h.view(*t.shape[:-1], t.shape[-1] // 8, 2, 8)      # groups of eight: [hi x8 | lo x8]
    return (g[..., 0, :] + g[..., 1, :]).reshape(t.shape)


def test_x3_pack_matches_the_python_packer_and_carries_23_bits():
    x = torch.cat([rnd(4096, seed=1), rnd(4096, seed=2) * 1e-3, rnd(4096, seed=3) * 300, rnd(4096, seed=4) * 1e-6]).view(32, 512)
    a, b = x3_pack(x), to_x3(x, 1.0)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))          # device kernel == pack.py (what the arena holds)
    aw, bw = x3_pack(x, 256.0), to_x3(x)
    assert torch.equal(aw.view(torch.int32), bw.view(torch.int32))
    err = (x3_unpack(a) - x.double()).abs()
    bound = torch.maximum(x.double().abs() * 2.0 ** -22, torch.full_like(err, 2.0 ** -24))   # 2^-23 relative (rne twice) / fp16 subnormal floor
    assert (err <= bound).all()
    big = x.abs() >= 0.25
    assert (err[big] <= x.double().abs()[big] * 2.0 ** -22).all()


@pytest.mark.parametrize("M,N,K,bm,bn,splitk", [
    (256, 256, 128, 128, 128, 1), (300, 192, 192, 128, 64, 1), (77, 768, 1024, 64, 64, 1), (128, 128, 64, 64, 128, 1),
    (288, 320, 1152, 128, 64, 4), (2, 1536, 384, 64, 64, 2), (1000, 8, 384, 128, 64, 1), (333, 200, 320, 0, 0, 0),
])
@pytest.mark.parametrize("gemm8", [0, 1])
def test_gemm_x3(M, N, K, bm, bn, splitk, gemm8):
    """generic implicit-GEMM kernel and gemm8 with a RAW fp32 A operand (split while the fragments are read), x3-chunk weights"""
    if gemm8 and N < 128:
        pytest.skip("gemm8 needs N >= 128")
    A, W = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    wp = to_x3(hp.pad_rows(W))
    out = torch.empty(M, N, device="cuda")
    partial = torch.empty(max(1, splitk if splitk else 16) * M * N + 64, device="cuda")
    _lib.check(_lib.lib().k22_set_option(b"gemm_algo", 10 if gemm8 else 0))
    try:
        _lib.check(_lib.lib().k22_gemm(A.data_ptr(), None, wp.data_ptr(), bias.data_ptr(), res.data_ptr(), out.data_ptr(), partial.data_ptr(),
                                       M, N, wp.shape[0], K, 0, K, 0, N, N, 0, 0, splitk, (256 if M >= 256 else 128) if gemm8 else bm,
                                       0 if gemm8 else bn, X3, hp.stream()))
    finally:
        _lib.check(_lib.lib().k22_set_option(b"gemm_algo", 0))
    close64(out, A.double() @ W.double().T + bias.double() + res.double(), f"x3 gemm {M}x{N}x{K}")


def test_gemm_x3_asymmetric_layout_and_small_operands():
    """A = I with an asymmetric W: any transposition of the fragment / accumulator maps shows; W spans 1e-4 .. 8: the lo halves matter"""
    K = 128
    A = torch.eye(K).cuda()
    W = ((torch.arange(64 * K, dtype=torch.float32).reshape(64, K) % 251 - 125) / 16).cuda() * torch.logspace(-4, 0, K).cuda()
    wp = to_x3(W)
    out = torch.empty(K, 64, device="cuda")
    _lib.check(_lib.lib().k22_gemm(A.data_ptr(), None, wp.data_ptr(), None, None, out.data_ptr(), None, K, 64, 64, K, 0, K, 0, 64, 64, 0, 0, 1,
                                   0, 0, X3, hp.stream()))
    err = (out.double() - W.double().T).abs()
    assert (err <= W.double().T.abs() * 2.0 ** -21 + 1e-9).all()


def _conv_x3(x, w, bias, res, splitk, bm, bn, algo, out_mode=0, stats=False):
    B, Cin, H, W_ = x.shape
    Cout = w.shape[0]
    xp = x3_pack(hp.nhwc_padded(x, torch.float32))
    wp = to_x3(hp.pack_conv3(w, torch.float32))
    r = None if res is None else res.permute(0, 2, 3, 1).contiguous()
    out = torch.empty((B, Cout, H, W_) if out_mode == _lib.OUT_NCHW_F32 else (B, H, W_, Cout), device="cuda")
    partial = torch.empty(max(1, splitk if splitk else 16) * B * H * W_ * Cout + 64, device="cuda")
    st = None
    _lib.check(_lib.lib().k22_set_option(b"conv_algo", algo))
    try:
        if stats:
            cap = B * (H * (W_ + 2) // 16 + 2)
            sbuf = torch.full((cap, Cout, 2), float("nan"), device="cuda")
            rpi = C.c_int(0)
            _lib.check(_lib.lib().k22_conv3x3_gnstats(xp.data_ptr(), wp.data_ptr(), _lib.ptr(bias), _lib.ptr(r), out.data_ptr(), partial.data_ptr(),
                                                      B, H, W_, Cin, Cout, wp.shape[0], splitk, bm, bn, sbuf.data_ptr(), cap, C.byref(rpi), X3,
                                                      hp.stream()))
            st = sbuf[: B * rpi.value].view(B, rpi.value, Cout, 2).double().sum(1)
        else:
            _lib.check(_lib.lib().k22_conv3x3(xp.data_ptr(), wp.data_ptr(), _lib.ptr(bias), _lib.ptr(r), out.data_ptr(), partial.data_ptr(),
                                              B, H, W_, Cin, Cout, wp.shape[0], out_mode, 0, splitk, bm, bn, X3, hp.stream()))
    finally:
        _lib.check(_lib.lib().k22_set_option(b"conv_algo", 0))
    o = out if out_mode == _lib.OUT_NCHW_F32 else out.permute(0, 3, 1, 2)
    ref = F.conv2d(x.double(), w.double(), None if bias is None else bias.double(), padding=1)
    if res is not None:
        ref = ref + res.double()
    return o, ref, st


@pytest.mark.parametrize("B,Cin,Cout,H,W,bm,splitk", [
    (2, 128, 128, 16, 16, 256, 1), (1, 64, 192, 9, 13, 128, 1), (2, 256, 128, 8, 8, 256, 3), (3, 192, 256, 6, 10, 128, 2),
    (2, 384, 384, 24, 24, 256, 1), (1, 128, 256, 96, 96, 256, 1), (2, 128, 128, 12, 12, 0, 0), (1, 128, 136, 48, 48, 128, 1),
])
@pytest.mark.parametrize("algo", [1, 7, 11, 12])
def test_conv3x3_x3(B, Cin, Cout, H, W, bm, splitk, algo):
    """generic implicit GEMM (1), lock-step halo kernel (7), specialised halo kernels (11 / 12): x3-chunk input and weights"""
    x, w = rnd(B, Cin, H, W, seed=1), rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
    bias, res = rnd(Cout, seed=3), rnd(B, Cout, H, W, seed=4)
    gen = algo == 1
    o, ref, _ = _conv_x3(x, w, bias, res, splitk, (128 if gen else bm), (64 if gen else 0), algo)
    close64(o, ref, f"x3 conv algo {algo} {B}x{Cin}->{Cout}@{H}x{W}")


@pytest.mark.parametrize("algo", [7, 11, 12])
def test_conv3x3_x3_groupnorm_partial_sums_and_nchw(algo):
    x, w = rnd(2, 128, 24, 24, seed=1), rnd(256, 128, 3, 3, seed=2, scale=(9 * 128) ** -0.5)
    bias, res = rnd(256, seed=3), rnd(2, 256, 24, 24, seed=4)
    o, ref, st = _conv_x3(x, w, bias, res, 1, 128, 0, algo, stats=True)
    close64(o, ref, "x3 conv with stats")
    od = o.double()
    assert (st[..., 0] - od.sum((2, 3))).abs().max().item() <= 1e-4 * (od.abs().sum((2, 3)).max().item() + 1)
    assert ((st[..., 1] - (od * od).sum((2, 3))).abs() / (od * od).sum((2, 3))).max().item() <= 1e-5
    w8 = rnd(8, 128, 3, 3, seed=5, scale=0.03)
    o, ref, _ = _conv_x3(x, w8, rnd(8, seed=6), None, 1, 0, 0, 0, out_mode=_lib.OUT_NCHW_F32)   # the UNet's output convolution (generic kernel)
    close64(o, ref, "x3 conv, 8 output channels, NCHW fp32")


@pytest.mark.parametrize("B,Cin,Cout,SK0,SK1,H,W,bm,splitk", [
    (2, 128, 128, 64, 0, 16, 16, 256, 1), (1, 128, 256, 128, 64, 24, 24, 128, 1), (2, 256, 128, 192, 128, 12, 12, 256, 2),
    (2, 128, 128, 320, 0, 8, 8, 128, 4),
])
@pytest.mark.parametrize("algo", [7, 11, 12])
def test_conv3x3_x3_with_fused_skip_connection(B, Cin, Cout, SK0, SK1, H, W, bm, splitk, algo):
    """out = conv3x3(h) + conv1x1(cat(x0, x1)): h in x3 chunks, the skip operands as RAW fp32 rows split at fragment-read time"""
    h, w3 = rnd(B, Cin, H, W, seed=1), rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
    x0 = rnd(B, SK0, H, W, seed=5)
    x1 = rnd(B, SK1, H, W, seed=6) if SK1 else None
    ws, b3, bs = rnd(Cout, SK0 + SK1, seed=7, scale=(SK0 + SK1) ** -0.5), rnd(Cout, seed=3), rnd(Cout, seed=8)
    hpad, w3p = x3_pack(hp.nhwc_padded(h, torch.float32)), to_x3(hp.pack_conv3(w3, torch.float32))
    x0n = x0.permute(0, 2, 3, 1).contiguous()
    x1n = None if x1 is None else x1.permute(0, 2, 3, 1).contiguous()
    wsp = to_x3(hp.pad_rows(ws))
    out = torch.empty(B, H, W, Cout, device="cuda")
    partial = torch.empty(max(1, splitk) * B * H * W * Cout + 64, device="cuda")
    _lib.check(_lib.lib().k22_set_option(b"conv_algo", algo))
    try:
        _lib.check(_lib.lib().k22_conv3x3_skip(hpad.data_ptr(), w3p.data_ptr(), b3.data_ptr(), x0n.data_ptr(), _lib.ptr(x1n), SK0, SK1,
                                               wsp.data_ptr(), bs.data_ptr(), out.data_ptr(), partial.data_ptr(), B, H, W, Cin, Cout,
                                               w3p.shape[0], splitk, bm, X3, hp.stream()))
    finally:
        _lib.check(_lib.lib().k22_set_option(b"conv_algo", 0))
    xin = x0.double() if x1 is None else torch.cat([x0.double(), x1.double()], 1)
    ref = F.conv2d(h.double(), w3.double(), b3.double(), padding=1) + F.conv2d(xin, ws.double()[:, :, None, None], bs.double())
    close64(out.permute(0, 3, 1, 2), ref, "x3 conv3x3 + fused 1x1 skip")


@pytest.mark.parametrize("C0,C1,H,W,act,mode,pad,film", [
    (128, 0, 16, 16, 1, 0, 1, False), (256, 128, 8, 8, 1, 0, 1, False), (384, 0, 12, 12, 1, 0, 1, True), (128, 0, 16, 16, 1, 1, 1, False),
    (128, 0, 8, 8, 1, 2, 1, False), (512, 0, 6, 6, 0, 0, 0, False), (1152, 768, 7, 9, 1, 0, 1, False),
])
def test_groupnorm_writes_x3_chunks(C0, C1, H, W, act, mode, pad, film):
    """k22_groupnorm with dtype K22_F16X3 = the fp32 GroupNorm kernels with the x3-chunk store the consuming convolution reads"""
    B, C = 2, C0 + C1
    x0 = rnd(B, C0, H, W, seed=1) * 1.7 + 0.3
    x1 = (rnd(B, C1, H, W, seed=2) * 0.6 - 0.2) if C1 else None
    gamma, beta = 1 + 0.1 * rnd(C, seed=3), 0.1 * rnd(C, seed=4)
    fl = 0.3 * rnd(B, 2 * C, seed=5) if film else None
    a0 = x0.permute(0, 2, 3, 1).contiguous()
    a1 = None if x1 is None else x1.permute(0, 2, 3, 1).contiguous()
    Ho, Wo = (H // 2, W // 2) if mode == 1 else ((H * 2, W * 2) if mode == 2 else (H, W))
    outs = []
    for dt in (_lib.K22_F32, X3):
        out = torch.full((B, Ho + 2 * pad, Wo + 2 * pad, C), float("nan"), device="cuda")
        scratch = torch.empty(_lib.lib().k22_groupnorm_scratch_bytes(B, C), dtype=torch.uint8, device="cuda")
        _lib.check(_lib.lib().k22_groupnorm(a0.data_ptr(), _lib.ptr(a1), C0, C1, B, H, W, gamma.data_ptr(), beta.data_ptr(), _lib.ptr(fl),
                                            0 if fl is None else fl.shape[1], 1e-5, act, mode, pad, scratch.data_ptr(), out.data_ptr(), dt, hp.stream()))
        outs.append(out)
    want = outs[0].double()
    got = x3_unpack(outs[1])
    err = (got - want).abs()
    if act == 0:
        assert (err <= torch.maximum(want.abs() * 2.0 ** -22, torch.full_like(err, 2.0 ** -24))).all()
        assert torch.equal(outs[1].view(torch.int32), to_x3(outs[0], 1.0).view(torch.int32))   # exactly the chunks of the fp32 kernel's values
    else:
        # the x3 store computes SiLU from the native exp2 / rcp (common.h: silu_fast, ~3 ulp of fp32), the fp32 kernel the IEEE form:
        # same value to a few fp32 ulps, then the same split
        assert (err <= torch.maximum(want.abs() * 2.0 ** -20, torch.full_like(err, 2.0 ** -22))).all()


@pytest.mark.parametrize("B,H,T,S", [(2, 2, 64, 87), (1, 3, 144, 87), (2, 4, 576, 87), (2, 1, 100, 5), (1, 12, 2304, 87)])
def test_attention_x3(B, H, T, S):
    C_ = 64 * H
    qkv, ctx = rnd(B * T, 3 * C_, seed=1) * 1.5, rnd(B * S, 2 * C_, seed=2) * 1.5
    Tkp = (S + T + 63) // 64 * 64
    kall = torch.full((B, H, Tkp, 64), float("nan"), device="cuda")
    vtall = torch.full((B, H, 64, Tkp), float("nan"), device="cuda")
    out = torch.empty(B * T, C_, device="cuda")
    _lib.check(_lib.lib().k22_attention(qkv.data_ptr(), ctx.data_ptr(), kall.data_ptr(), vtall.data_ptr(), out.data_ptr(), B, H, T, S, X3, hp.stream()))
    qf, cf = qkv.double().view(B, T, 3, H, 64), ctx.double().view(B, S, 2, H, 64)
    q = qf[:, :, 0].permute(0, 2, 1, 3)
    k = torch.cat([cf[:, :, 0], qf[:, :, 1]], 1).permute(0, 2, 1, 3)
    v = torch.cat([cf[:, :, 1], qf[:, :, 2]], 1).permute(0, 2, 1, 3)
    ref = (torch.softmax((q @ k.transpose(-1, -2)) * 0.125, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B * T, C_)
    close64(out, ref, f"x3 attention B{B} H{H} T{T}", tol=2e-5)


def test_attention_x3_online_softmax_rescale_branch():
    B, H, T, S = 1, 1, 200, 87
    qkv, ctx = rnd(B * T, 192, seed=1), rnd(B * S, 128, seed=2)
    qkv[180, 64:128] = qkv[3, 0:64] * 40.0
    Tkp = (S + T + 63) // 64 * 64
    kall, vtall = torch.empty(B, H, Tkp, 64, device="cuda"), torch.empty(B, H, 64, Tkp, device="cuda")
    out = torch.empty(B * T, 64, device="cuda")
    _lib.check(_lib.lib().k22_attention(qkv.data_ptr(), ctx.data_ptr(), kall.data_ptr(), vtall.data_ptr(), out.data_ptr(), B, H, T, S, X3, hp.stream()))
    q = qkv[:, :64].double(); k = torch.cat([ctx[:, :64], qkv[:, 64:128]]).double(); v = torch.cat([ctx[:, 64:], qkv[:, 128:]]).double()
    ref = torch.softmax(q @ k.T * 0.125, -1) @ v
    assert (out.double() - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("B,H,T,S,K,bm,gemm8", [(2, 2, 144, 87, 128, 64, 0), (2, 6, 64, 87, 384, 128, 0), (2, 2, 144, 87, 128, 256, 1), (1, 6, 100, 5, 192, 128, 1),
                                                 (2, 12, 2304, 87, 768, 256, 1)])
def test_qkv_projection_x3_writes_attention_operands(B, H, T, S, K, bm, gemm8):
    C_, Tkp = 64 * H, (S + T + 63) // 64 * 64
    x, W, bias = rnd(B * T, K, seed=1), rnd(3 * C_, K, seed=2, scale=K ** -0.5), rnd(3 * C_, seed=3)
    wt = to_x3(W)
    q = torch.empty(B * T, C_, device="cuda")
    kall = torch.full((B, H, Tkp, 64), 7.0, device="cuda")
    vtall = torch.full((B, H, 64, Tkp), 7.0, device="cuda")
    _lib.check(_lib.lib().k22_set_option(b"gemm_algo", 10 if gemm8 else 0))
    try:
        _lib.check(_lib.lib().k22_qkv_project(x.data_ptr(), wt.data_ptr(), bias.data_ptr(), q.data_ptr(), kall.data_ptr(), vtall.data_ptr(),
                                              B, H, T, S, K, bm, 0 if gemm8 else 64, X3, hp.stream()))
    finally:
        _lib.check(_lib.lib().k22_set_option(b"gemm_algo", 0))
    ref = (x.double() @ W.double().T + bias.double()).view(B, T, 3, H, 64)
    close64(q.view(B, T, H, 64), ref[:, :, 0], "q")
    close64(kall[:, :, S:S + T], ref[:, :, 1].permute(0, 2, 1, 3), "k")
    close64(vtall[:, :, :, S:S + T], ref[:, :, 2].permute(0, 2, 3, 1), "v^T")
    assert (kall[:, :, :S] == 7).all() and (kall[:, :, S + T:] == 7).all()
    assert (vtall[:, :, :, :S] == 7).all() and (vtall[:, :, :, S + T:] == 7).all()
