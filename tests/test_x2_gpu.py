"""GPU parity of the ASYMMETRIC split arithmetic (K22_F16X2, include/k22.h; round 5): every MFMA kernel that has an x2 instantiation,
called through the C ABI, against fp64 references that restate exactly what the arithmetic promises:

    conv / GEMM:  out = W . fp16(a)        weights at (hi, lo) precision (~22 bits), the ACTIVATION operand rounded to fp16 (rne) -
                                           two v_mfma_f32_32x32x16_f16 per product (w_hi.a_hi + w_lo.a_hi), fp32 accumulation;
                  the fused 1x1 skip connection inside an x2 convolution keeps the full split (its input is the un-normalised residual
                  stream: half of the error budget if rounded, tests/golden/drift_ablation_x2.json);
    attention:    q, k, v and P at fp16 (one MFMA per product), fp32 online softmax, output written as x3 chunks.

Because the reference applies the same operand rounding, the conv / GEMM tolerance is the split-precision one (1e-5 of the output scale):
a lost lo half of the weights (~3e-4) or an activation taken at bf16 / truncated instead of rounded-to-nearest fails it.  The distance to
the EXACT product (what the rounding costs) is printed and bounded at 2^-10 of the scale.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import helpers as hp
from kandinsky2_amd import _lib
from kandinsky2_amd.pack import to_x3

pytestmark = pytest.mark.gpu
X2 = _lib.K22_F16X2
TOL = 1e-5


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def h16(t):
    """the value an x2 kernel sees of an activation operand: rne to fp16, as float64"""
    return t.to(torch.float16).double()


def close64(out, ref64, what="", tol=TOL):
    scale = ref64.abs().max().item() + 1e-9
    err = (out.double() - ref64).abs().max().item()
    assert np.isfinite(err) and err <= tol * scale, f"{what}: max|d|={err:.3e} tol={tol * scale:.3e} scale={scale:.3f}"


def x3_pack(t, scale=1.0):
    t = t.contiguous()
    out = torch.empty_like(t)
    _lib.check(_lib.lib().k22_x3_pack(t.data_ptr(), out.data_ptr(), t.numel(), scale, hp.stream()))
    return out


@pytest.mark.parametrize("M,N,K,bm,bn,splitk", [(256, 256, 128, 128, 128, 1), (300, 192, 192, 128, 64, 1), (288, 320, 1152, 128, 64, 4),
                                                 (333, 200, 320, 0, 0, 0)])
@pytest.mark.parametrize("gemm8", [0, 1])
def test_gemm_x2(M, N, K, bm, bn, splitk, gemm8):
    """generic implicit-GEMM kernel and gemm8, RAW fp32 A operand (rounded to fp16 while the fragments are read), x3-chunk weights"""
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
                                       0 if gemm8 else bn, X2, hp.stream()))
    finally:
        _lib.check(_lib.lib().k22_set_option(b"gemm_algo", 0))
    close64(out, h16(A) @ W.double().T + bias.double() + res.double(), f"x2 gemm {M}x{N}x{K}")
    exact = A.double() @ W.double().T + bias.double() + res.double()
    assert (out.double() - exact).abs().max().item() <= 2.0 ** -10 * exact.abs().max().item()


def test_gemm_x2_keeps_the_low_halves_of_the_weights():
    """A = I (exact in fp16) with an asymmetric W spanning 1e-4 .. 8: the output is W^T itself to ~22 bits - a dropped w_lo shows as 2^-11"""
    K = 128
    A = torch.eye(K).cuda()
    W = ((torch.arange(64 * K, dtype=torch.float32).reshape(64, K) % 251 - 125) / 16).cuda() * torch.logspace(-4, 0, K).cuda()
    wp = to_x3(W)
    out = torch.empty(K, 64, device="cuda")
    _lib.check(_lib.lib().k22_gemm(A.data_ptr(), None, wp.data_ptr(), None, None, out.data_ptr(), None, K, 64, 64, K, 0, K, 0, 64, 64, 0, 0, 1,
                                   0, 0, X2, hp.stream()))
    err = (out.double() - W.double().T).abs()
    assert (err <= W.double().T.abs() * 2.0 ** -21 + 1e-9).all()


def _conv_x2(x, w, bias, res, splitk, bm, bn, algo, out_mode=0):
    B, Cin, H, W_ = x.shape
    Cout = w.shape[0]
    xp = x3_pack(hp.nhwc_padded(x, torch.float32))      # the operand format the engine's GroupNorm store writes (hi AND lo: one plan mixes x2 / x3 ops)
    wp = to_x3(hp.pack_conv3(w, torch.float32))
    r = None if res is None else res.permute(0, 2, 3, 1).contiguous()
    out = torch.empty((B, Cout, H, W_) if out_mode == _lib.OUT_NCHW_F32 else (B, H, W_, Cout), device="cuda")
    partial = torch.empty(max(1, splitk if splitk else 16) * B * H * W_ * Cout + 64, device="cuda")
    _lib.check(_lib.lib().k22_set_option(b"conv_algo", algo))
    try:
        _lib.check(_lib.lib().k22_conv3x3(xp.data_ptr(), wp.data_ptr(), _lib.ptr(bias), _lib.ptr(r), out.data_ptr(), partial.data_ptr(),
                                          B, H, W_, Cin, Cout, wp.shape[0], out_mode, 0, splitk, bm, bn, X2, hp.stream()))
    finally:
        _lib.check(_lib.lib().k22_set_option(b"conv_algo", 0))
    o = out if out_mode == _lib.OUT_NCHW_F32 else out.permute(0, 3, 1, 2)
    ref = F.conv2d(h16(x), w.double(), None if bias is None else bias.double(), padding=1)
    if res is not None:
        ref = ref + res.double()
    return o, ref


@pytest.mark.parametrize("B,Cin,Cout,H,W,bm,splitk", [
    (2, 128, 128, 16, 16, 256, 1), (1, 64, 192, 9, 13, 128, 1), (2, 256, 128, 8, 8, 256, 3), (2, 384, 384, 24, 24, 256, 1),
    (1, 128, 256, 96, 96, 256, 1), (1, 128, 136, 48, 48, 128, 1),
])
@pytest.mark.parametrize("algo", [1, 7, 11, 12])
def test_conv3x3_x2(B, Cin, Cout, H, W, bm, splitk, algo):
    """generic implicit GEMM (1), lock-step halo kernel (7), specialised halo kernels (11; 12 = the two-set fragment pipeline, which in
    this arithmetic fits the register file at BM = 256 too: activation fragments are 4 registers)"""
    x, w = rnd(B, Cin, H, W, seed=1), rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
    bias, res = rnd(Cout, seed=3), rnd(B, Cout, H, W, seed=4)
    gen = algo == 1
    o, ref = _conv_x2(x, w, bias, res, splitk, (128 if gen else bm), (64 if gen else 0), algo)
    close64(o, ref, f"x2 conv algo {algo} {B}x{Cin}->{Cout}@{H}x{W}")
    exact = F.conv2d(x.double(), w.double(), bias.double(), padding=1) + res.double()
    d = (o.double() - exact).abs().max().item()
    print(f"x2 conv algo {algo} {Cin}->{Cout}@{H}x{W}: distance to the exact product {d / exact.abs().max().item():.2e} of scale")
    assert d <= 2.0 ** -10 * exact.abs().max().item()


@pytest.mark.parametrize("B,Cin,Cout,SK0,SK1,H,W,bm,splitk", [(2, 128, 128, 64, 0, 16, 16, 256, 1), (1, 128, 256, 128, 64, 24, 24, 128, 1),
                                                              (2, 256, 128, 192, 128, 12, 12, 256, 2)])
@pytest.mark.parametrize("algo", [7, 12])
def test_conv3x3_x2_with_fused_skip_connection_at_full_split(B, Cin, Cout, SK0, SK1, H, W, bm, splitk, algo):
    """out = conv3x3(fp16(h)) + conv1x1(cat(x0, x1)): the skip K loop of an x2 convolution keeps all three MFMAs - the skip operands are
    scaled x 60 here (the residual stream is not normalised), so taking them at fp16 would miss the tolerance by two orders"""
    h, w3 = rnd(B, Cin, H, W, seed=1), rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
    x0 = rnd(B, SK0, H, W, seed=5) * 60.0
    x1 = rnd(B, SK1, H, W, seed=6) * 60.0 if SK1 else None
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
                                               w3p.shape[0], splitk, bm, X2, hp.stream()))
    finally:
        _lib.check(_lib.lib().k22_set_option(b"conv_algo", 0))
    xin = x0.double() if x1 is None else torch.cat([x0.double(), x1.double()], 1)
    conv = F.conv2d(h16(h), w3.double(), b3.double(), padding=1)
    skip = F.conv2d(xin, ws.double()[:, :, None, None], bs.double())
    # tolerance relative to the CONVOLUTION's scale: the x 60 skip term must not loosen it
    err = (out.permute(0, 3, 1, 2).double() - (conv + skip)).abs().max().item()
    assert err <= 3e-5 * skip.abs().max().item() and err <= 2e-3 * conv.abs().max().item(), f"max|d| {err:.3e} conv scale {conv.abs().max().item():.2f}"


@pytest.mark.parametrize("B,H,T,S", [(2, 2, 64, 87), (2, 4, 576, 87), (2, 1, 100, 5), (1, 12, 2304, 87)])
def test_attention_x2(B, H, T, S):
    """fp32 q / K_all / V^T_all in memory, fp16 tiles and one MFMA per product, fp32 softmax: fp16-class distance from the exact result"""
    C_ = 64 * H
    qkv, ctx = rnd(B * T, 3 * C_, seed=1) * 1.5, rnd(B * S, 2 * C_, seed=2) * 1.5
    Tkp = (S + T + 63) // 64 * 64
    kall = torch.full((B, H, Tkp, 64), float("nan"), device="cuda")
    vtall = torch.full((B, H, 64, Tkp), float("nan"), device="cuda")
    out = torch.empty(B * T, C_, device="cuda")
    _lib.check(_lib.lib().k22_attention(qkv.data_ptr(), ctx.data_ptr(), kall.data_ptr(), vtall.data_ptr(), out.data_ptr(), B, H, T, S, X2, hp.stream()))
    qf, cf = h16(qkv).view(B, T, 3, H, 64), h16(ctx).view(B, S, 2, H, 64)
    q = qf[:, :, 0].permute(0, 2, 1, 3)
    k = torch.cat([cf[:, :, 0], qf[:, :, 1]], 1).permute(0, 2, 1, 3)
    v = torch.cat([cf[:, :, 1], qf[:, :, 2]], 1).permute(0, 2, 1, 3)
    ref = (torch.softmax((q @ k.transpose(-1, -2)) * 0.125, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B * T, C_)
    close64(out, ref, f"x2 attention B{B} H{H} T{T}", tol=1.5e-3)   # P is rounded to fp16 inside (2^-11 per probability)


def test_attention_x2_online_softmax_rescale_branch():
    B, H, T, S = 1, 1, 200, 87
    qkv, ctx = rnd(B * T, 192, seed=1), rnd(B * S, 128, seed=2)
    qkv[180, 64:128] = qkv[3, 0:64] * 40.0
    Tkp = (S + T + 63) // 64 * 64
    kall, vtall = torch.empty(B, H, Tkp, 64, device="cuda"), torch.empty(B, H, 64, Tkp, device="cuda")
    out = torch.empty(B * T, 64, device="cuda")
    _lib.check(_lib.lib().k22_attention(qkv.data_ptr(), ctx.data_ptr(), kall.data_ptr(), vtall.data_ptr(), out.data_ptr(), B, H, T, S, X2, hp.stream()))
    q = h16(qkv[:, :64]); k = h16(torch.cat([ctx[:, :64], qkv[:, 64:128]])); v = h16(torch.cat([ctx[:, 64:], qkv[:, 128:]]))
    ref = torch.softmax(q @ k.T * 0.125, -1) @ v
    assert (out.double() - ref).abs().max().item() < 5e-3


@pytest.mark.parametrize("B,H,T,S,K,bm,gemm8", [(2, 2, 144, 87, 128, 64, 0), (2, 2, 144, 87, 128, 256, 1), (2, 12, 2304, 87, 768, 256, 1)])
def test_qkv_projection_x2_writes_attention_operands(B, H, T, S, K, bm, gemm8):
    C_, Tkp = 64 * H, (S + T + 63) // 64 * 64
    x, W, bias = rnd(B * T, K, seed=1), rnd(3 * C_, K, seed=2, scale=K ** -0.5), rnd(3 * C_, seed=3)
    wt = to_x3(W)
    q = torch.empty(B * T, C_, device="cuda")
    kall = torch.full((B, H, Tkp, 64), 7.0, device="cuda")
    vtall = torch.full((B, H, 64, Tkp), 7.0, device="cuda")
    _lib.check(_lib.lib().k22_set_option(b"gemm_algo", 10 if gemm8 else 0))
    try:
        _lib.check(_lib.lib().k22_qkv_project(x.data_ptr(), wt.data_ptr(), bias.data_ptr(), q.data_ptr(), kall.data_ptr(), vtall.data_ptr(),
                                              B, H, T, S, K, bm, 0 if gemm8 else 64, X2, hp.stream()))
    finally:
        _lib.check(_lib.lib().k22_set_option(b"gemm_algo", 0))
    ref = (h16(x) @ W.double().T + bias.double()).view(B, T, 3, H, 64)
    close64(q.view(B, T, H, 64), ref[:, :, 0], "q")
    close64(kall[:, :, S:S + T], ref[:, :, 1].permute(0, 2, 1, 3), "k")
    close64(vtall[:, :, :, S:S + T], ref[:, :, 2].permute(0, 2, 3, 1), "v^T")
    assert (kall[:, :, :S] == 7).all() and (kall[:, :, S + T:] == 7).all()
