"""Shared helpers of the parity tests: thin typed wrappers over the C ABI (through ctypes, exactly as the
product calls it) and torch fp32 references of single floating-point ops."""
import torch
import torch.nn.functional as F

from kandinsky2_amd import _lib


def tdt(dtype_code):
    return {_lib.K22_BF16: torch.bfloat16, _lib.K22_F16: torch.float16, _lib.K22_F32: torch.float32}[dtype_code]


def pad_rows(w, mult=64):
    o = w.shape[0]
    op = (o + mult - 1) // mult * mult
    if op == o:
        return w.contiguous()
    return torch.cat([w, torch.zeros((op - o,) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)], 0).contiguous()


def stream():
    return torch.cuda.current_stream().cuda_stream


def gemm(A0, W, bias=None, residual=None, A1=None, dtype=_lib.K22_BF16, splitk=1, bm=0, bn=0, out_f32=False, act=0):
    """A0 [M,K0], A1 [M,K1] or None, W [N,K0+K1] (all float32 cuda) -> out [M,N] (float32 copy) and the
    T-rounded operands actually used."""
    T = tdt(dtype)
    M, K0 = A0.shape
    K1 = 0 if A1 is None else A1.shape[1]
    N = W.shape[0]
    a0 = A0.to(T).contiguous()
    a1 = None if A1 is None else A1.to(T).contiguous()
    wp = pad_rows(W.to(T))
    res = None if residual is None else residual.to(T).contiguous()
    out = torch.empty(M, N, dtype=torch.float32 if out_f32 else T, device=A0.device)
    partial = torch.empty(max(1, abs(splitk) if splitk else 16) * M * N + 64, dtype=torch.float32, device=A0.device)
    _lib.check(_lib.lib().k22_gemm(
        a0.data_ptr(), _lib.ptr(a1), wp.data_ptr(), _lib.ptr(bias), _lib.ptr(res), out.data_ptr(), partial.data_ptr(),
        M, N, wp.shape[0], K0, K1, K0, K1, N, N, 1 if out_f32 else 0, act, splitk, bm, bn, dtype, stream()))
    a_full = a0.float() if a1 is None else torch.cat([a0.float(), a1.float()], 1)
    return out.float(), a_full, W.to(T).float(), (None if res is None else res.float())


def nhwc_padded(x, T):
    """[B,C,H,W] float -> zero-bordered NHWC [B,H+2,W+2,C] of dtype T."""
    return F.pad(x, (1, 1, 1, 1)).permute(0, 2, 3, 1).contiguous().to(T)


def pack_conv3(w, T):
    return pad_rows(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(T))


def conv3x3(x, w, bias=None, residual=None, dtype=_lib.K22_BF16, splitk=1, bm=0, bn=0, out_mode=0, algo=0, stats=False):
    """x [B,Cin,H,W], w [Cout,Cin,3,3], residual [B,Cout,H,W] (float32 cuda) -> NCHW float32.
    algo: 0 auto, 1 generic implicit GEMM, 2 LDS-resident halo kernel.  stats=True also returns the per-image,
    per-channel (sum, sumsq) reduced from the kernel's GroupNorm partial sums."""
    _lib.check(_lib.lib().k22_set_option(b"conv_algo", algo))
    try:
        return _conv3x3(x, w, bias, residual, dtype, splitk, bm, bn, out_mode, stats)
    finally:
        _lib.check(_lib.lib().k22_set_option(b"conv_algo", 0))


def _conv3x3(x, w, bias, residual, dtype, splitk, bm, bn, out_mode, stats):
    T = tdt(dtype)
    B, Cin, H, W_ = x.shape
    Cout = w.shape[0]
    xp = nhwc_padded(x, T)
    wp = pack_conv3(w, T)
    res = None if residual is None else residual.permute(0, 2, 3, 1).contiguous().to(T)
    if out_mode == _lib.OUT_NCHW_F32:
        out = torch.empty(B, Cout, H, W_, dtype=torch.float32, device=x.device)
    else:
        out = torch.empty(B, H, W_, Cout, dtype=T, device=x.device)
    partial = torch.empty(max(1, splitk if splitk else 16) * B * H * W_ * Cout + 64, dtype=torch.float32, device=x.device)
    st = None
    if stats:
        import ctypes as C
        cap = B * (H * (W_ + 2) // 16 + 2)
        sbuf = torch.full((cap, Cout, 2), float("nan"), dtype=torch.float32, device=x.device)
        rpi = C.c_int(0)
        _lib.check(_lib.lib().k22_conv3x3_gnstats(
            xp.data_ptr(), wp.data_ptr(), _lib.ptr(bias), _lib.ptr(res), out.data_ptr(), partial.data_ptr(),
            B, H, W_, Cin, Cout, wp.shape[0], splitk, bm, bn, sbuf.data_ptr(), cap, C.byref(rpi), dtype, stream()))
        st = sbuf[: B * rpi.value].view(B, rpi.value, Cout, 2).double().sum(1)  # [B, Cout, 2]
    else:
        _lib.check(_lib.lib().k22_conv3x3(
            xp.data_ptr(), wp.data_ptr(), _lib.ptr(bias), _lib.ptr(res), out.data_ptr(), partial.data_ptr(),
            B, H, W_, Cin, Cout, wp.shape[0], out_mode, 0, splitk, bm, bn, dtype, stream()))
    o = out.float() if out_mode == _lib.OUT_NCHW_F32 else out.float().permute(0, 3, 1, 2).contiguous()
    ref = F.conv2d(x.to(T).float(), w.to(T).float(), bias, padding=1)
    if res is not None:
        ref = ref + res.float().permute(0, 3, 1, 2)
    if stats:
        return o, ref, st
    return o, ref


def groupnorm(x0, gamma, beta, x1=None, film=None, act=0, mode=0, pad=0, dtype=_lib.K22_BF16):
    """x0 [B,C0,H,W] (+ x1 [B,C1,H,W]) float32 cuda -> (engine out NCHW float, torch reference NCHW float)."""
    T = tdt(dtype)
    B, C0, H, W_ = x0.shape
    C1 = 0 if x1 is None else x1.shape[1]
    C = C0 + C1
    a0 = x0.permute(0, 2, 3, 1).contiguous().to(T)
    a1 = None if x1 is None else x1.permute(0, 2, 3, 1).contiguous().to(T)
    Ho, Wo = (H // 2, W_ // 2) if mode == 1 else ((H * 2, W_ * 2) if mode == 2 else (H, W_))
    out = torch.full((B, Ho + 2 * pad, Wo + 2 * pad, C), float("nan"), dtype=T, device=x0.device)
    scratch = torch.empty(_lib.lib().k22_groupnorm_scratch_bytes(B, C), dtype=torch.uint8, device=x0.device)
    _lib.check(_lib.lib().k22_groupnorm(
        a0.data_ptr(), _lib.ptr(a1), C0, C1, B, H, W_, gamma.data_ptr(), beta.data_ptr(), _lib.ptr(film),
        0 if film is None else film.shape[1], 1e-5, act, mode, pad, scratch.data_ptr(), out.data_ptr(), dtype, stream()))
    xin = a0.float().permute(0, 3, 1, 2)
    if a1 is not None:
        xin = torch.cat([xin, a1.float().permute(0, 3, 1, 2)], 1)
    y = F.group_norm(xin, 32, gamma, beta, eps=1e-5)
    if film is not None:
        y = y * (1 + film[:, :C, None, None]) + film[:, C:2 * C, None, None]
    if act == 1:
        y = F.silu(y)
    if mode == 1:
        y = F.avg_pool2d(y, 2)
    elif mode == 2:
        y = F.interpolate(y, scale_factor=2, mode="nearest")
    if pad:
        y = F.pad(y, (1, 1, 1, 1))
    return out.float().permute(0, 3, 1, 2).contiguous(), y


def attention(qkv, ctxkv, B, H, T_, S, dtype=_lib.K22_BF16):
    """qkv [B*T,3C] columns [q|k|v] x [H][64]; ctxkv [B*S,2C] columns [k|v] (float32 cuda)."""
    T = tdt(dtype)
    C = H * 64
    q_ = qkv.to(T).contiguous()
    c_ = ctxkv.to(T).contiguous()
    Tkp = (S + T_ + 63) // 64 * 64
    kall = torch.full((B, H, Tkp, 64), float("nan"), dtype=T, device=qkv.device)
    vtall = torch.full((B, H, 64, Tkp), float("nan"), dtype=T, device=qkv.device)
    out = torch.empty(B * T_, C, dtype=T, device=qkv.device)
    _lib.check(_lib.lib().k22_attention(q_.data_ptr(), c_.data_ptr(), kall.data_ptr(), vtall.data_ptr(), out.data_ptr(),
                                        B, H, T_, S, dtype, stream()))
    qf = q_.float().view(B, T_, 3, H, 64)
    cf = c_.float().view(B, S, 2, H, 64)
    q = qf[:, :, 0].permute(0, 2, 1, 3)                                   # [B,H,T,64]
    k = torch.cat([cf[:, :, 0], qf[:, :, 1]], 1).permute(0, 2, 1, 3)      # [B,H,S+T,64]
    v = torch.cat([cf[:, :, 1], qf[:, :, 2]], 1).permute(0, 2, 1, 3)
    w = torch.softmax((q @ k.transpose(-1, -2)) * 0.125, dim=-1)
    ref = (w @ v).permute(0, 2, 1, 3).reshape(B * T_, C)
    return out.float(), ref
