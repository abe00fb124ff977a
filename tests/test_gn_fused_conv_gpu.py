"""GroupNorm-apply (+FiLM, +SiLU, zero border) FUSED into the 3x3 convolution's halo fill (conv3_halo_spec_kernel producers, k22_conv3x3_gn;
nn.py:26-37 + unet.py:150-152, 174-180, 212-216): against F.group_norm -> SiLU -> F.conv2d in fp32 / fp64 on the same operands, and against
the two-call form (k22_groupnorm + k22_conv3x3) BIT FOR BIT - the producers apply gn_apply_kernel's own expression to the same raw values."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import helpers as hp
from kandinsky2_amd import _lib
from kandinsky2_amd.pack import to_x3

pytestmark = [pytest.mark.gpu, pytest.mark.slow]   # K22_FUSE_GN ships OFF (21 % slower); run by default since round 6, K22_RUN_SLOW=0 skips them
X3 = _lib.K22_F16X3
DT = [_lib.K22_BF16, _lib.K22_F16, _lib.K22_F32, X3]


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def tdt(dtype):
    return torch.float32 if dtype == X3 else hp.tdt(dtype)


def wpack(w, dtype):
    wp = hp.pack_conv3(w, torch.float32)
    return to_x3(wp) if dtype == X3 else wp.to(hp.tdt(dtype)).contiguous()


CASES = [
    # B, C0, C1, Cout, H, W, bm, splitk, film
    (2, 128, 0, 128, 16, 16, 256, 1, False), (1, 64, 64, 192, 9, 13, 128, 1, True), (2, 256, 0, 128, 8, 8, 256, 3, True),
    (3, 128, 128, 256, 6, 10, 128, 2, False), (2, 384, 0, 384, 24, 24, 256, 1, True), (1, 128, 0, 256, 96, 96, 256, 1, False),
    (2, 384, 384, 384, 12, 12, 128, 4, True), (1, 128, 0, 136, 48, 48, 128, 1, True), (2, 768, 384, 384, 48, 48, 256, 1, True),
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,C0,C1,Cout,H,W,bm,splitk,film", CASES)
@pytest.mark.parametrize("algo", [11, 12])
def test_conv3x3_with_fused_groupnorm(dtype, B, C0, C1, Cout, H, W, bm, splitk, film, algo):
    L = _lib.lib()
    T = tdt(dtype)
    C = C0 + C1
    x0 = (rnd(B, C0, H, W, seed=1) * 1.7 + 0.3)
    x1 = (rnd(B, C1, H, W, seed=2) * 0.6 - 0.2) if C1 else None
    gamma, beta = 1 + 0.1 * rnd(C, seed=3), 0.1 * rnd(C, seed=4)
    fl = 0.3 * rnd(B, 2 * C, seed=5) if film else None
    w, bias, res = rnd(Cout, C, 3, 3, seed=6, scale=(9 * C) ** -0.5), rnd(Cout, seed=7), rnd(B, Cout, H, W, seed=8)
    a0 = x0.permute(0, 2, 3, 1).contiguous().to(T)
    a1 = None if x1 is None else x1.permute(0, 2, 3, 1).contiguous().to(T)
    r = res.permute(0, 2, 3, 1).contiguous().to(T)
    wp = wpack(w, dtype)
    scratch = torch.empty(L.k22_groupnorm_scratch_bytes(B, C), dtype=torch.uint8, device="cuda")
    partial = torch.empty(max(1, splitk) * B * H * W * Cout + 64, device="cuda")
    out = torch.full((B, H, W, Cout), float("nan"), dtype=T, device="cuda")
    _lib.check(L.k22_conv3x3_gn(a0.data_ptr(), _lib.ptr(a1), C0, C1, gamma.data_ptr(), beta.data_ptr(), _lib.ptr(fl), 0 if fl is None else fl.shape[1],
                                1e-5, 1, scratch.data_ptr(), wp.data_ptr(), bias.data_ptr(), r.data_ptr(), out.data_ptr(), partial.data_ptr(),
                                B, H, W, Cout, wp.shape[0], splitk, bm, algo, dtype, hp.stream()))
    # ---- two-call form: stand-alone GroupNorm (zero-bordered output) + the same convolution kernel ------------------------------------------------
    pad = torch.full((B, H + 2, W + 2, C), float("nan"), dtype=T, device="cuda")
    _lib.check(L.k22_groupnorm(a0.data_ptr(), _lib.ptr(a1), C0, C1, B, H, W, gamma.data_ptr(), beta.data_ptr(), _lib.ptr(fl), 0 if fl is None else fl.shape[1],
                               1e-5, 1, 0, 1, scratch.data_ptr(), pad.data_ptr(), dtype, hp.stream()))
    out2 = torch.full((B, H, W, Cout), float("nan"), dtype=T, device="cuda")
    _lib.check(L.k22_set_option(b"conv_algo", algo))
    try:
        _lib.check(L.k22_conv3x3(pad.data_ptr(), wp.data_ptr(), bias.data_ptr(), r.data_ptr(), out2.data_ptr(), partial.data_ptr(),
                                 B, H, W, C, Cout, wp.shape[0], 0, 0, splitk, bm, 0, dtype, hp.stream()))
    finally:
        _lib.check(L.k22_set_option(b"conv_algo", 0))
    assert torch.isfinite(out.float()).all()
    assert torch.equal(out.view(torch.uint8), out2.view(torch.uint8)), f"fused != two-call: max|d| {(out.float() - out2.float()).abs().max().item():.3e}"
    # ---- reference: the same ops in fp64 on the stored operands ----------------------------------------------------------------------------------
    xin = a0.double().permute(0, 3, 1, 2)
    if a1 is not None:
        xin = torch.cat([xin, a1.double().permute(0, 3, 1, 2)], 1)
    y = F.group_norm(xin, 32, gamma.double(), beta.double(), eps=1e-5)
    if fl is not None:
        y = y * (1 + fl.double()[:, :C, None, None]) + fl.double()[:, C:2 * C, None, None]
    y = F.silu(y)
    wd = w.double() if dtype == X3 else w.to(T).double()
    ref = F.conv2d(y, wd, bias.double(), padding=1) + r.double().permute(0, 3, 1, 2)
    scale = ref.abs().max().item()
    err = (out.double().permute(0, 3, 1, 2) - ref).abs().max().item()
    # 16-bit types: the normalised operand is rounded to T once (2^-8 / 2^-11 relative per element, averaged over K) and the output once
    tol = {_lib.K22_BF16: 1.5e-2, _lib.K22_F16: 2e-3, _lib.K22_F32: 2e-5, X3: 2e-5}[dtype] * scale
    assert np.isfinite(err) and err <= tol, f"max|d| {err:.3e} tol {tol:.3e} scale {scale:.2f}"
