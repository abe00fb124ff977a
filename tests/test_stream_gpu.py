"""GPU parity of the weight-streaming small-M kernel (kandinsky-2_amd/csrc/stream_gemm.hip), through the C ABI, against plain
PyTorch fp32 references of the same op on the same (dtype-rounded) operands - the 3x3 convolutions, qkv / proj_out GEMMs and
prior Linears whose M is a few hundred rows (kandinsky2/model/unet.py:152,180,191,251,258 at the 12x12 / 24x24 levels,
kandinsky2/model/prior.py:93-120).

Each case asserts that the streaming kernel really ran (k22_debug_counter) - a silent fallback to another kernel would pass
the numbers.  Tolerance: operands pre-rounded, fp32 accumulation, ONE rounding of the output (1.2e-2 of the output scale for
bf16, as in test_kernels_gpu.py).
"""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

import helpers as hp
from kandinsky2_amd import _lib
from test_kernels_gpu import close, rnd

pytestmark = pytest.mark.gpu
DT = [_lib.K22_BF16, _lib.K22_F16]
_SCRATCH = {}


@pytest.fixture(autouse=True, params=[pytest.param("rowmajor", marks=pytest.mark.slow), "fragmajor"])
def weight_layout(request):
    """Every case can run twice: through the fragment-major copy (k22_stream_repack; the C ABI's test entries repack into this scratch
    before each launch - k22_debug_set_stream_scratch), which is what the engines run, and with the weights read row-major as packed for
    the other kernels - a form that never won a measurement and that no engine selects: its cases are `slow` (skipped by K22_RUN_SLOW=0)."""
    if request.param == "fragmajor":
        if "buf" not in _SCRATCH:
            _SCRATCH["buf"] = torch.empty(96 << 20, dtype=torch.uint8, device="cuda")
        _lib.check(_lib.lib().k22_debug_set_stream_scratch(_SCRATCH["buf"].data_ptr(), _SCRATCH["buf"].numel()))
    yield request.param
    _lib.check(_lib.lib().k22_debug_set_stream_scratch(None, 0))


def test_repack_is_the_documented_permutation():
    """out[(((nb * items + it) * 4 + w) * 64 + lane) * 8 + e] = W[nb * 32 + lane % 32][tap * Kc + slab * 64 + 16 w + 8 (lane / 32) + e]"""
    Npad, taps, Kc = 128, 9, 192
    W = torch.arange(Npad * taps * Kc, dtype=torch.int32).remainder(65521).to(torch.int16).reshape(Npad, taps * Kc).cuda()
    out = torch.empty_like(W)
    assert _lib.lib().k22_stream_frag_bytes(Npad, taps, Kc, _lib.K22_BF16) == W.numel() * 2
    _lib.check(_lib.lib().k22_stream_repack(W.data_ptr(), out.data_ptr(), Npad, taps, Kc, _lib.K22_BF16, torch.cuda.current_stream().cuda_stream))
    Wc = W.cpu().reshape(Npad // 32, 32, taps, Kc // 64, 4, 2, 8)            # nb, r, tap, slab, w, half, e
    want = Wc.permute(0, 3, 2, 4, 5, 1, 6).reshape(-1)                        # nb, slab, tap, w, (half, r) = lane, e
    assert torch.equal(out.cpu().reshape(-1), want)


class _Stream:
    """conv_algo / gemm_algo = 20 for the duration of a call, and the launch counter must move."""

    def __init__(self, option):
        self.option = option

    def __enter__(self):
        self.before = _lib.lib().k22_debug_counter(b"stream_launches")
        _lib.check(_lib.lib().k22_set_option(self.option, 20))
        return self

    def __exit__(self, *exc):
        _lib.check(_lib.lib().k22_set_option(self.option, 0))
        if exc[0] is None:
            assert _lib.lib().k22_debug_counter(b"stream_launches") > self.before, "the streaming kernel did not run"


CONV_CASES = [
    # B, Cin, Cout, H, W, bm, splitk
    (2, 128, 128, 12, 12, 160, 1), (2, 256, 192, 12, 12, 160, 2), (1, 128, 136, 24, 24, 160, 1), (2, 192, 128, 24, 24, 288, 3),
    (2, 128, 64, 8, 8, 160, 1), (1, 64, 128, 16, 16, 288, 1), (3, 128, 128, 6, 10, 160, 2), (2, 128, 128, 8, 8, 160, 4),
    (2, 1536, 1536, 12, 12, 160, 5), (2, 1152, 1152, 24, 24, 288, 3),
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,Cin,Cout,H,W,bm,splitk", CONV_CASES)
def test_stream_conv3x3(dtype, B, Cin, Cout, H, W, bm, splitk):
    """row bands + halo plane, image boundaries, ragged last m-block, Cout not a multiple of 64, split-K with idle workgroups."""
    x, w = rnd(B, Cin, H, W, seed=1), rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
    bias, res = rnd(Cout, seed=3), rnd(B, Cout, H, W, seed=4)
    with _Stream(b"conv_algo"):
        out, ref = hp._conv3x3(x, w, bias, res, dtype, splitk, bm, 0, 0, False)
    close(out, ref, dtype, f"stream conv {B}x{Cin}->{Cout}@{H}x{W}")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,Cin,Cout,H,W,bm,splitk", [(2, 256, 128, 12, 12, 160, 2), (2, 128, 256, 24, 24, 288, 1), (2, 128, 128, 16, 16, 160, 1)])
def test_stream_conv3x3_groupnorm_partial_sums(dtype, B, Cin, Cout, H, W, bm, splitk):
    x, w = rnd(B, Cin, H, W, seed=1), rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
    bias, res = rnd(Cout, seed=3), rnd(B, Cout, H, W, seed=4)
    with _Stream(b"conv_algo"):
        out, ref, st = hp._conv3x3(x, w, bias, res, dtype, splitk, bm, 0, 0, True)
    close(out, ref, dtype, "stream conv with stats")
    o = out.double()
    s_ref, q_ref = o.sum((2, 3)), (o * o).sum((2, 3))
    n = H * W
    assert (st[..., 0] - s_ref).abs().max().item() <= 1e-4 * n ** 0.5 * (q_ref.max().item() / n) ** 0.5 + 1e-3
    assert ((st[..., 1] - q_ref).abs() / q_ref).max().item() <= 1e-5


@pytest.mark.parametrize("dtype", DT)
def test_stream_conv3x3_nchw_f32_output(dtype):
    x, w, bias = rnd(2, 128, 12, 12, seed=1), rnd(8, 128, 3, 3, seed=2, scale=0.03), rnd(8, seed=3)
    with _Stream(b"conv_algo"):
        out, ref = hp._conv3x3(x, w, bias, None, dtype, 2, 160, 0, _lib.OUT_NCHW_F32, False)
    assert (out - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,Cin,Cout,SK0,SK1,H,W,bm,splitk", [
    (2, 256, 128, 192, 128, 12, 12, 160, 2), (2, 128, 128, 320, 0, 8, 8, 160, 4), (1, 128, 256, 128, 64, 24, 24, 288, 1),
    (2, 1536, 1536, 1536, 1536, 12, 12, 160, 5),
])
def test_stream_conv3x3_with_fused_skip_connection(dtype, B, Cin, Cout, SK0, SK1, H, W, bm, splitk):
    """out = conv3x3(h) + conv1x1(cat(x0, x1)) (unet.py:180,191): the 1x1 skip as a second K phase over the unpadded rows."""
    T = hp.tdt(dtype)
    h, w3 = rnd(B, Cin, H, W, seed=1), rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
    x0 = rnd(B, SK0, H, W, seed=5)
    x1 = rnd(B, SK1, H, W, seed=6) if SK1 else None
    ws, b3, bs = rnd(Cout, SK0 + SK1, seed=7, scale=(SK0 + SK1) ** -0.5), rnd(Cout, seed=3), rnd(Cout, seed=8)
    hpad, w3p = hp.nhwc_padded(h, T), hp.pack_conv3(w3, T)
    x0n = x0.permute(0, 2, 3, 1).contiguous().to(T)
    x1n = None if x1 is None else x1.permute(0, 2, 3, 1).contiguous().to(T)
    wsp = hp.pad_rows(ws.to(T))
    out = torch.empty(B, H, W, Cout, dtype=T, device="cuda")
    partial = torch.empty(max(1, splitk) * B * H * W * Cout + 64, device="cuda")
    with _Stream(b"conv_algo"):
        _lib.check(_lib.lib().k22_conv3x3_skip(hpad.data_ptr(), w3p.data_ptr(), b3.data_ptr(), x0n.data_ptr(), _lib.ptr(x1n), SK0, SK1,
                                               wsp.data_ptr(), bs.data_ptr(), out.data_ptr(), partial.data_ptr(), B, H, W, Cin, Cout,
                                               w3p.shape[0], splitk, bm, dtype, hp.stream()))
    xin = x0.to(T).float() if x1 is None else torch.cat([x0.to(T).float(), x1.to(T).float()], 1)
    ref = F.conv2d(h.to(T).float(), w3.to(T).float(), b3, padding=1) + F.conv2d(xin, ws.to(T).float()[:, :, None, None], bs)
    close(out.float().permute(0, 3, 1, 2), ref, dtype, "stream conv3x3 + fused 1x1 skip")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K,bm,splitk", [
    (288, 1536, 1536, 160, 3), (288, 320, 1152, 160, 4), (162, 2048, 2048, 160, 1), (77, 768, 1024, 160, 2), (333, 200, 320, 288, 1),
    (2, 1536, 384, 160, 2), (288, 1536, 1536, 288, 2), (160, 64, 64, 160, 1), (161, 128, 192, 160, 3), (1000, 136, 384, 288, 1),
])
def test_stream_gemm(dtype, M, N, K, bm, splitk):
    """odd slab counts (zero-slab tail of the two-slab stages), ragged m-tiles, N below / not a multiple of the n-tile."""
    A, W = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    with _Stream(b"gemm_algo"):
        out, a, w, r = hp.gemm(A, W, bias, res, dtype=dtype, splitk=splitk, bm=bm, bn=0)
    close(out, a @ w.T + bias + r, dtype, f"stream gemm {M}x{N}x{K}")


@pytest.mark.parametrize("dtype", DT)
def test_stream_gemm_virtual_concat_and_f32_out(dtype):
    A0, A1, W = rnd(200, 128, seed=1), rnd(200, 192, seed=2), rnd(256, 320, seed=3, scale=0.05)
    with _Stream(b"gemm_algo"):
        out, a, w, _ = hp.gemm(A0, W, A1=A1, dtype=dtype, out_f32=True, splitk=2, bm=160)
    scale = (a @ w.T).abs().max().item()
    assert (out - a @ w.T).abs().max().item() <= 2e-4 * scale  # fp32 store: no output rounding


def test_stream_gemm_asymmetric_layout():
    """A = I with an asymmetric integer-valued W (exact in bf16): any transposition of the MFMA output map, any wrong k-quarter
    or plane-row offset shows up as a wrong element, bit for bit."""
    K = 128
    A = torch.eye(K).cuda()
    W = ((torch.arange(64 * K, dtype=torch.float32).reshape(64, K) * 7) % 251 - 125).cuda()
    with _Stream(b"gemm_algo"):
        out, a, w, _ = hp.gemm(A, W, dtype=_lib.K22_BF16, out_f32=True, splitk=1, bm=160)
    assert torch.equal(out, w.T.contiguous())


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,H,W_,N,K,bm,splitk", [(2, 12, 12, 384, 256, 160, 1), (2, 12, 12, 1536, 1536, 160, 3), (1, 20, 12, 136, 192, 288, 1)])
def test_stream_gemm_groupnorm_partial_sums(dtype, B, H, W_, N, K, bm, splitk):
    """proj_out (unet.py:258) on the streaming kernel: the row-tiled finish delivers the next GroupNorm's partial sums."""
    T_ = hp.tdt(dtype)
    M = B * H * W_
    A, W, bias, res = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3), rnd(M, N, seed=4)
    a, w, r = A.to(T_).contiguous(), hp.pad_rows(W.to(T_)), res.to(T_).contiguous()
    out = torch.empty(M, N, dtype=T_, device="cuda")
    partial = torch.empty(max(1, splitk) * M * N + 64, dtype=torch.float32, device="cuda")
    cap = B * (H * W_ // 16 + 2)
    sbuf = torch.full((cap, N, 2), float("nan"), dtype=torch.float32, device="cuda")
    rpi = C.c_int(0)
    before = _lib.lib().k22_debug_counter(b"stream_launches")
    _lib.check(_lib.lib().k22_gemm_gnstats(a.data_ptr(), w.data_ptr(), bias.data_ptr(), r.data_ptr(), out.data_ptr(), partial.data_ptr(),
                                           B, H, W_, N, w.shape[0], K, splitk, bm, sbuf.data_ptr(), cap, C.byref(rpi),
                                           dtype, hp.stream()))
    assert _lib.lib().k22_debug_counter(b"stream_launches") > before
    ref = a.float() @ W.to(T_).float().T + bias + r.float()
    close(out.float(), ref, dtype, "stream gemm + stats out")
    st = sbuf[: B * rpi.value].view(B, rpi.value, N, 2).double().sum(1)
    o = out.double().view(B, H * W_, N)
    assert torch.isfinite(st).all()
    assert (st[..., 0] - o.sum(1)).abs().max().item() <= 1e-3 * (o.abs().sum(1).max().item() + 1)
    assert (st[..., 1] - (o * o).sum(1)).abs().max().item() <= 1e-3 * ((o * o).sum(1).max().item() + 1)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,H,T,S,K,bm,splitk", [(2, 2, 144, 87, 128, 160, 1), (1, 6, 100, 5, 192, 160, 3), (2, 24, 144, 87, 1536, 160, 1),
                                                 (2, 18, 576, 87, 1152, 288, 1)])
def test_stream_qkv_projection_writes_attention_operands(dtype, B, H, T, S, K, bm, splitk):
    T_ = hp.tdt(dtype)
    Cc, Tkp = 64 * H, (S + T + 63) // 64 * 64
    x, W, bias = rnd(B * T, K, seed=1), rnd(3 * Cc, K, seed=2, scale=K ** -0.5), rnd(3 * Cc, seed=3)
    xt, wt = x.to(T_).contiguous(), W.to(T_).contiguous()
    q = torch.empty(B * T, Cc, dtype=T_, device="cuda")
    kall = torch.full((B, H, Tkp, 64), 7.0, dtype=T_, device="cuda")
    vtall = torch.full((B, H, 64, Tkp), 7.0, dtype=T_, device="cuda")
    partial = torch.empty(splitk * B * T * 3 * Cc + 64, dtype=torch.float32, device="cuda")
    before = _lib.lib().k22_debug_counter(b"stream_launches")
    _lib.check(_lib.lib().k22_qkv_project_stream(xt.data_ptr(), wt.data_ptr(), bias.data_ptr(), q.data_ptr(), kall.data_ptr(), vtall.data_ptr(),
                                                 partial.data_ptr(), B, H, T, S, K, bm, splitk, dtype, hp.stream()))
    assert _lib.lib().k22_debug_counter(b"stream_launches") > before
    ref = (xt.float() @ wt.float().T + bias).view(B, T, 3, H, 64)
    close(q.float().view(B, T, H, 64), ref[:, :, 0], dtype, "q")
    close(kall.float()[:, :, S:S + T], ref[:, :, 1].permute(0, 2, 1, 3), dtype, "k")
    close(vtall.float()[:, :, :, S:S + T], ref[:, :, 2].permute(0, 2, 3, 1), dtype, "v^T")
    assert (kall[:, :, :S] == 7).all() and (kall[:, :, S + T:] == 7).all()
    assert (vtall[:, :, :, :S] == 7).all() and (vtall[:, :, :, S + T:] == 7).all()
