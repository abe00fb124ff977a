"""GPU parity of the skinny-M weight-streaming GEMM family (kandinsky-2_amd/csrc/skinny.hip + small_attention_kernel in attention.hip),
through the C ABI, against plain PyTorch fp32 references on the same (dtype-rounded) operands: the prior transformer's Linears
(kandinsky2/model/prior.py:57-83), its LayerNorm + residual (prior.py:48-54, 105-127) and its attention with the additive mask of
prior.py:86-102, 262-263.

Tolerance: operands pre-rounded to T, fp32 accumulation, ONE rounding of the output - 1.2e-2 of the output scale for bf16, 1.5e-3 for
fp16 (as test_kernels_gpu.py); fp32 partials against the fp32 reference at 2e-5 of the scale (summation order only)."""
import pytest
import torch

import helpers as hp
from kandinsky2_amd import _lib
from test_kernels_gpu import rnd

pytestmark = pytest.mark.gpu
DT = [_lib.K22_BF16, _lib.K22_F16]
TOL = {_lib.K22_BF16: 1.2e-2, _lib.K22_F16: 1.5e-3}


def afrag_index(M, K, device="cpu"):
    """flat element offset of (m, k) in the A-fragment tensor [K/64][MA][4][64][8] (skinny.hip: afrag_off)"""
    MA = (M + 31) // 32
    m = torch.arange(M, device=device)[:, None]
    k = torch.arange(K, device=device)[None, :]
    return ((((k >> 6) * MA + (m >> 5)) * 4 + ((k >> 4) & 3)) * 64 + (m & 31) + 32 * ((k >> 3) & 1)) * 8 + (k & 7)


def from_afrag(buf, M, K):
    return buf.reshape(-1)[afrag_index(M, K, buf.device).reshape(-1)].reshape(M, K)


def pack_a(a):
    M, K = a.shape
    out = torch.zeros(_lib.lib().k22_afrag_bytes(M, K) // 2, dtype=a.dtype, device=a.device)
    _lib.check(_lib.lib().k22_afrag_pack(a.data_ptr(), K, out.data_ptr(), M, K, _lib.K22_BF16, hp.stream()))
    return out


def pack_w(w):
    Npad, K = w.shape
    out = torch.empty_like(w)
    _lib.check(_lib.lib().k22_stream_repack(w.data_ptr(), out.data_ptr(), Npad, 1, K, _lib.K22_BF16, hp.stream()))
    return out


def test_afrag_pack_is_the_documented_permutation():
    M, K = 81, 192
    a = torch.arange(M * K, dtype=torch.int32).remainder(65521).to(torch.int16).reshape(M, K).cuda()
    out = pack_a(a)
    assert torch.equal(from_afrag(out, M, K).cpu(), a.cpu())


GEMM_CASES = [
    # M, N, K, splitk, epi, (mt, nb)
    (162, 384, 512, 1, 0, (0, 0)), (162, 384, 512, 1, 0, (6, 1)), (162, 384, 512, 1, 0, (3, 1)), (162, 256, 512, 1, 1, (3, 2)),
    (162, 128, 1024, 4, 2, (3, 2)), (162, 128, 1024, 3, 2, (6, 1)), (81, 192, 256, 1, 0, (0, 0)), (81, 128, 256, 2, 2, (2, 2)),
    (20, 64, 128, 1, 0, (1, 2)), (33, 200, 192, 1, 0, (2, 1)), (324, 256, 512, 1, 0, (3, 2)), (324, 128, 512, 2, 2, (6, 1)),
    (162, 6144, 2048, 1, 0, (0, 0)), (162, 2048, 2048, 4, 2, (0, 0)), (162, 8192, 2048, 1, 1, (0, 0)), (162, 2048, 8192, 4, 2, (0, 0)),
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K,splitk,epi,cfg", GEMM_CASES)
def test_skinny_gemm(dtype, M, N, K, splitk, epi, cfg):
    """ragged M (rows past M in the last m-atom are never stored), N not a multiple of 64, every epilogue, split-K with an uneven last range,
    two m-tiles per n-tile, the prior's four shapes at full size."""
    T = hp.tdt(dtype)
    a, w, bias = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    aT, wT = a.to(T).contiguous(), hp.pad_rows(w.to(T))
    Npad = wT.shape[0]
    af, wf = pack_a(aT), pack_w(wT)
    act = _lib.ACT_GELU if epi == 1 else _lib.ACT_NONE
    ref = aT.float() @ wT[:N].float().t()
    L = _lib.lib()
    if epi == 2:
        partial = torch.full((splitk, M, N), float("nan"), dtype=torch.float32, device="cuda")
        _lib.check(L.k22_skinny_gemm(af.data_ptr(), wf.data_ptr(), None, None, partial.data_ptr(), M, N, Npad, K, splitk, 2, 0, N, cfg[0], cfg[1], dtype, hp.stream()))
        got = partial.sum(0)
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        assert torch.isfinite(partial).all() and err <= 2e-5, err
        return
    ref = ref + bias
    if act == _lib.ACT_GELU:
        ref = torch.nn.functional.gelu(ref)
    if epi == 0:
        out = torch.full((M, N), float("nan"), dtype=T, device="cuda")
        _lib.check(L.k22_skinny_gemm(af.data_ptr(), wf.data_ptr(), bias.data_ptr(), out.data_ptr(), None, M, N, Npad, K, 1, 0, act, N, cfg[0], cfg[1], dtype, hp.stream()))
        got = out.float()
    else:
        out = torch.zeros(L.k22_afrag_bytes(M, N) // 2, dtype=T, device="cuda")
        _lib.check(L.k22_skinny_gemm(af.data_ptr(), wf.data_ptr(), bias.data_ptr(), out.data_ptr(), None, M, N, Npad, K, 1, 1, act, N, cfg[0], cfg[1], dtype, hp.stream()))
        got = from_afrag(out, M, N).float()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert torch.isfinite(got).all() and err <= TOL[dtype], err


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,splitk,ln", [(162, 2048, 4, True), (162, 2048, 0, True), (162, 512, 2, True), (81, 2048, 3, False), (5, 64, 1, True)])
def test_finish_ln(dtype, M, N, splitk, ln):
    """x += bias + sum of the partials (fp32, in place) and LayerNorm of the updated row in A-fragment order; splitk = 0: LayerNorm only."""
    T = hp.tdt(dtype)
    x = rnd(M, N, seed=1) * 3.0 + 0.5
    partial = rnd(max(splitk, 1), M, N, seed=2)
    bias, g, b = rnd(N, seed=3), 1.0 + 0.1 * rnd(N, seed=4), 0.1 * rnd(N, seed=5)
    want_x = x.clone() if splitk == 0 else x + bias + partial[:splitk].sum(0)
    want_y = torch.nn.functional.layer_norm(want_x, (N,), g, b, 1e-5)
    xg = x.clone()
    y = torch.zeros(_lib.lib().k22_afrag_bytes(M, N) // 2, dtype=T, device="cuda")
    _lib.check(_lib.lib().k22_finish_ln(partial.data_ptr() if splitk else None, splitk, bias.data_ptr() if splitk else None, xg.data_ptr(), N,
                                       g.data_ptr() if ln else None, b.data_ptr() if ln else None, y.data_ptr() if ln else None, M, N, 1e-5, dtype, hp.stream()))
    assert (xg - want_x).abs().max().item() <= 2e-6 * want_x.abs().max().item() + 1e-6
    if ln:
        got = from_afrag(y, M, N).float()
        err = (got - want_y).abs().max().item() / want_y.abs().max().item()
        assert err <= TOL[dtype] / 2, err


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("nsplit", [0, 2, 3])
@pytest.mark.parametrize("B,H,T_,causal,n_valid,frag", [(2, 32, 81, 1, 40, 1), (2, 8, 81, 1, 77, 0), (1, 4, 128, 0, 0, 0), (3, 2, 17, 1, 9, 1), (2, 4, 96, 1, 50, 1)])
def test_small_attention(dtype, B, H, T_, causal, n_valid, frag, nsplit):
    """softmax(q.k / 8 + mask) v per (image, head): causal mask + padding keys (prior.py:86-102, 262-263), T not a multiple of 32, both
    output layouts; nsplit > 0: q / k / v arrive as fp32 split-K partials + bias and are finished (one rounding to T) in the staging loads."""
    T = hp.tdt(dtype)
    C = H * 64
    part = bias = None
    if nsplit:
        part, bias = rnd(nsplit, B * T_, 3 * C, seed=5) * nsplit ** -0.5, rnd(3 * C, seed=6) * 0.1
        acc = part[0].clone()
        for s_ in range(1, nsplit):
            acc += part[s_]
        qkv = (acc + bias).to(T).contiguous()
    else:
        qkv = rnd(B * T_, 3 * C, seed=1).to(T).contiguous()
    kv_n = 77 if n_valid else 0
    kv_n = min(kv_n, T_)
    valid = None
    if kv_n:
        valid = torch.zeros(B, kv_n, dtype=torch.float32, device="cuda")
        for b in range(B):
            valid[b, :max(1, min(kv_n, n_valid - 3 * b))] = 1.0
    M = B * T_
    out = torch.zeros(_lib.lib().k22_afrag_bytes(M, C) // 2 if frag else M * C, dtype=T, device="cuda")
    _lib.check(_lib.lib().k22_small_attention(None if nsplit else qkv.data_ptr(), _lib.ptr(part), nsplit, _lib.ptr(bias), out.data_ptr(), frag, B, H, T_, causal, _lib.ptr(valid), kv_n, dtype, hp.stream()))
    got = (from_afrag(out, M, C) if frag else out.reshape(M, C)).float()
    q, k, v = [t.reshape(B, T_, H, 64).permute(0, 2, 1, 3) for t in qkv.float().split(C, dim=1)]
    w = torch.einsum("bhtc,bhsc->bhts", q, k) * 0.125
    mask = torch.zeros(B, 1, T_, T_, device="cuda")
    if causal:
        mask = mask + torch.full((T_, T_), float("-inf"), device="cuda").triu(1)
    if valid is not None:
        pad = torch.zeros(B, T_, device="cuda")
        pad[:, :kv_n] = torch.where(valid > 0, 0.0, float("-inf"))
        mask = mask + pad[:, None, None, :]
    p = torch.softmax(w + mask, dim=-1)
    want = torch.einsum("bhts,bhsc->bhtc", p, v).permute(0, 2, 1, 3).reshape(M, C)
    err = (got - want).abs().max().item() / want.abs().max().item()
    assert torch.isfinite(got).all() and err <= TOL[dtype], err
