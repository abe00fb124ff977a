"""GPU parity of the MoVQ decoder engine (k22_movq_* through the drop-in MoVQDecoderHIP) against the golden outputs of
the REFERENCE's MOVQ.decode (kandinsky2/vqgan/autoencoder.py:182-185) and the uint8 epilogue of process_images.

Tolerances: fp32 engine 2e-4 of the output scale (exact-fp32 MFMA, different summation order); bf16 engine 6e-2 of
the output scale (about 60 bf16-rounded layers deep; reported, not an algorithmic difference).
"""
import os

import pytest
import torch

import kandinsky2_amd as k22
from oracle import movq_ref

pytestmark = pytest.mark.gpu


def _fixture(golden_dir, name):
    p = os.path.join(golden_dir, name + ".pt")
    if not os.path.exists(p):
        pytest.skip(f"{name}.pt not generated")
    return torch.load(p, weights_only=False)


_SD = {}


def _model(backend):
    arch = k22.MoVQArch(k22.MOVQ_CONFIG_2_1["ddconfig"])
    if "sd" not in _SD:
        _SD["sd"] = k22.init_movq_state_dict(arch, seed=0)
    m = k22.MoVQDecoderHIP(backend_dtype=backend)
    m.load_state_dict(_SD["sd"], strict=True)
    return arch, m.to("cuda")


@pytest.mark.parametrize("name", ["movq_small", "movq_wide"])
@pytest.mark.parametrize("backend,tol", [(torch.float32, 2e-4), (torch.bfloat16, 6e-2), (torch.float16, 8e-3)])
def test_movq_decode_vs_reference_golden(golden_dir, name, backend, tol):
    fx = _fixture(golden_dir, name)
    arch, m = _model(backend)
    g = torch.Generator().manual_seed(fx["seed_z"])
    z = torch.randn(fx["B"], 4, fx["h"], fx["w"], generator=g)
    out, u8 = m.decode(z.cuda(), return_uint8=True)
    ref = fx["out"]
    scale = ref.abs().max().item()
    err = (out.cpu() - ref).abs().max().item()
    print(f"{name} {backend}: max|d|={err:.3e} scale={scale:.3f}")
    assert err <= tol * scale
    # uint8 epilogue: identical to process_images applied to the engine's own float output ...
    assert torch.equal(u8.cpu(), movq_ref.process_images_u8(out.cpu()))
    # ... and, on the fp32 path, within one grey level of the reference image (ties at .5 can round either way)
    if backend == torch.float32:
        assert (u8.cpu().int() - fx["out_u8"].int()).abs().max().item() <= 1


def _compact_err(out, c):
    """max-abs distance on the stored sub-grid, row band and column band of a compact fixture (make_golden._compact)."""
    s = c["stride"]
    e = (out[..., ::s, ::s] - c["sub"]).abs().max().item()
    e = max(e, (out[..., c["r0"]: c["r0"] + c["rows"].shape[-2], :] - c["rows"]).abs().max().item())
    return max(e, (out[..., :, c["c0"]: c["c0"] + c["cols"].shape[-1]] - c["cols"]).abs().max().item())


@pytest.mark.parametrize("name", ["movq_256px", "movq_768px"])
@pytest.mark.parametrize("backend,tol", [(torch.float32, 2e-4), (torch.bfloat16, 6e-2), (torch.float16, 8e-3)])
def test_movq_decode_real_sizes_vs_reference_golden(golden_dir, name, backend, tol):
    """MOVQ.decode at 32x32 latents (256 px, attention over T = 1024) and at C2's 96x96 latents (768 px, T = 9216: the
    wide-image convolutions and the long softmax rows the small fixtures never reach).  The float output is compared on the
    fixture's sub-grid + full-resolution row / column bands, the uint8 image everywhere."""
    fx = _fixture(golden_dir, name)
    arch, m = _model(backend)
    g = torch.Generator().manual_seed(fx["seed_z"])
    z = torch.randn(fx["B"], 4, fx["h"], fx["w"], generator=g)
    out, u8 = m.decode(z.cuda(), return_uint8=True)
    out, u8 = out.cpu(), u8.cpu()
    scale = fx["absmax"]
    err = _compact_err(out, fx["out_compact"])
    du8 = (u8.int() - fx["out_u8"].int()).abs()
    print(f"{name} {backend}: max|d|={err:.3e} = {err / scale:.3e} of scale {scale:.3f}; uint8: max diff {du8.max().item()}, "
          f"{(du8 > 0).float().mean().item() * 100:.3f} % of the bytes differ")
    assert err <= tol * scale
    assert torch.equal(u8, movq_ref.process_images_u8(out))
    if backend == torch.float32:
        assert du8.max().item() <= 1
    if backend == torch.float16:
        # the drivers decode in fp16 beside 16-bit UNets, as the reference does under use_fp16 (kandinsky2_1_model.py:92-94, 287-288).  Yardstick:
        # the REFERENCE'S OWN MOVQ.half() decode of this very latent against its fp32 decode (oracle/ref_movq_half_drift.py, CPU): the engine
        # must be at least as close to the fp32 reference image as that
        import json
        yp = os.path.join(golden_dir, "ref_movq_half_drift.json")
        if os.path.exists(yp):
            y = json.load(open(yp))["cases"][name]
            frac = (du8 > 0).float().mean().item()
            print(f"{name}: reference's own half decode: {y['rel']:.3e} of scale, uint8 max diff {y['uint8_max_diff']}, {100 * y['uint8_frac_differ']:.3f} % differ; "
                  f"fp16 engine: {err / scale:.3e}, {du8.max().item()}, {100 * frac:.3f} %")
            assert err / scale <= y["rel"] and du8.max().item() <= y["uint8_max_diff"] and frac <= y["uint8_frac_differ"]
            # regression bound of THIS build (round 6: no packed-fp32 instructions, so multiply-adds contract differently than in rounds 3-5,
            # whose build read 3 levels / 11.5 % at 768 px): 6 levels, 14 % of the bytes off by one; the yardstick above reads 11 / 21 %
            print(f"{name}: bytes off by more than one level: {100 * (du8 > 1).float().mean().item():.4f} %")
            assert du8.max().item() <= 6 and (du8 > 1).float().mean().item() <= 5e-3


def test_movq_decode_is_deterministic_and_batch_independent(golden_dir):
    """Images are independent chains: decoding a batch equals decoding its elements one by one."""
    arch, m = _model(torch.float32)
    g = torch.Generator().manual_seed(11)
    z = torch.randn(2, 4, 8, 8, generator=g).cuda()
    both = m.decode(z)
    again = m.decode(z)
    assert torch.equal(both, again)
    one = m.decode(z[1:2])
    assert (one - both[1:2]).abs().max().item() <= 1e-5 * both.abs().max().item()


def test_movq_rejects_cpu_tensor():
    arch, m = _model(torch.float32)
    with pytest.raises(RuntimeError):
        m.decode(torch.zeros(1, 4, 8, 8))


# ---- encoder (MOVQ.encode: the img2img / inpainting pre-step) -----------------------------------------------
def _encoder(backend):
    arch = k22.MoVQArch(k22.MOVQ_CONFIG_2_1["ddconfig"])
    if "sd_enc" not in _SD:
        _SD["sd_enc"] = k22.init_movq_encoder_state_dict(arch, seed=0)
    m = k22.MoVQEncoderHIP(backend_dtype=backend)
    m.load_state_dict(_SD["sd_enc"], strict=True)
    return arch, m.to("cuda")


@pytest.mark.parametrize("name", ["movq_enc_small", "movq_enc_wide", "movq_enc_256px", "movq_enc_768px"])
@pytest.mark.parametrize("backend,tol", [(torch.float32, 2e-4), (torch.bfloat16, 6e-2), (torch.float16, 8e-3)])
def test_movq_encode_vs_reference_golden(golden_dir, name, backend, tol):
    """Encoder.forward + quant_conv: plain GroupNorm ResnetBlocks, single-head attention at the last level, Downsample as
    the stride-1 convolution gathered at the odd positions (asymmetric (0,1,0,1) padding)."""
    fx = _fixture(golden_dir, name)
    arch, m = _encoder(backend)
    g = torch.Generator().manual_seed(fx["seed_x"])
    x = torch.randn(fx["B"], 3, fx["H"], fx["W"], generator=g).clamp(-2, 2) * 0.5
    out = m.encode(x.cuda())
    ref = fx["out"]
    scale = ref.abs().max().item()
    err = (out.cpu() - ref).abs().max().item()
    print(f"{name} {backend}: max|d|={err:.3e} scale={scale:.3f}")
    assert out.shape == ref.shape and err <= tol * scale


def test_movq_encode_then_decode_shapes_and_batch_independence():
    arch, m = _encoder(torch.float32)
    g = torch.Generator().manual_seed(12)
    x = (torch.randn(2, 3, 64, 64, generator=g) * 0.5).cuda()
    both = m.encode(x)
    assert both.shape == (2, 4, 8, 8) and torch.equal(both, m.encode(x))
    one = m.encode(x[1:2])
    assert (one - both[1:2]).abs().max().item() <= 1e-5 * both.abs().max().item()
    with pytest.raises(RuntimeError):
        m.encode(torch.zeros(1, 3, 64, 64))


def test_prestep_prepare_mask_and_q_sample_vs_reference_golden(golden_dir):
    """k22_prepare_mask (the reference's O(h*w) Python loop as one gather kernel) bit-exact on every golden mask, corner / soft /
    multi-channel cases included; q_sample within one ulp of the reference's fp32 expression (device multiply-add order)."""
    fx = _fixture(golden_dir, "prestep")
    for m, want in zip(fx["masks"], fx["mask_out"]):
        got = k22.prestep.prepare_mask(m.cuda())
        assert got.shape == want.shape and torch.equal(got.cpu(), want)
    for t, want in fx["q"].items():
        got = k22.prestep.q_sample(fx["x"].cuda(), torch.tensor(t), noise=fx["noise"].cuda())
        assert torch.allclose(got.cpu(), want, rtol=1e-6, atol=1e-6)
    with pytest.raises(RuntimeError):
        k22.prestep.prepare_mask(fx["masks"][0])


def test_img2img_init_latent_pipeline():
    """encode -> scale -> q_sample at the loop's start step (kandinsky2_1_model.py:458-469) against the oracle pieces."""
    from oracle import movq_ref, prestep_ref
    from kandinsky2_amd.movq import movq_encoder_blocks
    arch, enc = _encoder(torch.float32)
    g = torch.Generator().manual_seed(21)
    img = torch.randn(1, 3, 64, 64, generator=g) * 0.5
    noise = torch.randn(1, 4, 8, 8, generator=g)
    d = k22.create_gaussian_diffusion(**dict(k22.DIFFUSION_CONFIG_2_1, timestep_respacing="50"))
    got = k22.prestep.img2img_init_latent(enc, img.cuda(), 0.18215, d.timestep_map, d.num_timesteps, 0.6, noise=noise.cuda())
    blocks, last = movq_encoder_blocks(arch)
    with torch.no_grad():
        lat = movq_ref.movq_encode(_SD["sd_enc"], blocks, last, img) * 0.18215
    start = int(d.num_timesteps * (1 - 0.6))
    want = prestep_ref.q_sample(lat, d.timestep_map[start - 1], noise=noise)
    assert (got.cpu() - want).abs().max().item() <= 2e-4 * want.abs().max().item()
