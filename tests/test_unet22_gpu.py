"""Kandinsky 2.2 decoder path on the GPU (row a19, config C5): UNet2DConditionHIP (diffusers state_dict keys, image-only
conditioning head, ControlNet-depth hint stack), DDPMSchedulerHIP.step and the fused decoder loop, against oracle/unet22_ref.py.

PARITY UNPINNED: the oracle restates diffusers' arithmetic from memory of its source (diffusers is absent from the reference
tree and from this image; oracle/unet22_ref.py header).  These tests therefore check that the ENGINE computes what the written
restatement says - key mapping, head, hint stack, scheduler - not that the restatement equals diffusers.
Tolerances as for 2.1: fp32 engine 2e-4 of the output scale per forward, 1e-3 max-abs on the final latent.
"""
import pytest
import torch

import kandinsky2_amd as k22
from oracle import unet22_ref

pytestmark = pytest.mark.gpu

_SD = {}


def _weights(controlnet=False, full=False):
    key = (controlnet, full)
    if key not in _SD:
        _SD.clear()
        cfg = k22.UNET_CONFIG_2_2 if full else k22.tiny_unet22_config()
        arch = k22.make_arch22(cfg, controlnet=controlnet)
        _SD[key] = (cfg, arch, k22.init_unet22_state_dict(arch, seed=0))
    return _SD[key]


def _unet(controlnet, backend, full=False, use_graph=True):
    cfg, arch, sd = _weights(controlnet, full)
    m = k22.UNet2DConditionHIP(arch, backend_dtype=backend, use_graph=use_graph)
    m.load_state_dict(sd)
    return cfg, sd, m.to("cuda").eval()


def _inputs(B, h, w, seed=4, hint=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 4, h, w, generator=g)
    emb = torch.randn(B, 1280, generator=g)
    hi = torch.rand(B, 3, 8 * h, 8 * w, generator=g) if hint else None
    return x, emb, hi


@pytest.mark.parametrize("controlnet", [False, True])
@pytest.mark.parametrize("backend,tol", [(torch.float32, 2e-4), (torch.bfloat16, 2.5e-2), (torch.float16, 3.2e-3)])
def test_unet22_forward_vs_oracle(controlnet, backend, tol):
    cfg, sd, m = _unet(controlnet, backend)
    B, h, w = 4, 16, 24
    x, emb, hint = _inputs(B, h, w, hint=controlnet)
    t = torch.tensor([980.0, 500.0, 20.0, 0.0])
    ref = unet22_ref.unet22_forward(sd, cfg, x, t, emb, hint)
    ack = {"image_embeds": emb.cuda()}
    if controlnet:
        ack["hint"] = hint.cuda()
    out = m(sample=x.cuda(), timestep=t.cuda(), encoder_hidden_states=None, added_cond_kwargs=ack, return_dict=False)[0].cpu()
    scale = ref.abs().max().item()
    err = (out - ref).abs().max().item()
    print(f"unet22 {'controlnet ' if controlnet else ''}{backend}: max|d| {err:.3e} = {err / scale:.3e} of scale {scale:.3f} (oracle unpinned)")
    assert out.shape == (B, 8, h, w) and err <= tol * scale
    # scalar timestep + return_dict, cached conditioning, and the module attributes the diffusers pipelines read
    o2 = m(x.cuda(), 500, added_cond_kwargs=ack).sample
    assert o2.shape == out.shape and m.config.in_channels == (8 if controlnet else 4) and m.device.type == "cuda" and m.dtype == torch.float32
    with pytest.raises(ValueError):
        m(x.cuda(), 500, encoder_hidden_states=torch.zeros(1).cuda(), added_cond_kwargs=ack)
    with pytest.raises(ValueError):
        m(x.cuda(), 500, added_cond_kwargs={"image_embeds": emb[:2].cuda()} if not controlnet else {"image_embeds": emb.cuda(), "hint": hint[:, :, :8].cuda()})


SCHED_VARIANTS = {
    "config_2_2": (k22.SCHEDULER_CONFIG_2_2, unet22_ref.SCHED_2_2),                              # fixed_small, no clipping (what the notebook log implies)
    "learned_range": (k22.SCHEDULER_CONFIG_2_2_LEARNED_RANGE, unet22_ref.SCHED_2_2_LEARNED_RANGE),  # round 2's assumption
    "diffusers_defaults": ({}, {}),                                                               # betas 1e-4 .. 0.02, fixed_small, clip +-1
    "other": (dict(k22.SCHEDULER_CONFIG_2_2, beta_schedule="scaled_linear", variance_type="fixed_large", timestep_spacing="trailing",
                   clip_sample=True, clip_sample_range=1.5),
              dict(unet22_ref.SCHED_2_2, beta_schedule="scaled_linear", variance_type="fixed_large", timestep_spacing="trailing",
                   clip_sample=True, clip_sample_range=1.5)),
}


@pytest.mark.parametrize("variant", list(SCHED_VARIANTS))
def test_ddpm_scheduler_step_vs_oracle(variant):
    """DDPMScheduler.step as one k22_sampler_step launch at every timestep of a 10-step schedule, for the scheduler_config.json
    variants (fixed_small = variance channels ignored; learned_range; clipping on / off; spacing).  Oracle unpinned."""
    cfg_hip, cfg_ref = SCHED_VARIANTS[variant]
    g = torch.Generator().manual_seed(5)
    N, h, w = 2, 8, 8
    sch = k22.DDPMSchedulerHIP.from_config(cfg_hip).set_timesteps(10)
    ref = unet22_ref.RefDDPMScheduler(10, cfg_ref)
    assert sch.timesteps.tolist() == ref.timesteps.tolist()
    for t in sch.timesteps.tolist():
        mo = torch.randn(N, 8, h, w, generator=g)
        mo[:, 4:] = mo[:, 4:].clamp(-1, 1)
        x, nz = torch.randn(N, 4, h, w, generator=g) * 1.5, torch.randn(N, 4, h, w, generator=g)
        want = ref.step(mo if ref.learned else mo[:, :4], t, x, nz)       # the pipeline drops the variance channels for fixed variance
        got = sch.step(mo.cuda(), t, x.cuda(), noise=nz.cuda()).prev_sample.cpu()
        assert (got - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item()), t
    # diffusers' randn_tensor accepts a CPU generator for a GPU sample: drawn on the CPU, moved over
    gc = torch.Generator().manual_seed(7)
    a = sch.step(mo.cuda(), 100 if 100 in sch._row else sch.timesteps.tolist()[5], x.cuda(), generator=gc).prev_sample
    assert torch.isfinite(a).all()


@pytest.mark.parametrize("variant", ["config_2_2", "learned_range"])
@pytest.mark.parametrize("controlnet", [False, True])
def test_decoder_loop_vs_oracle_fp32(controlnet, variant):
    """KandinskyV22[Controlnet]Pipeline denoising loop (CFG, conditional variance, DDPM step), 5 steps, final latent <= 1e-3."""
    cfg, sd, m = _unet(controlnet, torch.float32)
    bs, h, w, steps, gs = 2, 16, 16, 5, 4.0
    g = torch.Generator().manual_seed(6)
    lat = torch.randn(bs, 4, h, w, generator=g)
    pos, neg = torch.randn(bs, 1280, generator=g), torch.randn(bs, 1280, generator=g)
    hint = torch.rand(bs, 3, 8 * h, 8 * w, generator=g) if controlnet else None
    nz = torch.randn(steps, bs, 4, h, w, generator=g)
    cfg_hip, cfg_ref = SCHED_VARIANTS[variant]
    want = unet22_ref.decoder_loop(lambda xx, t, e, hh: unet22_ref.unet22_forward(sd, cfg, xx, t, e, hh), lat, pos, neg, steps, gs, nz, hint,
                                   sched_cfg=cfg_ref)
    marc = k22.MoVQArch(k22.MOVQ_CONFIG_2_1["ddconfig"])
    movq = k22.MoVQDecoderHIP(backend_dtype=torch.float32)
    movq.load_state_dict(k22.init_movq_state_dict(marc, seed=0), strict=True)
    dec = k22.pipeline22.KandinskyV22DecoderHIP(m, movq.to("cuda"), scheduler=k22.DDPMSchedulerHIP.from_config(cfg_hip))
    got = dec(pos.cuda(), neg.cuda(), height=8 * h, width=8 * w, num_inference_steps=steps, guidance_scale=gs,
              hint=None if hint is None else hint.cuda(), latents=lat.cuda(), noise_seq=nz.cuda(), output_type="latent").cpu()
    err = (got - want).abs().max().item()
    print(f"2.2 decoder loop {'controlnet ' if controlnet else ''}fp32: final latent max|d| {err:.3e} (scale {want.abs().max().item():.2f}, oracle unpinned)")
    assert err <= 1e-3
    img = dec(pos.cuda(), neg.cuda(), height=8 * h, width=8 * w, num_inference_steps=2, guidance_scale=gs,
              hint=None if hint is None else hint.cuda(), output_type="uint8")
    assert img.shape == (bs, 8 * h, 8 * w, 3)


@pytest.mark.slow   # 17 s on an UNPINNED path (oracle/unet22_ref.py): skipped by K22_RUN_SLOW=0
def test_unet22_full_width_forward_vs_oracle_fp32():
    """The 1.25 B-parameter configuration (UNET_CONFIG_2_2) at 32x32 latents, one forward."""
    cfg, sd, m = _unet(False, torch.float32, full=True)
    n_params = sum(v.numel() for v in sd.values())
    x, emb, _ = _inputs(2, 32, 32, seed=8)
    t = torch.tensor([980.0, 40.0])
    ref = unet22_ref.unet22_forward(sd, cfg, x, t, emb)
    out = m(x.cuda(), t.cuda(), added_cond_kwargs={"image_embeds": emb.cuda()}).sample.cpu()
    scale = ref.abs().max().item()
    err = (out - ref).abs().max().item()
    print(f"unet22 full width ({n_params / 1e9:.3f} B params) fp32: max|d| {err:.3e} = {err / scale:.3e} of scale {scale:.3f} (oracle unpinned)")
    assert err <= 2e-4 * scale


def _movq_pair(backend=torch.float32):
    marc = k22.MoVQArch(k22.MOVQ_CONFIG_2_1["ddconfig"])
    sd = dict(k22.init_movq_state_dict(marc, seed=0))
    sd.update(k22.init_movq_encoder_state_dict(marc, seed=0))
    dec, enc = k22.MoVQDecoderHIP(backend_dtype=backend), k22.MoVQEncoderHIP(backend_dtype=backend)
    dec.load_state_dict(sd, strict=True)
    enc.load_state_dict(sd, strict=True)
    return sd, dec.to("cuda"), enc.to("cuda")


def test_scheduler_add_noise_and_blend_kernel():
    """k22_blend_noised: DDPMScheduler.add_noise and the inpainting re-imposition, against the torch expressions"""
    from kandinsky2_amd import _lib
    g = torch.Generator().manual_seed(9)
    sch = k22.DDPMSchedulerHIP.from_config(k22.SCHEDULER_CONFIG_2_2).set_timesteps(10)
    ref = unet22_ref.RefDDPMScheduler(10)
    x, nz = torch.randn(3, 4, 12, 20, generator=g), torch.randn(3, 4, 12, 20, generator=g)
    for t in (900, 400, 0):
        got = sch.add_noise(x.cuda(), nz.cuda(), torch.tensor([t])).cpu()
        # the oracle keeps diffusers' float32 alphas_cumprod, the scheduler fp64 tables: a few ulp of the coefficients
        assert (got - unet22_ref.add_noise(ref, x, nz, t)).abs().max().item() <= 5e-6
    m = (torch.rand(1, 1, 12, 20, generator=g) > 0.4).float()
    cur = torch.randn(1, 4, 12, 20, generator=g)
    out = torch.empty(1, 4, 12, 20, device="cuda")
    a = float(ref.alphas_cumprod[300])
    xc, ic, nc, mc = cur.cuda(), x[:1].contiguous().cuda(), nz[:1].contiguous().cuda(), m.cuda()
    _lib.check(_lib.lib().k22_blend_noised(xc.data_ptr(), ic.data_ptr(), nc.data_ptr(), mc.data_ptr(), a ** 0.5, (1 - a) ** 0.5, out.data_ptr(),
                                           1, 4, 240, 0, _lib.current_stream()))
    want = m * (a ** 0.5 * x[:1] + (1 - a) ** 0.5 * nz[:1]) + (1 - m) * cur
    assert (out.cpu() - want).abs().max().item() <= 5e-6
    with pytest.raises(RuntimeError):
        _lib.check(_lib.lib().k22_blend_noised(None, ic.data_ptr(), nc.data_ptr(), mc.data_ptr(), 1.0, 0.0, out.data_ptr(), 1, 4, 240, 0, _lib.current_stream()))


def test_img2img_decoder_vs_oracle_fp32():
    """KandinskyV22Img2ImgPipeline path: movq.encode -> add_noise -> loop over timesteps[t_start:] (oracle unpinned)."""
    cfg, sd, m = _unet(False, torch.float32)
    msd, dec, enc = _movq_pair()
    bs, h, w, steps, strength, gs = 2, 16, 16, 10, 0.5, 4.0
    g = torch.Generator().manual_seed(10)
    img = (torch.randn(1, 3, 8 * h, 8 * w, generator=g) * 0.5).clamp(-1, 1)
    pos, neg = torch.randn(bs, 1280, generator=g), torch.randn(bs, 1280, generator=g)
    nz0, nzs = torch.randn(bs, 4, h, w, generator=g), torch.randn(steps, bs, 4, h, w, generator=g)
    lat0 = enc.encode(img.cuda()).cpu().repeat(bs, 1, 1, 1)          # the MoVQ encoder has its own parity tests (test_movq_gpu.py)
    want = unet22_ref.img2img_loop(lambda xx, t, e, hh: unet22_ref.unet22_forward(sd, cfg, xx, t, e, hh), lat0, pos, neg, steps, strength, gs, nz0, nzs)
    pipe = k22.pipeline22.KandinskyV22Img2ImgDecoderHIP(m, dec, enc)
    got = pipe(pos.cuda(), neg.cuda(), image=img.cuda(), height=8 * h, width=8 * w, num_inference_steps=steps, guidance_scale=gs, strength=strength,
               noise=nz0.cuda(), noise_seq=nzs.cuda(), output_type="latent").cpu()
    err = (got - want).abs().max().item()
    print(f"2.2 img2img decoder fp32: final latent max|d| {err:.3e} (scale {want.abs().max().item():.2f}, oracle unpinned)")
    assert err <= 1e-3
    u8 = pipe(pos.cuda(), neg.cuda(), image=img.cuda(), height=8 * h, width=8 * w, num_inference_steps=4, strength=0.5, output_type="uint8")
    assert u8.shape == (bs, 8 * h, 8 * w, 3)
    with pytest.raises(ValueError):
        pipe(pos.cuda(), neg.cuda(), image=img.cuda(), height=8 * h, width=8 * w, num_inference_steps=4, strength=0.1, output_type="latent")


def test_inpaint_decoder_vs_oracle_fp32():
    """KandinskyV22InpaintPipeline path on the 9-channel UNet: [latents | masked image latents | mask] input, per-step re-imposition of
    the known region at the next noise level (oracle unpinned)."""
    from oracle import prestep_ref
    import torch.nn.functional as F
    cfgu = k22.tiny_unet22_config()
    arch = k22.make_arch22(cfgu, inpainting=True)
    sd = k22.init_unet22_state_dict(arch, seed=0)
    m = k22.UNet2DConditionHIP(arch, backend_dtype=torch.float32)
    m.load_state_dict(sd)
    m = m.to("cuda").eval()
    msd, dec, enc = _movq_pair()
    bs, h, w, steps, gs = 2, 16, 16, 5, 4.0
    g = torch.Generator().manual_seed(11)
    img = (torch.randn(1, 3, 8 * h, 8 * w, generator=g) * 0.5).clamp(-1, 1)
    mask_px = torch.ones(8 * h, 8 * w)
    mask_px[24:90, 40:100] = 0.0
    pos, neg = torch.randn(bs, 1280, generator=g), torch.randn(bs, 1280, generator=g)
    x_T, nzs = torch.randn(bs, 4, h, w, generator=g), torch.randn(steps, bs, 4, h, w, generator=g)
    lat0 = enc.encode(img.cuda()).cpu()
    mlat = prestep_ref.prepare_mask(F.interpolate(mask_px[None, None], (h, w), mode="nearest"))
    cfg9 = dict(cfgu, in_channels=9)
    want = unet22_ref.inpaint_loop(lambda xx, t, e, hh: unet22_ref.unet22_forward(sd, cfg9, xx, t, e, hh), lat0, mlat, x_T, pos, neg, steps, gs, nzs)
    pipe = k22.pipeline22.KandinskyV22InpaintDecoderHIP(m, dec, enc)
    got = pipe(pos.cuda(), neg.cuda(), image=img.cuda(), mask_image=mask_px.numpy(), height=8 * h, width=8 * w, num_inference_steps=steps,
               guidance_scale=gs, latents=x_T.cuda(), noise_seq=nzs.cuda(), output_type="latent").cpu()
    err = (got - want).abs().max().item()
    keep = mlat[0, 0] == 1
    print(f"2.2 inpainting decoder fp32: final latent max|d| {err:.3e} (scale {want.abs().max().item():.2f}, oracle unpinned)")
    assert err <= 1e-3
    assert (got[:, :, keep] - lat0[:, :, keep]).abs().max().item() <= 1e-5          # the known region ends un-noised
    inv = pipe(pos.cuda(), neg.cuda(), image=img.cuda(), mask_image=1.0 - mask_px.numpy(), repaint_white=True, height=8 * h, width=8 * w,
               num_inference_steps=steps, guidance_scale=gs, latents=x_T.cuda(), noise_seq=nzs.cuda(), output_type="latent").cpu()
    assert torch.equal(inv, got)


def test_kandinsky2_2_wrapper_task_types():
    """Kandinsky2_2HIP: the reference's (task_type -> pipeline, UNet) table and its generate_* / mix_images signatures."""
    msd, _, _ = _movq_pair()
    H = 128
    img = torch.zeros(1, 3, H, H).cuda()
    for task, inp in (("text2img", False), ("img2img", False), ("inpainting", True)):
        arch = k22.make_arch22(k22.tiny_unet22_config(), inpainting=inp)
        mdl = k22.pipeline22.Kandinsky2_2HIP("cuda", task, unet_state_dict=k22.init_unet22_state_dict(arch, seed=0), movq_state_dict=msd,
                                             unet_config=k22.tiny_unet22_config(), conditioner="seeded", backend_dtype=torch.float32)
        if task == "text2img":
            out = mdl.generate_text2img("a cat", batch_size=2, decoder_steps=3, h=H, w=H, output_type="uint8")
            mix = mdl.mix_images(["a cat", img], [0.3, 0.7], batch_size=1, decoder_steps=2, h=H, w=H, output_type="uint8")
            assert mix.shape == (1, H, H, 3)
            with pytest.raises(ValueError):
                mdl.generate_img2img("a cat", img)
        elif task == "img2img":
            out = mdl.generate_img2img("a cat", img, strength=0.5, batch_size=2, decoder_steps=4, h=H, w=H, output_type="uint8")
        else:
            mk = torch.ones(H, H).numpy()
            mk[:, :64] = 0
            out = mdl.generate_inpainting("a cat", img, mk, batch_size=2, decoder_steps=3, h=H, w=H, output_type="uint8")
        assert out.shape == (2, H, H, 3) and out.dtype.name == "uint8"
    with pytest.raises(ValueError):
        k22.pipeline22.Kandinsky2_2HIP("cuda", "superres", unet_state_dict={}, movq_state_dict={})


def test_kandinsky2_2_builds_itself_from_a_cache_dir(tmp_path):
    """kandinsky2_2_model.py:26-41 through local files: the decoder repository's unet/ (safetensors + config.json), movq/ (.bin) and
    scheduler/scheduler_config.json drive architecture, weights and scheduler; the result equals the directly constructed model bit
    for bit, and a learned_range scheduler_config.json changes the samples (the file decides, nothing is hard-coded)."""
    import json
    from safetensors.torch import save_file
    msd, _, _ = _movq_pair()
    cfgu = k22.tiny_unet22_config()
    arch = k22.make_arch22(cfgu)
    usd = k22.init_unet22_state_dict(arch, seed=0)
    root = tmp_path / "kandinsky-2-2-decoder"
    for sub in ("unet", "movq", "scheduler"):
        (root / sub).mkdir(parents=True)
    save_file({k: v.contiguous() for k, v in usd.items()}, str(root / "unet" / "diffusion_pytorch_model.safetensors"))
    json.dump(dict({k: (list(v) if isinstance(v, tuple) else v) for k, v in cfgu.items()}, _class_name="UNet2DConditionModel"), open(root / "unet" / "config.json", "w"))
    torch.save(msd, str(root / "movq" / "diffusion_pytorch_model.bin"))
    H = 128
    g = torch.Generator().manual_seed(21)
    lat, nz = torch.randn(1, 4, H // 8, H // 8, generator=g).cuda(), torch.randn(3, 1, 4, H // 8, H // 8, generator=g).cuda()
    outs = {}
    for name, sc in (("file", k22.SCHEDULER_CONFIG_2_2), ("learned", k22.SCHEDULER_CONFIG_2_2_LEARNED_RANGE)):
        json.dump(dict(sc, _diffusers_version="0.18.0.dev0"), open(root / "scheduler" / "scheduler_config.json", "w"))
        mdl = k22.get_kandinsky2("cuda", task_type="text2img", cache_dir=str(tmp_path), model_version="2.2", conditioner="seeded",
                                 backend_dtype=torch.float32)
        assert mdl.decoder.scheduler.config.variance_type == sc.get("variance_type", "fixed_small")
        outs[name] = mdl.generate_text2img("a cat", batch_size=1, decoder_steps=3, h=H, w=H, latents=lat, noise_seq=nz, output_type="latent")
    direct = k22.pipeline22.Kandinsky2_2HIP("cuda", "text2img", unet_state_dict=usd, movq_state_dict=msd, unet_config=cfgu, conditioner="seeded",
                                            backend_dtype=torch.float32)
    want = direct.generate_text2img("a cat", batch_size=1, decoder_steps=3, h=H, w=H, latents=lat, noise_seq=nz, output_type="latent")
    assert torch.equal(outs["file"], want)
    assert not torch.equal(outs["learned"], want)
    with pytest.raises(ValueError, match="conditioner"):
        k22.get_kandinsky2("cuda", task_type="text2img", cache_dir=str(tmp_path), model_version="2.2")
