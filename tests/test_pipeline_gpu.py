"""End-to-end drop-in driver on the GPU: Kandinsky2_1HIP.generate_* (the reference's Kandinsky2_1 methods,
kandinsky2/kandinsky2_1_model.py:135-548) chains prior -> CFG denoise loop -> MoVQ decode -> uint8 on the HIP engines and is
compared with the same chain on the CPU oracle (oracle/*_ref.py, pinned bit-identical to the reference's modules) on the same
seeded weights, conditioning and injected noise.  1/3-width UNet, 512x4 prior, full-width MoVQ, 128x128 px.

Tolerances (fp32 engines): final latent 1e-3 max-abs, uint8 image within one grey level of the oracle chain's.
"""
import copy

import numpy as np
import pytest
import torch

import kandinsky2_amd as k22
from oracle import diffusion_ref, movq_ref, prestep_ref, prior_ref, unet_ref

pytestmark = pytest.mark.gpu

H = W = 128          # pixels -> 16x16 latents
PRIOR_STEPS = 4
_CACHE = {}


def _weights(task_type):
    if task_type not in _CACHE:
        cfg = copy.deepcopy(k22.CONFIG_2_1)
        cfg["model_config"] = k22.tiny_model_config()
        hp = k22.tiny_prior_hparams()
        cfg["prior"]["params"]["model"]["hparams"] = hp
        g = torch.Generator().manual_seed(17)
        cm, cs = torch.randn(768, generator=g) * 0.1, torch.rand(768, generator=g) + 0.5
        cfg["prior"]["clip_mean_std_path"] = (cm, cs)
        marc = k22.MoVQArch(k22.MOVQ_CONFIG_2_1["ddconfig"])
        movq_sd = dict(k22.init_movq_state_dict(marc, seed=0))
        movq_sd.update(k22.init_movq_encoder_state_dict(marc, seed=0))
        cfg["image_enc_params"]["ckpt_path"] = movq_sd
        arch = k22.make_arch(cfg["model_config"], inpainting=task_type == "inpainting")
        unet_sd = k22.init_unet_state_dict(arch, seed=0)
        prior_sd = k22.init_prior_state_dict(hp, seed=0)
        _CACHE.clear()
        _CACHE[task_type] = (cfg, arch, unet_sd, hp, prior_sd, marc, movq_sd, cm, cs)
    return _CACHE[task_type]


def _pipe(task_type, backend=torch.float32):
    cfg, arch, unet_sd, hp, prior_sd, marc, movq_sd, cm, cs = _weights(task_type)
    return k22.Kandinsky2_1HIP(cfg, unet_sd, prior_sd, "cuda", task_type=task_type, conditioner="seeded", backend_dtype=backend)


def _oracle_image_emb(pipe, prompt, bs, scale, g):
    """generate_clip_emb + create_zero_img_emb on the oracle, with the noise the test injects into the HIP prior."""
    cfg, arch, unet_sd, hp, prior_sd, marc, movq_sd, cm, cs = _weights(pipe.task_type)
    cond = pipe.conditioner
    txt_feat, txt_seq, mask = cond.clip_text([prompt] * bs, "", "cpu")
    x_T = torch.randn(2 * bs, 768, generator=g)
    nz = torch.randn(PRIOR_STEPS, 2 * bs, 768, generator=g)
    emb = prior_ref.prior_sample(prior_sd, hp, txt_feat, txt_seq, mask, torch.full((bs,), float(scale)), PRIOR_STEPS, x_T, nz, cm[None], cs[None])
    zero = cond.zero_image_emb("cpu").repeat(bs, 1)
    return torch.cat([emb, zero], 0), x_T, nz


@pytest.mark.parametrize("sampler,steps", [("p_sampler", 6), ("ddim_sampler", 5), ("plms_sampler", 6)])
def test_generate_text2img_matches_the_oracle_chain_fp32(sampler, steps):
    pipe = _pipe("text2img")
    cfg, arch, unet_sd, hp, prior_sd, marc, movq_sd, cm, cs = _weights("text2img")
    bs, guidance, prompt = 2, 4.0, "a red cat, 4k photo"
    g = torch.Generator().manual_seed(5)
    image_emb, p_xT, p_nz = _oracle_image_emb(pipe, prompt, bs, 4, g)
    x_T = torch.randn(2 * bs, 4, H // 8, W // 8, generator=g)
    nz = torch.randn(steps, 2 * bs, 4, H // 8, W // 8, generator=g)
    full, pooled = pipe.conditioner.encode_text(prompt, bs, "cpu")
    fn = lambda xc, tt: unet_ref.unet_forward(unet_sd, arch, xc, tt, full, pooled, image_emb)  # noqa: E731
    if sampler == "p_sampler":
        lat = diffusion_ref.RefDiffusion(steps).p_sample_loop(fn, x_T, nz, guidance)
    elif sampler == "ddim_sampler":
        lat = diffusion_ref.ddim_sample_loop(fn, x_T, steps, guidance)
    else:
        lat = diffusion_ref.plms_sample_loop(fn, x_T, steps, guidance)
    lat = lat[:bs]
    with torch.no_grad():
        want = movq_ref.process_images_u8(movq_ref.movq_decode(movq_sd, marc, lat / 1)[:, :, :H, :W])
    got = pipe.generate_text2img(prompt, num_steps=steps, batch_size=bs, guidance_scale=guidance, h=H, w=W, sampler=sampler,
                                 prior_cf_scale=4, prior_steps=str(PRIOR_STEPS), noise=x_T.cuda(), noise_seq=nz.cuda(),
                                 prior_noise=p_xT.cuda(), prior_noise_seq=p_nz.cuda(), output_type="uint8")
    e_lat = (pipe.last_latent.cpu() - lat).abs().max().item()
    scale = lat.abs().max().item()
    d = np.abs(got.astype(np.int32) - want.numpy().astype(np.int32))
    print(f"{sampler}: final latent max|d| {e_lat:.3e} (scale {scale:.2f}); uint8 image max diff {d.max()}, {100.0 * (d > 0).mean():.3f} % of bytes differ")
    assert got.shape == (bs, H, W, 3) and got.dtype == np.uint8
    assert e_lat <= 1e-3 * max(1.0, scale)
    if sampler == "p_sampler":
        assert d.max() <= 1
    else:
        # DDIM / PLMS do not threshold: the latent of a random-weight UNet grows to ~50-150, far outside what MoVQ decodes
        # linearly, and its 1e-6-relative error moves a few saturated pixels by tens of grey levels: bound the fraction instead
        assert (d > 1).mean() <= 0.01
    # PIL output, as the reference returns it
    pil = pipe.generate_text2img(prompt, num_steps=steps, batch_size=1, guidance_scale=guidance, h=H, w=W, sampler=sampler,
                                 prior_steps=str(PRIOR_STEPS))
    assert len(pil) == 1 and pil[0].size == (W, H)


def test_generate_inpainting_matches_the_oracle_chain_fp32():
    """encode -> prepare_mask -> masked-latent UNet + blend in the sampler step -> decode (kandinsky2_1_model.py:485-548)."""
    from kandinsky2_amd.movq import movq_encoder_blocks
    pipe = _pipe("inpainting")
    cfg, arch, unet_sd, hp, prior_sd, marc, movq_sd, cm, cs = _weights("inpainting")
    bs, steps, guidance, prompt = 1, 5, 4.0, "a hat"
    g = torch.Generator().manual_seed(6)
    img = (torch.randn(1, 3, H, W, generator=g) * 0.5).clamp(-1, 1)
    mask_px = np.ones((H, W), dtype=np.float32)
    mask_px[32:80, 40:100] = 0.0
    x_T = torch.randn(2, 4, H // 8, W // 8, generator=g)
    nz = torch.randn(steps, 2, 4, H // 8, W // 8, generator=g)
    got = pipe.generate_inpainting(prompt, img, mask_px, num_steps=steps, batch_size=bs, guidance_scale=guidance, h=H, w=W,
                                   sampler="p_sampler", prior_steps=str(PRIOR_STEPS), noise=x_T.cuda(), noise_seq=nz.cuda(), output_type="uint8")
    # oracle chain; the image embedding comes from the pipeline's own prior call (un-injected noise): read it back
    image_emb = pipe._last_image_emb.cpu()
    blocks, last = movq_encoder_blocks(marc)
    with torch.no_grad():
        lat0 = movq_ref.movq_encode(movq_sd, blocks, last, img) * 1
    m = torch.nn.functional.interpolate(torch.from_numpy(mask_px)[None, None], lat0.shape[-2:], mode="nearest")
    m = prestep_ref.prepare_mask(m)
    init, mm = lat0.repeat(2, 1, 1, 1), m.repeat(2, 1, 1, 1)
    full, pooled = pipe.conditioner.encode_text(prompt, bs, "cpu")
    fn = lambda xc, tt: unet_ref.unet_forward(unet_sd, arch, xc, tt, full, pooled, image_emb, init * mm, mm)  # noqa: E731
    lat = diffusion_ref.RefDiffusion(steps).p_sample_loop(fn, x_T, nz, guidance, init, mm)[:bs]
    with torch.no_grad():
        want = movq_ref.process_images_u8(movq_ref.movq_decode(movq_sd, marc, lat)[:, :, :H, :W])
    e_lat = (pipe.last_latent.cpu() - lat).abs().max().item()
    d = np.abs(got.astype(np.int32) - want.numpy().astype(np.int32))
    print(f"inpainting: final latent max|d| {e_lat:.3e}; uint8 image max diff {d.max()}, {100.0 * (d > 0).mean():.3f} % of bytes differ")
    assert e_lat <= 1e-3 and d.max() <= 1


def test_reference_style_call_of_p_sample_loop_equals_the_fused_call():
    """Kandinsky2_1.generate_img's own closures (model_fn with the guidance in PyTorch ops, denoised_fun) passed to
    p_sample_loop with the reference's keyword set give the fused call's result (guidance folded into the sampler kernel)."""
    pipe = _pipe("inpainting")
    model = pipe.model
    bs, steps, guidance_scale = 2, 4, 4.0
    arch = model.arch
    g = torch.Generator().manual_seed(8)
    full, pooled, image = k22.make_conditioning(arch, 2 * bs, seed=2)
    x_T = torch.randn(2 * bs, 4, 16, 16, generator=g).cuda()
    nz = torch.randn(steps, 2 * bs, 4, 16, 16, generator=g).cuda()
    init_img = torch.randn(2, 4, 16, 16, generator=g).cuda().repeat(bs, 1, 1, 1)
    img_mask = (torch.rand(2, 1, 16, 16, generator=g) > 0.4).float().cuda().repeat(bs, 1, 1, 1)
    model_kwargs = dict(full_emb=full.cuda(), pooled_emb=pooled.cuda(), image_emb=image.cuda(), inpaint_image=init_img * img_mask, inpaint_mask=img_mask)
    diffusion = k22.create_gaussian_diffusion(**dict(k22.DIFFUSION_CONFIG_2_1, timestep_respacing=str(steps)))

    def model_fn(x_t, ts, **kwargs):           # kandinsky2_1_model.py:222-233, sampler == "p_sampler"
        half = x_t[: len(x_t) // 2]
        combined = torch.cat([half, half], dim=0)
        model_out = model(combined, ts, **kwargs)
        eps, rest = model_out[:, :4], model_out[:, 4:]
        cond_eps, uncond_eps = torch.split(eps, len(eps) // 2, dim=0)
        half_eps = uncond_eps + guidance_scale * (cond_eps - uncond_eps)
        eps = torch.cat([half_eps, half_eps], dim=0)
        return torch.cat([eps, rest], dim=1)

    def denoised_fun(x_start):                 # kandinsky2_1_model.py:237-240
        x_start = x_start.clamp(-2, 2)
        return x_start * (1 - img_mask) + init_img * img_mask

    model.del_cache()
    ref_style = diffusion.p_sample_loop(model_fn, (2 * bs, 4, 16, 16), device="cuda", noise=x_T, progress=False, model_kwargs=model_kwargs,
                                        init_step=None, denoised_fn=denoised_fun, noise_seq=nz)
    model.del_cache()
    fused = diffusion.p_sample_loop(model, (2 * bs, 4, 16, 16), noise=x_T, model_kwargs=model_kwargs, guidance_scale=guidance_scale,
                                    init_img=init_img, img_mask=img_mask, noise_seq=nz)
    err = (ref_style - fused).abs().max().item()
    print(f"reference-style call vs fused call: max|d| = {err:.3e}")
    assert err <= 1e-5
    with pytest.raises(NotImplementedError):
        diffusion.p_sample_loop(model_fn, (2 * bs, 4, 16, 16), device="cuda", noise=x_T, model_kwargs=model_kwargs, denoised_fn=lambda x: x.tanh())


def test_forward_rejects_mismatched_batches():
    """shapes are checked before anything crosses the C ABI (the engine copies B * ... bytes from each pointer)."""
    pipe = _pipe("inpainting")
    model, arch = pipe.model, pipe.model.arch
    full, pooled, image = k22.make_conditioning(arch, 4, seed=2)
    x = torch.randn(4, 4, 16, 16).cuda()
    t = torch.full((4,), 10.0).cuda()
    kw = dict(full_emb=full.cuda(), pooled_emb=pooled.cuda(), image_emb=image.cuda())
    model.del_cache()
    with pytest.raises(ValueError):
        model(x, t[:2], **kw)
    model.del_cache()      # (a cached conditioning is NOT re-read, like the reference: text2im_model2_1.py:58-59)
    with pytest.raises(ValueError):
        model(x, t, full_emb=full[:2].cuda(), pooled_emb=pooled.cuda(), image_emb=image.cuda())
    with pytest.raises(ValueError):
        model(x, t, inpaint_image=torch.zeros(3, 4, 16, 16).cuda(), inpaint_mask=torch.zeros(4, 1, 16, 16).cuda(), **kw)
    # a [1,1,h,w] mask broadcasts like the reference's expression would
    out = model(x, t, inpaint_image=torch.zeros(1, 4, 16, 16).cuda(), inpaint_mask=torch.ones(1, 1, 16, 16).cuda(), **kw)
    assert out.shape == (4, 8, 16, 16)
    d = k22.create_gaussian_diffusion(**dict(k22.DIFFUSION_CONFIG_2_1, timestep_respacing="3"))
    with pytest.raises(ValueError):
        d.p_sample_loop(model, (4, 4, 16, 16), model_kwargs=kw, guidance_scale=4.0, init_img=torch.zeros(3, 4, 16, 16).cuda(),
                        img_mask=torch.zeros(3, 1, 16, 16).cuda())
    # a failed plan (batch > 8) must not poison the engine for the shape that worked
    with pytest.raises(RuntimeError):
        model(torch.randn(10, 4, 16, 16).cuda(), torch.zeros(10).cuda(), **kw)
    model.del_cache()
    assert model(x, t, **kw).shape == (4, 8, 16, 16)


def test_generate_text2img_bf16_runs_and_is_close():
    """Product dtype through the whole chain: finite, and the decoded image stays near the fp32 engines' image."""
    p32, pbf = _pipe("text2img", torch.float32), _pipe("text2img", torch.bfloat16)
    g = torch.Generator().manual_seed(9)
    x_T = torch.randn(2, 4, H // 8, W // 8, generator=g).cuda()
    nz = torch.randn(6, 2, 4, H // 8, W // 8, generator=g).cuda()
    pn, pz = torch.randn(2, 768, generator=g).cuda(), torch.randn(PRIOR_STEPS, 2, 768, generator=g).cuda()
    kw = dict(num_steps=6, batch_size=1, guidance_scale=4.0, h=H, w=W, sampler="p_sampler", prior_steps=str(PRIOR_STEPS), noise=x_T,
              noise_seq=nz, prior_noise=pn, prior_noise_seq=pz, output_type="uint8")
    a, b = p32.generate_text2img("green tree", **kw), pbf.generate_text2img("green tree", **kw)
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    lat = (p32.last_latent - pbf.last_latent).abs().max().item()
    print(f"bf16 chain vs fp32 chain: latent max|d| {lat:.3e}; uint8 mean |d| {d.mean():.2f}, max {d.max()}")
    assert np.isfinite(lat) and d.mean() < 12.0


def test_generate_img2img_and_mix_images_compose_the_same_pieces():
    """generate_img2img = MOVQ.encode * scale -> q_sample at the loop's start step -> loop from init_step (kandinsky2_1_model.py:428-483);
    mix_images = weighted sum of prior / image embeddings -> generate_img (:353-426).  Both must equal the same pieces called by hand
    with the same injected noise (bit for bit: same engines, same order), and produce an image of the requested size."""
    pipe = _pipe("text2img")
    steps, bs, prompt = 6, 1, "a blue bird"
    g = torch.Generator().manual_seed(11)
    img = (torch.randn(1, 3, H, W, generator=g) * 0.5).clamp(-1, 1).cuda()
    qn = torch.randn(1, 4, H // 8, W // 8, generator=g).cuda()
    nz = torch.randn(steps, 2, 4, H // 8, W // 8, generator=g).cuda()
    out = pipe.generate_img2img(prompt, img, strength=0.5, num_steps=steps, batch_size=bs, guidance_scale=4.0, h=H, w=W, sampler="p_sampler",
                                prior_steps=str(PRIOR_STEPS), q_noise=qn, noise_seq=nz, output_type="uint8")
    lat1, emb = pipe.last_latent.clone(), pipe._last_image_emb.clone()
    # by hand
    cfg, diffusion = pipe._diffusion("p_sampler", steps)
    start = int(diffusion.num_timesteps * (1 - 0.5))
    lat0 = pipe.image_encoder.encode(img) * pipe.scale
    x0 = k22.prestep.q_sample(lat0, torch.tensor(diffusion.timestep_map[start - 1]), schedule_name="linear", num_steps=1000, noise=qn)
    out2 = pipe.generate_img(prompt=prompt, img_prompt=emb, batch_size=bs, guidance_scale=4.0, h=H, w=W, sampler="p_sampler", num_steps=steps,
                             diffusion=diffusion, noise=x0.repeat(2, 1, 1, 1), init_step=start, noise_seq=nz, output_type="uint8")
    assert out.shape == (bs, H, W, 3) and torch.equal(lat1, pipe.last_latent) and np.array_equal(out, out2)
    # fewer steps were run than a full loop: the result differs from a text2img run with the same noise
    full = pipe.generate_img(prompt=prompt, img_prompt=emb, batch_size=bs, guidance_scale=4.0, h=H, w=W, sampler="p_sampler", num_steps=steps,
                             diffusion=diffusion, noise=x0.repeat(2, 1, 1, 1), noise_seq=nz, output_type="uint8")
    assert not np.array_equal(full, out)

    # mix_images: 0.3 * prior("a") + 0.7 * clip_image(img)
    torch.manual_seed(5)
    mixed = pipe.mix_images(["a castle", img], [0.3, 0.7], num_steps=4, batch_size=2, guidance_scale=4.0, h=H, w=W, sampler="ddim_sampler",
                            prior_steps=str(PRIOR_STEPS), output_type="uint8")
    emb_used = pipe._last_image_emb
    assert mixed.shape == (2, H, W, 3) and emb_used.shape == (4, 768)
    want_img_part = 0.7 * pipe.encode_images(img)
    # rows 0-1 = the mixed embedding repeated, rows 2-3 = the zero-image embedding (kandinsky2_1_model.py:391-403)
    assert torch.equal(emb_used[0], emb_used[1]) and torch.equal(emb_used[2], emb_used[3])
    assert torch.allclose(emb_used[2], pipe.create_zero_img_emb(1)[0])
    assert (emb_used[0] - want_img_part[0]).abs().max().item() > 0      # the prior's part is in there too


def test_generate_text2img_f16x2_chain_with_fp16_aux_engines_stays_at_the_fp32_image():
    """The gate-holding engine end to end (round 6: beside "f16x2" the prior, the towers and the MoVQ decode run in fp16 - pipeline.py:
    aux_engine_dtypes): prompt -> image with injected noise against the all-fp32 chain.  Bounds = 3x what MI355X measured; the same chain with
    fp32 aux engines (movq_dtype / the old rule) is printed beside it."""
    p32, px2 = _pipe("text2img", torch.float32), _pipe("text2img", k22.F16X2)
    assert px2.movq_dtype == torch.float16 and px2.prior.backend_dtype == torch.float16
    g = torch.Generator().manual_seed(9)
    x_T = torch.randn(2, 4, H // 8, W // 8, generator=g).cuda()
    nz = torch.randn(6, 2, 4, H // 8, W // 8, generator=g).cuda()
    pn, pz = torch.randn(2, 768, generator=g).cuda(), torch.randn(PRIOR_STEPS, 2, 768, generator=g).cuda()
    kw = dict(num_steps=6, batch_size=1, guidance_scale=4.0, h=H, w=W, sampler="p_sampler", prior_steps=str(PRIOR_STEPS), noise=x_T,
              noise_seq=nz, prior_noise=pn, prior_noise_seq=pz, output_type="uint8")
    a, b = p32.generate_text2img("green tree", **kw), px2.generate_text2img("green tree", **kw)
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    lat = (p32.last_latent - px2.last_latent).abs()
    emb = (p32._last_image_emb - px2._last_image_emb).abs().max().item() / p32._last_image_emb.abs().max().item()
    print(f"f16x2 chain (fp16 prior / MoVQ) vs fp32 chain: image_emb {emb:.3e} of scale; latent max|d| {lat.max().item():.3e} rms {lat.pow(2).mean().sqrt().item():.3e}; "
          f"uint8 mean |d| {d.mean():.3f}, max {d.max()}, {100 * (d > 1).mean():.3f} % off by more than one level")
    assert np.isfinite(lat.max().item()) and emb <= 5e-3 and d.mean() < 1.0


def test_generate_text2img_many_pipelines_prompts_and_equals_the_sequential_calls():
    """generate_text2img_many: prior of prompt i + 1 | denoise loop of prompt i | MoVQ decode of prompt i - 1 on three streams.  Same engines, same
    operands, same order inside each engine: every image and final latent equals the sequential generate_text2img call bit for bit - also on a
    second pass (graphs replayed) and for the 16-bit engines."""
    for backend in (torch.float32, torch.bfloat16):
        pipe = _pipe("text2img", backend)
        prompts = ["green tree", "a red cat", "blue bird on a wire", "green tree"]
        g = torch.Generator().manual_seed(21)
        mk = lambda *sh: [torch.randn(*sh, generator=g).cuda() for _ in prompts]   # noqa: E731
        x_T, nz = mk(2, 4, H // 8, W // 8), mk(6, 2, 4, H // 8, W // 8)
        pn, pz = mk(2, 768), mk(PRIOR_STEPS, 2, 768)
        kw = dict(num_steps=6, batch_size=1, guidance_scale=4.0, h=H, w=W, sampler="p_sampler", prior_steps=str(PRIOR_STEPS))
        seq, lats = [], []
        for i, p in enumerate(prompts):
            seq.append(pipe.generate_text2img(p, noise=x_T[i], noise_seq=nz[i], prior_noise=pn[i], prior_noise_seq=pz[i], output_type="tensor", **kw))
            lats.append(pipe.last_latent.clone())
        for _ in range(2):
            many = pipe.generate_text2img_many(prompts, noises=x_T, noise_seqs=nz, prior_noises=pn, prior_noise_seqs=pz, output_type="tensor", **kw)
            torch.cuda.synchronize()
            assert len(many) == len(prompts)
            for i in range(len(prompts)):
                assert torch.equal(many[i], seq[i]), (backend, i, (many[i].int() - seq[i].int()).abs().max().item())
        assert not torch.equal(many[0], many[1])          # different prompts / noise: different images
        del pipe


def test_generate_text2img_many_with_a_grouped_prior_equals_the_sequential_calls_to_rounding():
    """prior_group = 4: ONE prior call samples the embeddings of four prompts (batch 8).  Rows are independent, tile shapes are not the
    same as at batch 2: the image embeddings / final latents equal the per-prompt calls to fp32 rounding (fp32 engines), images within a level."""
    pipe = _pipe("text2img", torch.float32)
    prompts = ["green tree", "a red cat", "blue bird on a wire", "a house"]
    g = torch.Generator().manual_seed(23)
    mk = lambda *sh: [torch.randn(*sh, generator=g).cuda() for _ in prompts]   # noqa: E731
    x_T, nz = mk(2, 4, H // 8, W // 8), mk(6, 2, 4, H // 8, W // 8)
    pn, pz = mk(2, 768), mk(PRIOR_STEPS, 2, 768)
    kw = dict(num_steps=6, batch_size=1, guidance_scale=4.0, h=H, w=W, sampler="p_sampler", prior_steps=str(PRIOR_STEPS))
    seq, lats = [], []
    for i, p in enumerate(prompts):
        seq.append(pipe.generate_text2img(p, noise=x_T[i], noise_seq=nz[i], prior_noise=pn[i], prior_noise_seq=pz[i], output_type="tensor", **kw))
        lats.append(pipe.last_latent.clone())
    many = pipe.generate_text2img_many(prompts, noises=x_T, noise_seqs=nz, prior_noises=pn, prior_noise_seqs=pz, output_type="tensor", prior_group=4, **kw)
    torch.cuda.synchronize()
    worst = 0
    for i in range(len(prompts)):
        d = (many[i].int() - seq[i].int()).abs()
        worst = max(worst, int(d.max().item()))
        assert d.float().mean().item() < 0.05, (i, d.float().mean().item())
    print(f"grouped prior vs per-prompt calls: uint8 max |d| {worst}")
    assert worst <= 2
