"""CPU suite (-m "not gpu"): oracle vs golden fixtures made from the reference, host-side logic of the
product (architecture walk, schedule tables, percentile index), and the C-ABI library surface."""
import json
import os
import re

import numpy as np
import pytest
import torch

import kandinsky2_amd as k22
from kandinsky2_amd import _lib
from oracle import diffusion_ref, unet_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name + ".pt"), weights_only=False)


@pytest.mark.parametrize("name", ["tiny_text2img", "tiny_inpaint"])
def test_oracle_forward_matches_reference_golden(golden_dir, name):
    fx = _load(golden_dir, name)
    arch = k22.make_arch(fx["model_config"], inpainting=fx["inpainting"])
    sd = k22.init_unet_state_dict(arch, seed=fx["seed_w"])
    full, pooled, image = k22.make_conditioning(arch, fx["B"], seed=2)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(fx["B"], 4, fx["h"], fx["w"], generator=g)
    img = torch.randn(fx["B"], 4, fx["h"], fx["w"], generator=g)
    mask = (torch.rand(fx["B"], 1, fx["h"], fx["w"], generator=g) > 0.5).float()
    ii, mm = (img * mask, mask) if fx["inpainting"] else (None, None)
    out = unet_ref.unet_forward(sd, arch, x, fx["t"], full, pooled, image, ii, mm)
    assert out.abs().max() > 0.1  # not the vacuous all-zero output of zero_module()
    assert (out - fx["forward_out"]).abs().max().item() <= 1e-5


def test_oracle_sampler_loop_matches_reference_golden(golden_dir):
    fx = _load(golden_dir, "tiny_text2img")
    arch = k22.make_arch(fx["model_config"])
    sd = k22.init_unet_state_dict(arch, seed=fx["seed_w"])
    full, pooled, image = k22.make_conditioning(arch, fx["B"], seed=2)
    g = torch.Generator().manual_seed(42)
    x_T = torch.randn(fx["B"], 4, fx["h"], fx["w"], generator=g)
    noise_seq = torch.randn(fx["steps"], fx["B"], 4, fx["h"], fx["w"], generator=g)
    od = diffusion_ref.RefDiffusion(fx["steps"])
    final = od.p_sample_loop(lambda xc, tt: unet_ref.unet_forward(sd, arch, xc, tt, full, pooled, image), x_T, noise_seq, fx["guidance"])
    assert (final - fx["final"]).abs().max().item() <= 1e-4


def test_param_shapes_equal_reference_state_dict(golden_dir):
    with open(os.path.join(golden_dir, "ref_unet_keys.json")) as f:
        keys = json.load(f)
    for nm, inp in (("text2img", False), ("inpainting", True)):
        mine = {k: list(v) for k, v in k22.param_shapes(k22.make_arch(k22.MODEL_CONFIG_2_1, inpainting=inp)).items()}
        assert mine == keys[nm]


@pytest.mark.parametrize("steps", [10, 50, 100])
def test_schedule_tables_equal_reference(golden_dir, steps):
    with open(os.path.join(golden_dir, "ref_diffusion_tables.json")) as f:
        ref = json.load(f)[str(steps)]
    d = k22.create_gaussian_diffusion(**dict(k22.DIFFUSION_CONFIG_2_1, timestep_respacing=str(steps)))
    o = diffusion_ref.RefDiffusion(steps)
    assert d.timestep_map == ref["timestep_map"] == o.timestep_map
    for mine, theirs, key in [
        (d.betas, o.betas, "betas"),
        (d.sqrt_recip_alphas_cumprod, o.sqrt_recip, "sqrt_recip_alphas_cumprod"),
        (d.sqrt_recipm1_alphas_cumprod, o.sqrt_recipm1, "sqrt_recipm1_alphas_cumprod"),
        (d.posterior_log_variance_clipped, o.post_logvar, "posterior_log_variance_clipped"),
        (d.posterior_mean_coef1, o.c1, "posterior_mean_coef1"),
        (d.posterior_mean_coef2, o.c2, "posterior_mean_coef2"),
    ]:
        assert np.array_equal(mine, np.array(ref[key])), key   # bit-exact fp64
        assert np.array_equal(theirs, np.array(ref[key])), key
    tab = d.step_table()
    assert tab.shape == (steps, 8) and tab[0, 6] == 0 and (tab[1:, 6] == 1).all()
    assert np.array_equal(tab[:, 7], np.array([o.model_t(i) for i in range(steps)], dtype=np.float32))


def test_space_timesteps_ddim_and_sections():
    assert k22.space_timesteps(1000, "10") == [0, 111, 222, 333, 444, 555, 666, 777, 888, 999]
    assert k22.space_timesteps(300, [10, 15, 20])[:3] == [0, 11, 22]
    assert k22.space_timesteps(1000, "ddim50")[:3] == [1, 21, 41]
    with pytest.raises(ValueError):
        k22.space_timesteps(10, "20")


@pytest.mark.parametrize("n", [4 * 8 * 8, 4 * 16 * 16, 4 * 32 * 32, 4 * 96 * 96, 4 * 128 * 128, 7, 1])
def test_percentile_index_reproduces_numpy(n):
    rng = np.random.default_rng(n)
    a = np.abs(rng.standard_normal(n).astype(np.float32))
    lo, gamma = k22.percentile_index(n)
    s = np.sort(a)
    hi = min(lo + 1, n - 1)
    d = np.float32(s[hi] - s[lo])
    t = np.float32(gamma)
    r = np.float32(s[lo] + np.float32(d * t))
    if t >= 0.5:
        r = np.float32(s[hi] - np.float32(d * np.float32(np.float32(1) - t)))
    assert r == np.percentile(a, 99.5)


def test_c_abi_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "k22.h")).read()
    declared = set(re.findall(r"\b(k22_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("k22_sampler_table_columns")
    assert declared, "no declarations parsed"
    assert os.path.exists(_lib.LIB_PATH), "libk22hip.so not built: run __graft_entry__.build()"
    L = _lib.lib()
    for sym in sorted(declared):
        assert hasattr(L, sym), sym
        assert sym in _lib.SIGNATURES, f"{sym} has no ctypes signature"
    assert L.k22_version() >= 100


def test_product_fails_loudly_without_gpu():
    arch = k22.make_arch(k22.tiny_model_config())
    m = k22.Text2ImUNetHIP(arch)
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 4, 8, 8), torch.zeros(2), full_emb=torch.zeros(2, 77, 1024), pooled_emb=torch.zeros(2, 768),
          image_emb=torch.zeros(2, 768))


def test_pack_layouts():
    from kandinsky2_amd.pack import pack_arena, packed_entries
    arch = k22.make_arch(k22.tiny_model_config())
    sd = k22.init_unet_state_dict(arch, seed=1)
    ent = packed_entries(arch, sd, torch.float32, "cpu")
    w = sd["input_blocks.1.0.in_layers.2.weight"]
    p = ent["input_blocks.1.0.in_layers.2.weight"]
    assert p.shape == (128, 9 * 128)
    assert torch.equal(p[5].view(3, 3, 128)[1, 2], w[5, :, 1, 2])
    # qkv rows: head*192 + part*64 + d  ->  part*C + head*64 + d
    pfx = "input_blocks.5.1"
    wq, pq = sd[pfx + ".qkv.weight"][:, :, 0], ent[pfx + ".qkv.weight"]
    C = wq.shape[1]
    assert torch.equal(pq[1 * C + 2 * 64 + 7], wq[2 * 192 + 1 * 64 + 7])
    tot = sum(2 * b[3] for b in arch.blocks if b[0] == "res")
    assert ent["emb_layers.weight"].shape == (tot, arch.time_embed_dim)
    arena, table = pack_arena(arch, sd, torch.bfloat16, "cpu")
    meta = {k: torch.empty(v, device="meta") for k, v in k22.param_shapes(arch).items()}
    _, table2 = pack_arena(arch, meta, torch.bfloat16, "meta")
    assert table == table2 and all(o % 256 == 0 for o, _ in table.values())


# ---- MoVQ decoder ----------------------------------------------------------------
def test_movq_oracle_matches_reference_golden(golden_dir):
    from oracle import movq_ref
    fx = _load(golden_dir, "movq_small")
    arch = k22.MoVQArch(k22.MOVQ_CONFIG_2_1["ddconfig"])
    sd = k22.init_movq_state_dict(arch, seed=fx["seed_w"])
    g = torch.Generator().manual_seed(fx["seed_z"])
    z = torch.randn(fx["B"], 4, fx["h"], fx["w"], generator=g)
    out = movq_ref.movq_decode(sd, arch, z)
    assert out.abs().max() > 0.5
    assert (out - fx["out"]).abs().max().item() <= 1e-5
    assert torch.equal(movq_ref.process_images_u8(fx["out"]), fx["out_u8"])


def test_movq_encoder_oracle_matches_reference_golden(golden_dir):
    """oracle/movq_ref.movq_encode == the reference MOVQ.encode (golden made by importing it, make_golden.movq_enc_case,
    which also asserts that the encoder.* / quant_conv.* key set equals the reference module's)."""
    from oracle import movq_ref
    from kandinsky2_amd.movq import movq_encoder_blocks
    fx = _load(golden_dir, "movq_enc_small")
    arch = k22.MoVQArch(k22.MOVQ_CONFIG_2_1["ddconfig"])
    sd = k22.init_movq_encoder_state_dict(arch, seed=fx["seed_w"])
    x = torch.randn(fx["B"], 3, fx["H"], fx["W"], generator=torch.Generator().manual_seed(fx["seed_x"])).clamp(-2, 2) * 0.5
    blocks, last = movq_encoder_blocks(arch)
    with torch.no_grad():
        out = movq_ref.movq_encode(sd, blocks, last, x)
    assert torch.equal(out, fx["out"])


def test_prestep_oracle_matches_reference_golden(golden_dir):
    """prepare_mask (gather form) and q_sample of oracle/prestep_ref.py == kandinsky2/utils.py:11-54 (goldens made by calling it)."""
    from oracle import prestep_ref
    fx = _load(golden_dir, "prestep")
    for m, want in zip(fx["masks"], fx["mask_out"]):
        assert torch.equal(prestep_ref.prepare_mask(m), want)
    for t, want in fx["q"].items():
        assert torch.equal(prestep_ref.q_sample(fx["x"], t, noise=fx["noise"]), want)


def test_movq_state_dict_keys_match_reference(golden_dir):
    with open(os.path.join(golden_dir, "ref_movq_keys.json")) as f:
        ref = json.load(f)
    arch = k22.MoVQArch(k22.MOVQ_CONFIG_2_1["ddconfig"])
    mine = {k: list(v) for k, v in k22.movq_param_shapes(arch).items()}
    assert mine == ref
    assert arch.attn_levels == [3]  # resolution 256 / 2^3 = 32 is the only entry of attn_resolutions


def test_movq_arena_layout_is_shape_determined():
    from kandinsky2_amd.movq import pack_movq_arena
    arch = k22.MoVQArch(k22.MOVQ_CONFIG_2_1["ddconfig"])
    sd = {k: torch.zeros(v) for k, v in k22.movq_param_shapes(arch).items()}
    arena, table = pack_movq_arena(arch, sd, torch.bfloat16, "cpu")
    assert table["decoder.conv_in.weight"][1] == 512 * 9 * 64 * 2          # Cin 4 zero-extended to one 64-channel K slab
    assert table["decoder.conv_out.weight"][1] == 64 * 9 * 128 * 2         # 3 output rows padded to 64
    assert table["decoder.mid.block_1.norm1.conv_y.weight"][1] == 512 * 4 * 4  # fp32, evaluated inside the norm kernel
    assert all(off % 256 == 0 for off, _ in table.values()) and arena.dtype == torch.uint8


# ---- diffusion prior ---------------------------------------------------------------
@pytest.mark.parametrize("name", ["prior_tiny", "prior_full"])        # prior_full: the production 2048 x 20 transformer (1.0 B parameters)
def test_prior_oracle_matches_reference_golden(golden_dir, name):
    from oracle import prior_ref
    fx = _load(golden_dir, name)
    hp = fx["hp"]
    sd = k22.init_prior_state_dict(hp, seed=fx["seed_w"])
    g = torch.Generator().manual_seed(7)
    bs = fx["bs"]; N = 2 * bs
    cm, cs = torch.randn(768, generator=g) * 0.1, torch.rand(768, generator=g) + 0.5
    txt_feat, txt_seq = torch.randn(N, 768, generator=g), torch.randn(N, 77, 768, generator=g)
    mask = torch.zeros(N, 77, dtype=torch.bool)
    for r in range(bs):
        mask[r, : 9 + 11 * r] = True
    mask[bs:, :2] = True
    x = torch.randn(N, 768, generator=g)
    out = prior_ref.transformer_forward(sd, hp, x, fx["t"], txt_feat, txt_seq, mask)
    assert (out - fx["forward_out"]).abs().max().item() <= 1e-5
    x_T, noise_seq = torch.randn(N, 768, generator=g), torch.randn(fx["steps"], N, 768, generator=g)
    smp = prior_ref.prior_sample(sd, hp, txt_feat, txt_seq, mask, fx["scales"], fx["steps"], x_T, noise_seq, cm[None], cs[None])
    assert (smp - fx["sample"]).abs().max().item() <= 1e-4


def test_prior_state_dict_keys_match_reference(golden_dir):
    with open(os.path.join(golden_dir, "ref_prior_keys.json")) as f:
        ref = json.load(f)
    mine = {k: list(v) for k, v in k22.prior_param_shapes(k22.PRIOR_HPARAMS_2_1).items()}
    assert mine == ref


def test_prior_schedule_matches_oracle_tables():
    from oracle import prior_ref
    for n in (5, 25, 100):
        a, b = k22.PriorSchedule(timestep_respacing=str(n)), prior_ref.RefPriorSchedule(n)
        assert a.timestep_map == b.timestep_map and a.num_timesteps == b.T
        tab = a.step_table()
        assert np.array_equal(tab[:, 0], b.c1.astype(np.float32)) and np.array_equal(tab[:, 2], b.logvar.astype(np.float32))
        assert tab[0, 3] == 0 and (tab[1:, 3] == 1).all()


def test_ddim_oracle_matches_reference_golden(golden_dir):
    fx = _load(golden_dir, "tiny_ddim")
    arch = k22.make_arch(fx["model_config"])
    sd = k22.init_unet_state_dict(arch, seed=fx["seed_w"])
    full, pooled, image = k22.make_conditioning(arch, fx["B"], seed=2)
    g = torch.Generator().manual_seed(42)
    x_T = torch.randn(fx["B"], 4, fx["h"], fx["w"], generator=g)
    out = diffusion_ref.ddim_sample_loop(lambda xc, tt: unet_ref.unet_forward(sd, arch, xc, tt, full, pooled, image), x_T, fx["steps"], fx["guidance"])
    assert (out - fx["final"]).abs().max().item() <= 1e-4 * fx["final"].abs().max().item()


def test_plms_oracle_matches_reference_golden(golden_dir):
    """oracle/diffusion_ref.plms_sample_loop == the reference PLMSSampler (golden made by importing it, make_golden.plms_case)."""
    fx = _load(golden_dir, "tiny_plms")
    arch = k22.make_arch(fx["model_config"])
    sd = k22.init_unet_state_dict(arch, seed=fx["seed_w"])
    full, pooled, image = k22.make_conditioning(arch, fx["B"], seed=2)
    x_T = torch.randn(fx["B"], 4, fx["h"], fx["w"], generator=torch.Generator().manual_seed(43))
    out = diffusion_ref.plms_sample_loop(lambda xc, tt: unet_ref.unet_forward(sd, arch, xc, tt, full, pooled, image), x_T, fx["steps"], fx["guidance"])
    assert torch.equal(out, fx["final"])


def test_ddim_host_loop_and_schedule_on_cpu(golden_dir):
    """DDIMSamplerHIP's host side (uniform timesteps, table rows, step order) with the fused device step replaced by the same
    tensor expressions on the CPU and the CPU oracle as the UNet: lands on the reference DDIMSampler's golden latent."""
    fx = _load(golden_dir, "tiny_ddim")
    arch = k22.make_arch(fx["model_config"])
    sd = k22.init_unet_state_dict(arch, seed=fx["seed_w"])
    full, pooled, image = k22.make_conditioning(arch, fx["B"], seed=2)

    class CpuStep(k22.DDIMSamplerHIP):
        def _ddim_step(self, x, model_out, noise, table_row, x_out, x0_out):
            eps = model_out[:, :4]
            c, u = torch.split(eps, len(eps) // 2, dim=0)
            he = u + self.guidance_scale * (c - u)
            e = torch.cat([he, he], 0)
            a_t, a_prev, sigma, s1m = table_row[0], table_row[1], table_row[2], table_row[3]
            pred_x0 = (x - s1m * e) / a_t.sqrt()
            x_out.copy_(a_prev.sqrt() * pred_x0 + (1.0 - a_prev - sigma ** 2).sqrt() * e + sigma * torch.zeros_like(x))
            x0_out.copy_(pred_x0)

    def model(xc, ts, **_kw):
        with torch.no_grad():
            return unet_ref.unet_forward(sd, arch, xc, ts, full, pooled, image)

    old = k22.create_gaussian_diffusion(**k22.DIFFUSION_CONFIG_2_1)
    x_T = torch.randn(fx["B"], 4, fx["h"], fx["w"], generator=torch.Generator().manual_seed(42))
    out, _ = CpuStep(model, old, fx["guidance"]).sample(fx["steps"], fx["B"], (4, fx["h"], fx["w"]), x_T=x_T, device="cpu")
    assert (out - fx["final"]).abs().max().item() <= 1e-4 * fx["final"].abs().max().item()


def test_plms_host_loop_ring_and_schedule_on_cpu(golden_dir):
    """PLMSSamplerHIP's host side (schedule table, two-stage start, eps history ring, order selection) with the fused device
    step replaced by the same tensor expressions on the CPU and the CPU oracle as the UNet: must land exactly on the
    reference PLMSSampler's golden latent."""
    fx = _load(golden_dir, "tiny_plms")
    arch = k22.make_arch(fx["model_config"])
    sd = k22.init_unet_state_dict(arch, seed=fx["seed_w"])
    full, pooled, image = k22.make_conditioning(arch, fx["B"], seed=2)

    class CpuStep(k22.PLMSSamplerHIP):
        def _step(self, x, model_out, hist, order, table_row, x_out, eps_out, x0_out):
            eps = model_out[:, :4]
            c, u = torch.split(eps, len(eps) // 2, dim=0)
            he = u + self.guidance_scale * (c - u)
            e = torch.cat([he, he], 0)
            if order == 0:
                ep = e
            elif order == 4:
                ep = (hist[0] + e) / 2
            elif order == 1:
                ep = (3 * e - hist[0]) / 2
            elif order == 2:
                ep = (23 * e - 16 * hist[0] + 5 * hist[1]) / 12
            else:
                ep = (55 * e - 59 * hist[0] + 37 * hist[1] - 9 * hist[2]) / 24
            a_t, a_prev, s1m = table_row[0], table_row[1], table_row[3]
            pred_x0 = (x - s1m * ep) / a_t.sqrt()
            x_out.copy_(a_prev.sqrt() * pred_x0 + (1.0 - a_prev - torch.tensor(0.0) ** 2).sqrt() * ep)
            if eps_out is not None:
                eps_out.copy_(e)
            if x0_out is not None:
                x0_out.copy_(pred_x0)

    def model(xc, ts, **_kw):
        with torch.no_grad():
            return unet_ref.unet_forward(sd, arch, xc, ts, full, pooled, image)

    old = k22.create_gaussian_diffusion(**k22.DIFFUSION_CONFIG_2_1)
    x_T = torch.randn(fx["B"], 4, fx["h"], fx["w"], generator=torch.Generator().manual_seed(43))
    out, _ = CpuStep(model, old, fx["guidance"]).sample(fx["steps"], fx["B"], (4, fx["h"], fx["w"]), x_T=x_T, device="cpu")
    assert torch.equal(out, fx["final"])


def test_ddim_schedule_matches_oracle():
    old = k22.create_gaussian_diffusion(**k22.DIFFUSION_CONFIG_2_1)
    s = k22.DDIMSamplerHIP(None, old, 4.0)
    s.make_schedule(50)
    assert list(s.ddim_timesteps[:3]) == [1, 21, 41] and len(s.ddim_timesteps) == 50
    ac = np.cumprod(1.0 - diffusion_ref.linear_betas(1000, 0.00085, 0.012))
    assert np.array_equal(s.table[:, 0], ac[s.ddim_timesteps].astype(np.float32))
    assert s.table[0, 1] == np.float32(ac[0]) and (s.table[:, 2] == 0).all()
    s.make_schedule(50, init_step=500)
    assert s.ddim_timesteps.max() <= 500


def test_unet22_key_mapping_covers_every_diffusers_key_once():
    """Kandinsky 2.2 UNet (diffusers keys): the mapping onto the engine's packed layout uses every key exactly once, for the plain,
    the ControlNet-depth and the inpainting variant; the architecture has the published size (1.25 B parameters)."""
    from kandinsky2_amd.unet22 import sd22_to_internal, names22
    from kandinsky2_amd.pack import pack_arena
    for kw in (dict(), dict(controlnet=True), dict(inpainting=True)):
        arch = k22.make_arch22(k22.tiny_unet22_config(), **kw)
        shapes = k22.param_shapes22(arch)
        sd = {k: torch.empty(v, device="meta") for k, v in shapes.items()}
        used = set()

        class Spy(dict):
            def __getitem__(self, k):
                used.add(k)
                return dict.__getitem__(self, k)

        internal = sd22_to_internal(arch, Spy(sd))
        assert used == set(shapes.keys()), sorted(set(shapes.keys()) - used)[:5]
        arena, table = pack_arena(arch, sd, torch.bfloat16, "meta")
        assert "head22.ctx_proj.weight" in table and ("hint.7.weight" in table) == bool(arch.hint_channels)
        assert len(set(names22(arch).values())) == len(names22(arch))
    full = k22.make_arch22()
    n = sum(int(np.prod(v)) for v in k22.param_shapes22(full).values())
    assert 1.24e9 < n < 1.27e9
    assert k22.param_shapes22(k22.make_arch22(controlnet=True))["conv_in.weight"] == (384, 8, 3, 3)
    assert k22.param_shapes22(k22.make_arch22(inpainting=True))["conv_in.weight"] == (384, 9, 3, 3)


def test_unet22_oracle_runs_and_scheduler_tables_agree():
    """oracle/unet22_ref.py (parity unpinned) executes on the tiny configuration; DDPMSchedulerHIP's step table equals the
    oracle scheduler's per-step scalars; with 'leading' spacing the table is improved-DDPM respacing over the same timesteps."""
    from oracle import unet22_ref
    cfg = k22.tiny_unet22_config()
    arch = k22.make_arch22(cfg, controlnet=True)
    sd = k22.init_unet22_state_dict(arch, seed=0)
    g = torch.Generator().manual_seed(1)
    x, emb, hint = torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 1280, generator=g), torch.rand(2, 3, 64, 64, generator=g)
    out = unet22_ref.unet22_forward(sd, cfg, x, torch.tensor([980, 0]), emb, hint)
    assert out.shape == (2, 8, 8, 8) and torch.isfinite(out).all() and out.abs().max() > 1e-3
    for cfg_hip, cfg_ref in ((k22.SCHEDULER_CONFIG_2_2_LEARNED_RANGE, unet22_ref.SCHED_2_2_LEARNED_RANGE), (k22.SCHEDULER_CONFIG_2_2, unet22_ref.SCHED_2_2)):
        sch = k22.DDPMSchedulerHIP.from_config(cfg_hip).set_timesteps(50, device="cpu")
        ref = unet22_ref.RefDDPMScheduler(50, cfg_ref)
        assert sch.timesteps.tolist() == ref.timesteps.tolist()
        tab = sch._table_host
        ac = ref.alphas_cumprod.double().numpy()
        for row, t in enumerate(sch.timesteps.tolist()):
            prev = t - 20
            a_prev = ac[prev] if prev >= 0 else 1.0
            assert abs(tab[row, 0] - (1 / ac[t]) ** 0.5) < 1e-5 * tab[row, 0]
            log_var = np.log(max((1 - a_prev) / (1 - ac[t]) * (1 - ac[t] / a_prev), 1e-20))
            assert abs(tab[row, 4] - log_var) < 1e-4 * max(1.0, abs(log_var))
            # learned_range interpolates between log(posterior variance) and log(beta_t); fixed_small pins both bounds to the former
            want_hi = np.log(1 - ac[t] / a_prev) if ref.learned else log_var
            assert abs(tab[row, 5] - want_hi) < 1e-4 * max(1.0, abs(want_hi))
    # respacing identity: betas' = 1 - abar_t / abar_prev over the retained timesteps
    d = k22.create_gaussian_diffusion(**dict(k22.DIFFUSION_CONFIG_2_1, timestep_respacing="50"))
    assert d.num_timesteps == 50


def test_shipped_tile_table_loads_and_round_trips(tmp_path):
    """kandinsky-2_amd/tiles_gfx950.txt (csrc/tuning.h): the table that fixes every tile configuration - and with it every fp32
    summation order - of the benchmarked shapes.  Host logic only: load, save, reload give the same entries."""
    L = _lib.lib()
    shipped = [l for l in open(_lib.TILE_TABLE_PATH).read().splitlines() if l and not l.startswith("#")]
    assert len(shipped) >= 800 and len(set(tuple(l.split()[:10]) for l in shipped)) == len(shipped), "one line per problem"
    try:
        L.k22_tile_table_clear()
        assert L.k22_tile_table_size() == 0
        assert L.k22_tile_table_load(_lib.TILE_TABLE_PATH.encode()) == len(shipped) and L.k22_tile_table_size() == len(shipped)
        out = str(tmp_path / "tiles.txt")
        assert L.k22_tile_table_save(out.encode()) == len(shipped)
        again = [l for l in open(out).read().splitlines() if l and not l.startswith("#")]
        assert sorted(again) == sorted(shipped)
        assert L.k22_tile_table_load(b"/nonexistent/tiles.txt") < 0
        assert L.k22_tile_table_measured() == 0          # nothing is ever timed without a GPU forward
    finally:
        L.k22_tile_table_clear()
        L.k22_tile_table_load(_lib.TILE_TABLE_PATH.encode())


def test_shipped_tile_table_names_the_streaming_kernel_only_where_an_engine_can_run_it():
    """Table columns: dtype taps M N Kc K0 H W outmode stats | algo bm bn splitk stages | us.  algo 20 = stream_kernel (csrc/stream_gemm.hip): the
    engine lists it as a candidate only for 16-bit 3x3 convolutions of at most 1152 rows whose weights it holds fragment-major (engine.hip op_conv,
    tuning.h), with 160- or 288-row tiles, K a multiple of 64 and at most 256 workgroups' worth of split-K.  A line outside that envelope would be
    ignored at load (tuned_is_candidate) and silently re-measured on every start - i.e. the table would no longer fix the bits."""
    rows = [l.split() for l in open(_lib.TILE_TABLE_PATH).read().splitlines() if l and not l.startswith("#")]
    stream = [r for r in rows if r[10] == "20"]
    assert len(stream) >= 16, "the C2 12x12 level runs the weight-streaming kernel"
    for r in stream:
        dtype, taps, M, N, Kc = int(r[0]), int(r[1]), int(r[2]), int(r[3]), int(r[4])
        bm, bn, splitk = int(r[11]), int(r[12]), int(r[13])
        assert dtype == 0 and taps == 9 and M <= 1152 and Kc % 64 == 0 and N % 4 == 0, r
        assert bm in (160, 288) and bn == 0 and 1 <= splitk <= Kc // 64, r
    # the C2 bench shape: both 1536 -> 1536 convolutions of the 12x12 level (with and without the fused 1x1 skip) are among them
    assert any(r[2] == "288" and r[3] == "1536" and r[4] == "1536" and r[5] == "1536" for r in stream)


def test_fragment_major_weight_layout_formula():
    """k22_stream_repack's documented permutation (include/k22.h, csrc/stream_gemm.hip), restated on the host: element e of lane l of k quarter w of
    item (slab, tap) of n-block nb is W[nb * 32 + l % 32][tap * Kc + slab * 64 + 16 w + 8 (l / 32) + e] - i.e. exactly the 16 bytes that lane feeds
    to v_mfma_f32_32x32x16 (B operand: column = lane % 32, k = 8 (lane / 32) + e within the wave's 16-wide k step).  The GPU test
    (tests/test_stream_gpu.py::test_repack_is_the_documented_permutation) checks the kernel against the same tensor expression."""
    Npad, taps, Kc = 64, 9, 128
    W = torch.arange(Npad * taps * Kc).reshape(Npad, taps * Kc)
    frag = W.reshape(Npad // 32, 32, taps, Kc // 64, 4, 2, 8).permute(0, 3, 2, 4, 5, 1, 6).reshape(-1)     # nb, slab, tap, w, half, row, e
    n_items = (Kc // 64) * taps
    for nb, slab, tap, w, lane, e in [(0, 0, 0, 0, 0, 0), (1, 1, 8, 3, 63, 7), (0, 1, 4, 2, 37, 5), (1, 0, 2, 1, 31, 3)]:
        it = slab * taps + tap
        idx = (((nb * n_items + it) * 4 + w) * 64 + lane) * 8 + e
        assert frag[idx].item() == W[nb * 32 + lane % 32, tap * Kc + slab * 64 + 16 * w + 8 * (lane // 32) + e].item()
    assert _lib.lib().k22_stream_frag_bytes(Npad, taps, Kc, _lib.K22_BF16) == Npad * taps * Kc * 2
    assert _lib.lib().k22_stream_frag_bytes(Npad, taps, Kc, _lib.K22_F32) == 0          # 16-bit kernel


# ---- the oracle at the FULL shapes of the bench (the fixtures the -m gpu parity tests of tests/test_full_size_gpu.py use) ---------
def test_oracle_c2_first_forward_matches_reference_golden(golden_dir):
    """C2 (768x768 bs 1 -> CFG batch 2x4x96x96, 1.23 B parameters): the restatement against the reference create_model(...)'s
    output stored by oracle/make_golden.py --only c2.  One forward (the 50-step trajectory is what the GPU tests walk)."""
    p = os.path.join(golden_dir, "c2_text2img.pt")
    if not os.path.exists(p):
        pytest.skip("c2_text2img.pt not generated")
    fx = torch.load(p, weights_only=False)
    arch = k22.make_arch(k22.MODEL_CONFIG_2_1)
    sd = k22.init_unet_state_dict(arch, seed=fx["seed_w"])
    full, pooled, image = k22.make_conditioning(arch, fx["B"], seed=2)
    g = torch.Generator().manual_seed(42)
    x_T = torch.randn(fx["B"], 4, fx["lat"], fx["lat"], generator=g)
    bs = fx["bs"]
    with torch.no_grad():
        out = unet_ref.unet_forward(sd, arch, torch.cat([x_T[:bs], x_T[:bs]], 0), fx["first_ts"].float(), full, pooled, image)
    scale = fx["first_out"].abs().max().item()
    assert scale > 0.1 and (out - fx["first_out"]).abs().max().item() <= 1e-5 * scale


def test_movq_oracle_matches_reference_golden_256px(golden_dir):
    """MOVQ.decode at 32x32 latents (T = 1024 tokens in the attention blocks): compact fp32 samples + the whole uint8 image."""
    from oracle import movq_ref
    p = os.path.join(golden_dir, "movq_256px.pt")
    if not os.path.exists(p):
        pytest.skip("movq_256px.pt not generated")
    fx = torch.load(p, weights_only=False)
    arch = k22.MoVQArch(k22.MOVQ_CONFIG_2_1["ddconfig"])
    sd = k22.init_movq_state_dict(arch, seed=fx["seed_w"])
    z = torch.randn(fx["B"], 4, fx["h"], fx["w"], generator=torch.Generator().manual_seed(fx["seed_z"]))
    with torch.no_grad():
        out = movq_ref.movq_decode(sd, arch, z)
    c = fx["out_compact"]
    st, r0, c0 = c["stride"], c["r0"], c["c0"]
    tol = 1e-5 * fx["absmax"]
    assert (out[..., ::st, ::st] - c["sub"]).abs().max().item() <= tol
    assert (out[..., r0:r0 + c["rows"].shape[-2], :] - c["rows"]).abs().max().item() <= tol
    assert (out[..., :, c0:c0 + c["cols"].shape[-1]] - c["cols"]).abs().max().item() <= tol
    u8 = movq_ref.process_images_u8(out)
    d = (u8.int() - fx["out_u8"].int()).abs()
    assert d.max().item() <= 1 and (d > 0).float().mean().item() <= 1e-3       # a grey level can flip on an exact .5


# ---- split-precision engine (K22_F16X3, round 4): host side ---------------------------------------------------------------------------
def test_x3_chunk_packer_carries_23_bits_and_keeps_the_shape():
    """pack.to_x3 (what the arena holds for the split-precision engine; csrc/common.h "x3 chunk"): every 8 consecutive K elements ->
    [hi x8 | lo x8] fp16, hi = rne(x * scale), lo = rne(x * scale - hi); 4 bytes per element, same shape."""
    from kandinsky2_amd.pack import to_x3
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.randn(2048, generator=g), torch.randn(2048, generator=g) * 1e-3, torch.randn(2048, generator=g) * 50]).view(48, 128)
    for scale in (1.0, 256.0):
        p = to_x3(x, scale)
        assert p.shape == x.shape and p.dtype == torch.float32
        h = p.view(torch.float16).double().view(48, 16, 2, 8)
        rec = (h[:, :, 0] + h[:, :, 1]).reshape(48, 128) / scale
        err = (rec - x.double()).abs()
        assert (err <= torch.maximum(x.double().abs() * 2.0 ** -22, torch.full_like(err, 2.0 ** -24 / scale))).all()
        # the hi half alone is the fp16 rounding of the (scaled) value: what a plain fp16 engine would see
        assert torch.equal(h[:, :, 0].reshape(48, 128).float(), (x * scale).half().float())
    with pytest.raises(ValueError):
        to_x3(torch.zeros(3, 6))


def test_x3_arena_holds_split_weights_and_fp32_vector_weights():
    """packed_entries(..., "f16x3"): the MFMA weights (conv3x3, qkv, encoder_kv, proj_out, skip, to_model_dim_n) as x3 chunks of the
    fp32 weight x 2^8, emb_layers (the GEMV) in fp32; the arena table is shape-determined like every other engine type's."""
    from kandinsky2_amd.pack import pack_arena, packed_entries, to_x3
    arch = k22.make_arch(k22.tiny_model_config())
    sd = k22.init_unet_state_dict(arch, seed=1)
    ent32 = packed_entries(arch, sd, torch.float32, "cpu")
    ent = packed_entries(arch, sd, k22.F16X3, "cpu")
    assert list(ent) == list(ent32)
    for k in ("input_blocks.1.0.in_layers.2.weight", "input_blocks.5.1.qkv.weight", "input_blocks.5.1.encoder_kv.weight",
              "input_blocks.5.1.proj_out.weight", "to_model_dim_n.weight", "out.2.weight"):
        assert ent[k].shape == ent32[k].shape and torch.equal(ent[k].view(torch.int32), to_x3(ent32[k]).view(torch.int32)), k
    for k in ("emb_layers.weight", "time_embed.0.weight", "input_blocks.1.0.in_layers.0.weight", "input_blocks.1.0.in_layers.2.bias"):
        assert torch.equal(ent[k], ent32[k]), k
    _, table = pack_arena(arch, sd, k22.F16X3, "cpu")
    meta = {k: torch.empty(v, device="meta") for k, v in k22.param_shapes(arch).items()}
    _, table2 = pack_arena(arch, meta, k22.F16X3, "meta")
    assert table == table2
    assert _lib.dtype_code(k22.F16X3) == _lib.K22_F16X3 == 3
    with pytest.raises(ValueError):
        _lib.dtype_code("fp8")


def test_split_precision_emulation_meets_the_gate_on_the_oracle(golden_dir):
    """oracle/drift_ablation.py with the MFMA OPERANDS of the reference UNet rounded to fp16 (hi, lo) pairs (fp32 everywhere else), C2
    shape, 50 steps, against the reference golden (committed result, ~17 CPU-minutes per mode to regenerate): both operands split ->
    2.1e-6; only the weights split (two MFMAs) -> 8.8e-4, too close to the 1e-3 gate to ship; bf16 halves -> 2.9e-5.  The engine mode
    (three fp16 MFMAs) was chosen from this table; tests/test_full_size_gpu.py asserts what the GPU then measures (3.4e-6)."""
    import json
    import os
    p = os.path.join(golden_dir, "drift_ablation_x3.json")
    runs = {r["mode"]: r for r in json.load(open(p))["runs"]}
    assert runs["w:x3w/g+a+s:x3"]["final_max_abs"] <= 5e-6
    assert 5e-4 < runs["w:x3w/g+a+s:fp16"]["final_max_abs"] < 1e-3       # the two-MFMA candidate: inside the gate by 12 % only
    assert runs["w:bf16x3/g+a+s:bf16x3"]["final_max_abs"] <= 5e-5


def test_winograd_emulation_matches_direct_convolution_and_its_recorded_drift(golden_dir):
    """oracle/drift_ablation.py's Winograd F(2x2, 3x3) path (the CPU half of the go / no-go of DESIGN.md 9 R4-7): exact against F.conv2d
    without rounding, and the recorded C2 numbers (tests/golden/drift_ablation_wino.json) say what 16-bit Winograd costs: the first bf16
    forward stays under the reference's own bf16 mode (1.09e-2 of scale), the 50-step final latent under its 2.9e-2."""
    import json
    import os
    import torch.nn.functional as F
    from oracle import drift_ablation as da

    class NoRound:
        kinds = {"wino"}
        fn = {"wino": None}

        def __call__(self, x, kind):
            return x

    g = torch.Generator().manual_seed(0)
    x, w, b = torch.randn(2, 24, 12, 8, generator=g), torch.randn(40, 24, 3, 3, generator=g) * 0.1, torch.randn(40, generator=g)
    da._UCACHE.clear()
    ref = F.conv2d(x, w, b, padding=1)
    assert (da.conv3(x, w, b, NoRound()) - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    da._UCACHE.clear()
    r = da.Rounder("wino:bf16")
    xb = x.bfloat16().float()
    e_w = (da.conv3(xb, w, b, r) - ref).abs().max().item()
    e_d = (F.conv2d(xb, w.bfloat16().float(), b, padding=1) - ref).abs().max().item()
    assert e_d < e_w < 4 * e_d        # Winograd in bf16 is worse than direct bf16, by a small factor
    da._UCACHE.clear()
    runs = {r_["mode"]: r_ for r_ in json.load(open(os.path.join(golden_dir, "drift_ablation_wino.json")))["runs"]}
    bw, fw = runs["all:bf16/wino:bf16"], runs["all:fp16/wino:fp16"]
    assert bw["first_forward_rel"] < 1.09e-2 and bw["final_max_abs"] < 2.9e-2 and bw["final_rms"] < 6.1e-3
    assert fw["first_forward_rel"] < 2e-3 and fw["final_max_abs"] < 4.7e-3


def test_shipped_tile_table_covers_the_split_precision_engine():
    """Round 4: the split-precision engine (dtype code 3) resolves its C2 / C3 / C4 problems from the shipped table like the other engines
    (bench: tile_configs_measured_in_this_process = 0), only through algorithms its kernels exist for (csrc/tuning.h: no algos 2 / 6 / 20,
    i.e. no first-generation halo variants and no weight-streaming kernel, which needs 16-bit fragment-major weights)."""
    rows = [l.split() for l in open(_lib.TILE_TABLE_PATH) if l.strip() and not l.startswith("#")]
    x3 = [r for r in rows if int(r[0]) == _lib.K22_F16X3]
    assert len(x3) >= 280
    algos = {int(r[r.index("|") + 1]) if "|" in r else int(r[10]) for r in x3}
    assert algos <= {0, 1, 3, 4, 5, 7, 10, 11, 12} and not (algos & {2, 6, 20})
    # the C2 shapes: 3x3 convolutions at 96 / 48 / 24 / 12 pixels, batch 2
    c2 = {(int(r[6]), int(r[7])) for r in x3 if int(r[1]) == 9 and int(r[2]) == 2 * int(r[6]) * int(r[7])}
    assert {(96, 96), (48, 48), (24, 24), (12, 12)} <= c2


# ---- CLIP byte-pair tokenizer (round 5, ADVICE r4: the package goes prompt -> image on its own) -----------------------------------------
def _train_merges(corpus, n):
    """a small byte-pair merge list in the format of bpe_simple_vocab_16e6.txt (what OpenAI's tokenizer reads): greedy most-frequent pair"""
    from collections import Counter
    from kandinsky2_amd.tokenizer import bytes_to_unicode
    b2u = bytes_to_unicode()
    words = Counter()
    for w in corpus.lower().split():
        sym = [b2u[b] for b in w.encode("utf-8")]
        sym[-1] += "</w>"
        words[tuple(sym)] += 1
    merges = []
    for _ in range(n):
        pairs = Counter()
        for w, c in words.items():
            for a, b in zip(w[:-1], w[1:]):
                pairs[(a, b)] += c
        if not pairs:
            break
        best = max(sorted(pairs), key=lambda p: pairs[p])
        merges.append(best)
        new = Counter()
        for w, c in words.items():
            out, i = [], 0
            while i < len(w):
                if i + 1 < len(w) and (w[i], w[i + 1]) == best:
                    out.append(w[i] + w[i + 1]); i += 2
                else:
                    out.append(w[i]); i += 1
            new[tuple(out)] += c
        words = new
    return merges


def test_clip_bpe_tokenizer_equals_the_transformers_implementation(tmp_path):
    """kandinsky2_amd.tokenizer.ClipBPETokenizer restates OpenAI clip's SimpleTokenizer (an un-vendored dependency of the reference,
    prior.py:10-12, 387-416).  transformers' CLIPTokenizer is an independent implementation of the same algorithm: both are given the
    same generated merges list and must produce the same ids; padded_tokens_and_mask follows the reference's wrapper."""
    import gzip
    import json
    from kandinsky2_amd.tokenizer import ClipBPETokenizer, _byte_symbols
    corpus = ("a red cat sitting on the green grass near the old house , photo in 4k resolution . the cat's whiskers are long ; "
              "don't say it's a dog ! café naïve über straße красная кошка сидит на траве 12 345 cats & dogs (best) #1 ") * 3
    merges = _train_merges(corpus, 180)
    bpe = tmp_path / "bpe_simple_vocab_16e6.txt.gz"
    with gzip.open(bpe, "wb") as f:
        f.write(("#version: 0.2\n" + "\n".join(a + " " + b for a, b in merges) + "\n").encode("utf-8"))
    tok = ClipBPETokenizer(str(bpe))
    assert len(tok.encoder) == 512 + len(merges) + 2 and tok.sot_token == 512 + len(merges) and tok.eot_token == tok.sot_token + 1
    # the same vocabulary in the HF file format
    base = _byte_symbols()
    vocab = base + [s + "</w>" for s in base] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
    (tmp_path / "vocab.json").write_text(json.dumps({s: i for i, s in enumerate(vocab)}), encoding="utf-8")
    (tmp_path / "merges.txt").write_text("#version: 0.2\n" + "\n".join(a + " " + b for a, b in merges) + "\n", encoding="utf-8")
    from transformers import CLIPTokenizer
    hf = CLIPTokenizer(str(tmp_path / "vocab.json"), str(tmp_path / "merges.txt"))
    prompts = ["a red cat, 4k photo", "The CAT's whiskers   are long;  don't say it's a dog!", "green grass near the old house.", "café naïve über straße",
               "красная кошка сидит на траве", "12 345 cats & dogs (best) #1", "", "unseenword xyzzy", "tabs\tand\nnewlines"]
    for ptxt in prompts:
        want = hf(ptxt, add_special_tokens=False)["input_ids"]
        assert tok.encode(ptxt) == want, ptxt
        assert tok.encode(tok.decode(tok.encode(ptxt))) == want      # decode separates the word-final symbols by spaces, as OpenAI's does
    # html entities are unescaped (twice) before anything else, as OpenAI's basic_clean does (transformers' fallback cleaner does not)
    assert tok.encode("cats &amp;amp; dogs") == tok.encode("cats & dogs")
    # ftfy.fix_text (the first step of the reference's basic_clean): typographic quotes, decomposed accents, fullwidth letters,
    # ligatures and control characters tokenize like their plain forms - the fixes of ftfy's default configuration, restated where ftfy is absent
    for fancy, plain in (("the cat\u2019s \u201cwhiskers\u201d", "the cat's \"whiskers\""), ("cafe\u0301 nai\u0308ve", "caf\u00e9 na\u00efve"),
                         ("\uff52\uff45\uff44 \uff43\uff41\uff54", "red cat"), ("\ufb01ne \ufb02oor", "fine floor"), ("a red\x07 cat\u0085photo", "a red cat photo")):
        assert tok.encode(fancy) == tok.encode(plain), fancy
    # the plain-text merges.txt form reads the same
    assert ClipBPETokenizer(str(tmp_path / "merges.txt")).encode(prompts[1]) == tok.encode(prompts[1])
    # padded_tokens_and_mask: prior.py:394-416
    ids, mask = tok.padded_tokens_and_mask(["a red cat", " ".join(["cat"] * 100)], 77)
    assert ids.dtype == torch.int and ids.shape == (2, 77) and mask.dtype == torch.bool
    n0 = 2 + len(tok.encode("a red cat"))
    assert ids[0, 0] == tok.sot_token and ids[0, n0 - 1] == tok.eot_token and (ids[0, n0:] == 0).all() and mask[0].sum() == n0
    assert ids[1, 0] == tok.sot_token and ids[1, 76] == tok.eot_token and mask[1].all()      # truncated: still ends in <|endoftext|>


def test_x2_precision_plan_follows_the_operand_rounding_ablation(golden_dir):
    """The K22_F16X2 plan (csrc/engine.hip: conv_dt / gemm_dt) is read off oracle/drift_ablation.py's first-forward runs (committed:
    tests/golden/drift_ablation_x2.json).  Asserted here: (1) the error variance is ADDITIVE over operand classes - which is what lets a plan
    be composed from per-class prices; (2) the classes the plan keeps at three MFMAs are the expensive ones; (3) the final-latent rms the GPU
    measured for the four plans follows final_rms^2 = c * first_forward_rms^2 with the c of the all-fp16-activations run."""
    import json
    with open(os.path.join(golden_dir, "drift_ablation_x2.json")) as f:
        ab = json.load(f)
    v = {r["mode"].replace("w:x3w/", ""): r["first_forward_rms"] ** 2 for r in ab["first_forward_only"]}
    whole = v["g+a+s:fp16"]
    assert abs(v["g:fp16"] + v["a:fp16"] + v["s:fp16"] - whole) <= 0.02 * whole
    assert abs(v["g@96:fp16"] + v["g@48:fp16"] + v["g@24:fp16"] + v["g@12:fp16"] - v["g:fp16"]) <= 0.02 * v["g:fp16"]
    assert abs(v["g1:fp16"] + v["g2:fp16"] + v["gq:fp16"] + v["go:fp16"] - v["g:fp16"]) <= 0.02 * v["g:fp16"]
    # what the plan keeps at the full split: skip inputs (half of everything), the out head, the top level (default plan 0)
    assert v["s:fp16"] > 0.45 * whole and v["go:fp16"] > 0.12 * whole and v["g1@96:fp16"] + v["g2@96:fp16"] > 0.2 * whole
    # what it runs at two MFMAs (one in the attention) is cheap
    cheap = v["g@48:fp16"] + v["g@24:fp16"] + v["g@12:fp16"] + v["gq:fp16"] + v["a:fp16"]
    assert cheap < 0.1 * whole
    c = 0.3725   # (final rms 1.99e-4)^2 / (first-forward rms 3.26e-4)^2 of 'w:x3w/g+a+s:fp16' (drift_ablation_x3.json)
    plans = {"0": cheap, "1": cheap + v["g1@96:fp16"], "2": cheap + v["g2@96:fp16"], "3": cheap + v["g1@96:fp16"] + v["g2@96:fp16"]}
    for k, ff2 in plans.items():
        pred = (c * ff2) ** 0.5
        meas = ab["gpu_measured_plans_c2"][k]["rms"]
        assert abs(pred - meas) <= 0.08 * meas, (k, pred, meas)
    # fp16 WEIGHTS at the 12x12 level: cheap in one forward, but a systematic error - the 50-step run is what prices it
    base = next(r for r in ab["first_forward_only"] if r["mode"].endswith("gq+a:fp16"))
    w12 = next(r for r in ab["first_forward_only"] if r["mode"].endswith("/w@12:fp16"))
    assert w12["first_forward_rms"] ** 2 < 1.1 * base["first_forward_rms"] ** 2
    f12 = next(r for r in ab["final_runs"] if "w@12+s@12" in r["mode"] and "w@24" not in r["mode"])
    assert f12["final_max_abs"] < 5e-4 and f12["final_rms"] ** 2 > 1.4 * ab["gpu_measured_plans_c2"]["0"]["rms"] ** 2
    # ... the 24x24 level as well breaks the 5e-4 bar; and the emulation of plan 0 itself lands on what the GPU measured (5.4e-5 rms)
    f24 = next(r for r in ab["final_runs"] if "w@24" in r["mode"])
    assert f24["final_max_abs"] > 5e-4
    f0 = next(r for r in ab["final_runs"] if r["mode"].endswith("gq+a:fp16"))
    assert abs(f0["final_rms"] - ab["gpu_measured_plans_c2"]["0"]["rms"]) <= 0.08 * f0["final_rms"] and f0["final_max_abs"] < 5e-4


# ---- static audit of the built library (round 5): what DESIGN.md says about registers is checked on the code objects themselves ----------
def _kernel_metadata(tmp_path):
    """name -> {vgpr, agpr, sgpr, scratch, lds} of every kernel in libk22hip.so (the gfx950 code objects inside its fat binary, extracted with
    llvm-objdump --offloading from a COPY of the library; metadata notes read with llvm-readelf)."""
    import glob
    import re
    import shutil
    import subprocess
    from kandinsky2_amd import _lib
    llvm = "/opt/rocm/lib/llvm/bin"
    if not (os.path.exists(_lib.LIB_PATH) and os.path.exists(llvm + "/llvm-objdump")):
        pytest.skip("library or llvm tools not present")
    so = str(tmp_path / "lib.so")
    shutil.copy(_lib.LIB_PATH, so)
    subprocess.run([llvm + "/llvm-objdump", "--offloading", so], check=True, capture_output=True, cwd=str(tmp_path))
    out = {}
    for co in sorted(glob.glob(so + ".*gfx950")):
        notes = subprocess.run([llvm + "/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
        for blk in notes.split("  - .agpr_count:")[1:]:
            def g(key):
                m = re.search(r"\." + key + r":\s+(\S+)", blk)
                return m.group(1) if m else None
            name = g("name")
            if name:
                out[name] = dict(agpr=int(blk.split()[0]), vgpr=int(g("vgpr_count")), sgpr=int(g("sgpr_count")),
                                 scratch=int(g("private_segment_fixed_size")), lds=int(g("group_segment_fixed_size")))
    return out


def test_no_kernel_of_the_library_spills_and_the_register_claims_hold(tmp_path):
    """A spilled MFMA kernel still passes parity and loses 2x (MI355X_MICROARCH.md: spills at 512 registers per SIMD lane, i.e. 256 per wave
    of a 512-thread workgroup).  Asserted on the shipped binary: (1) NO kernel uses scratch; (2) the 512-thread convolution / GEMM kernels
    fit two waves per SIMD (<= 256 VGPR + AGPR); (3) the asymmetric split's two-set fragment pipeline exists at BM = 256 (its activation
    fragments are 4 registers) - the instantiation the x3 arithmetic cannot have; (4) attention keeps two workgroups per CU."""
    md = _kernel_metadata(tmp_path)
    assert len(md) > 400, len(md)
    spilled = {k: v["scratch"] for k, v in md.items() if v["scratch"]}
    assert not spilled, spilled
    big = {k: v for k, v in md.items() if any(t in k for t in ("conv3_halo_spec_kernel", "conv3_halo_kernel", "conv3_halo3_kernel", "gemm8_kernel"))}
    assert len(big) > 100
    over = {k: v["vgpr"] + v["agpr"] for k, v in big.items() if v["vgpr"] + v["agpr"] > 256}
    assert not over, over
    x2_pipe_256 = [k for k in md if "conv3_halo_spec_kernelI4x2_tLi256E" in k and "Lb1E" in k]
    x3_pipe_256 = [k for k in md if "conv3_halo_spec_kernelI4x3_tLi256E" in k and "Lb1E" in k]
    assert len(x2_pipe_256) == 4 and not x3_pipe_256
    for k in x2_pipe_256:
        assert 200 <= md[k]["vgpr"] <= 256
    att = {k: v for k, v in md.items() if k.startswith("_Z16attention_kernel")}
    assert len(att) == 5 and all(v["vgpr"] + v["agpr"] <= 256 for v in att.values())
    # round 6: the software-pipelined kernel holds two score sets + all 16 fragments of a tile and still fits two workgroups per CU
    pipe = {k: v for k, v in md.items() if k.startswith("_Z21attention_pipe_kernel")}
    assert len(pipe) == 3 and all(v["vgpr"] + v["agpr"] <= 256 and v["lds"] == 32768 for v in pipe.values()), pipe


def test_isa_mix_tool_counts_a_synthetic_loop():
    """tools/isa_mix.py (the static instruction-mix report of profiles/r05_isa_mix.txt): loop detection through the `<kernel+0xOFF>` branch
    targets and the cycle tables, on a hand-made body."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_mix", os.path.join(ROOT, "tools", "isa_mix.py"))
    im = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(im)
    body = [(0x100, "s_load_dwordx2", "s[0:1], s[4:5], 0x0"),
            (0x108, "ds_read_b128", "v[0:3], v9"),                        # loop head
            (0x110, "ds_read2st64_b64", "v[4:7], v9 offset1:8"),
            (0x118, "s_waitcnt", "lgkmcnt(0)"),
            (0x11c, "v_mfma_f32_32x32x16_bf16", "v[10:25], v[0:3], v[4:7], v[10:25]"),
            (0x124, "v_mov_b32_e32", "v0, v4"),
            (0x128, "v_mfma_f32_16x16x32_f16", "v[26:29], v[0:3], v[4:7], v[26:29]"),
            (0x130, "s_barrier", ""),
            (0x134, "s_cbranch_scc1", "65524 <k+0x8>"),
            (0x138, "s_endpgm", "")]
    assert im.loops_of(body) == [(1, 8)]
    c, cyc, kinds = im.mix(body, 1, 8)
    assert c["mfma"] == 2 and cyc["mfma"] == 32 + 16
    assert cyc["lds"] == 4 + 8 and c["ds_read"] == 2          # ds_read_b128: 4 LDS cycles; ds_read2 of 8-byte words: 8
    assert c["valu_mov"] == 1 and cyc["valu"] == 4 and c["s_barrier"] == 1 and c["s_waitcnt"] == 1
    assert im.lds_cycles("ds_write_b128") == 13 and im.lds_cycles("ds_read_b64") == 2 and im.lds_cycles("ds_read_b64_tr_b16") == 2
    assert im.classify("global_load_lds_dwordx4") == "lds_dma" and im.classify("global_load_dwordx4") == "vmem_load"
