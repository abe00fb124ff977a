"""GPU parity at the FULL shapes of BASELINE.json's configs, against golden outputs of the REFERENCE's own modules
(oracle/make_golden.py --only c2,c3,c4: reference create_model(...) 1.23 B params + verbatim model_fn +
SpacedDiffusion.p_sample_loop_progressive with injected noise, kandinsky2_1_model.py:222-257,
gaussian_diffusion.py:384-475).

  C2  text2img 768x768, bs 1, 50 steps  -> CFG batch [2,4,96,96]      (the configuration bench.py times)
  C3  1024x1024, 4 images per GPU       -> one forward of [8,4,128,128]
  C4  inpainting 768x768, bs 4, 50 steps -> [8,9ch,96,96] + masked-latent blend

Gates:
  * fp32 engine (exact-fp32 MFMA): final latent within 1e-3 max-abs of the reference p_sampler (the north-star statement),
    every stored intermediate latent too; first forward within 2e-4 of the output scale;
  * split-precision engine (backend_dtype="f16x3", round 4: fp32 tensors, fp16 (hi, lo) operand pairs, three fp16 MFMAs per product): the
    same 1e-3 gate, asserted at 1e-4, at C2 and C4 - the mode that carries the gate at several times the fp32 engine's speed;
  * bf16 engine (the benchmarked path): bf16 storage of activations / weights cannot meet 1e-3 after 50 chained,
    thresholded steps of a random-weight UNet (the reference's own fp16 mode does not either: DESIGN.md section 3 has its
    measured drift); its max-abs / rms distance from the fp32 reference is MEASURED here and bounded at 2x the value
    observed on MI355X with the shipped tile table (BF16_BOUNDS below; numbers in DESIGN.md section 3), so a wrong tile
    variant, a broken split-K or a mis-rounded epilogue shows up as a failed bound, not as "drift".
"""
import json
import os

import pytest
import torch

import kandinsky2_amd as k22

pytestmark = pytest.mark.gpu

# bounds = 2x the values measured on MI355X with the shipped tile table (profiles/r02_parity_*.json):
#   first_forward_rel: max-abs of one raw UNet output / its scale      (C2 measured 7.4e-3)
#   traj: (max-abs, rms) over every stored latent of the 50-step loop  (C2 measured 2.8e-2 at step 25, 3.6e-3 rms at step 50)
BF16_BOUNDS = {
    "c2_text2img": {"first_forward_rel": 1.5e-2, "traj": (0.06, 0.0075)},
    "c4_inpaint": {"first_forward_rel": 1.7e-2, "traj": (0.04, 0.003)},        # measured 8.1e-3; 1.85e-2 at step 25, 1.3e-3 rms
    "c3_forward": {"first_forward_rel": 1.5e-2},                                 # measured 7.4e-3
}

# fp16 engine (the reference's own use_fp16 mode; same kernels and speed as bf16, 3 more mantissa bits): bounds = 2x the distance the
# storage-rounding emulation of oracle/drift_ablation.py predicts at C2 (tests/golden/drift_ablation.json "all:fp16": first forward
# 8.8e-4 of scale, final latent 2.8e-3 max-abs / 5.1e-4 rms); the measured values are printed and recorded
FP16_BOUNDS = {
    "c2_text2img": {"first_forward_rel": 1.8e-3, "traj": (8e-3, 1.1e-3), "final": (5.6e-3, 1.1e-3)},
    "c4_inpaint": {"first_forward_rel": 2.0e-3, "traj": (5e-3, 4e-4), "final": (5e-3, 4e-4)},
}

_SD = {}
_REPORT = {}


def _slow_unless_asked(cond, why):
    if cond and os.environ.get("K22_RUN_SLOW", "1") == "0":
        pytest.skip(why + " - skipped because K22_RUN_SLOW=0")


def _load(golden_dir, name):
    p = os.path.join(golden_dir, name + ".pt")
    if not os.path.exists(p):
        pytest.skip(f"{name}.pt not generated")
    return torch.load(p, weights_only=False)


def _state_dict(inpainting):
    if inpainting not in _SD:
        _SD.clear()   # one 4.9 GB fp32 state dict at a time
        arch = k22.make_arch(k22.MODEL_CONFIG_2_1, inpainting=inpainting)
        _SD[inpainting] = (arch, k22.init_unet_state_dict(arch, seed=0))
    return _SD[inpainting]


def _model(inpainting, backend):
    arch, sd = _state_dict(inpainting)
    m = k22.Text2ImUNetHIP(arch, backend_dtype=backend, use_graph=True)
    m.load_state_dict(sd)
    m = m.to("cuda").eval()
    m.prepare(free_params=True)
    return arch, m


def _record(name, key, **vals):
    _REPORT.setdefault(name, {})[key] = vals
    out = os.environ.get("K22_PARITY_REPORT")
    if out:
        with open(out, "w") as f:
            json.dump(_REPORT, f, indent=1, sort_keys=True)


def _loop_case(fx, backend):
    """Runs the fused p_sampler on the fixture's seeded inputs; returns {step: latent} for the stored steps + 'final' + the
    first raw UNet output."""
    inp, bs, B, lat, steps = fx["inpainting"], fx["bs"], fx["B"], fx["lat"], fx["steps"]
    arch, m = _model(inp, backend)
    full, pooled, image = k22.make_conditioning(arch, B, seed=2)
    g = torch.Generator().manual_seed(42)
    x_T = torch.randn(B, 4, lat, lat, generator=g)
    noise_seq = torch.randn(steps, B, 4, lat, lat, generator=g)
    kw = dict(full_emb=full.cuda(), pooled_emb=pooled.cuda(), image_emb=image.cuda())
    ii = mm = None
    if inp:
        g3 = torch.Generator().manual_seed(3)
        _x = torch.randn(B, 4, lat, lat, generator=g3)
        ii = torch.randn(B, 4, lat, lat, generator=g3).cuda()
        mm = torch.zeros(B, 1, lat, lat)
        mm[..., : lat // 2] = 1.0
        mm = mm.cuda()
        kw.update(inpaint_image=ii * mm, inpaint_mask=mm)
    d = k22.create_gaussian_diffusion(**dict(k22.DIFFUSION_CONFIG_2_1, timestep_respacing=str(steps)))
    first = m(torch.cat([x_T[:bs], x_T[:bs]], 0).cuda(), fx["first_ts"].float().cuda(), **kw).cpu()
    out = {}
    # the loop is run in segments that end at the stored steps (the engine keeps no state between steps other than x, so a
    # segmented run is the same computation as one p_sample_loop call: asserted bit-for-bit in the fp32 test)
    x = x_T.cuda()
    done = 0
    for stop in sorted(fx["traj"].keys()) + [steps]:
        if stop > done:
            x = _segment(d, m, x, kw, fx, noise_seq, done, stop, ii, mm)
            done = stop
        out[stop] = x.cpu()
    out["final"] = out[steps]
    return first, out


def _segment(d, m, x, kw, fx, noise_seq, done, stop, ii, mm):
    """steps done+1 .. stop of the T-step loop: loop indices T-1-done down to T-stop."""
    from kandinsky2_amd import _lib
    L = _lib.lib()
    B, lat, T, bs = fx["B"], fx["lat"], fx["steps"], fx["bs"]
    HW = lat * lat
    table = torch.from_numpy(d.step_table()).cuda()
    ts_rows = torch.from_numpy(d.model_timesteps()).cuda()[:, None].expand(-1, B).contiguous()
    scratch = torch.empty(L.k22_sampler_scratch_bytes(B, HW), dtype=torch.uint8, device="cuda")
    lo, gamma = k22.percentile_index(4 * HW)
    x = x.clone()
    x_next = torch.empty_like(x)
    for k in range(done, stop):
        i = T - 1 - k
        half = x[:bs]
        out = m(torch.cat([half, half], 0), ts_rows[i], **kw)
        nz = noise_seq[k].cuda().contiguous()
        _lib.check(L.k22_sampler_step(x.data_ptr(), out.data_ptr(), nz.data_ptr(), _lib.ptr(ii), _lib.ptr(mm), table.data_ptr(), i,
                                      float(fx["guidance"]), 1, -2.0, 2.0, lo, gamma, scratch.data_ptr(), x_next.data_ptr(), None, B, HW,
                                      _lib.current_stream()))
        x, x_next = x_next, x
    return x


def _dist(a, b):
    d = (a - b).float()
    return d.abs().max().item(), d.pow(2).mean().sqrt().item()


@pytest.mark.parametrize("name", ["c2_text2img", "c4_inpaint"])
def test_full_size_p_sampler_fp32_gate(golden_dir, name):
    """North-star gate at the benchmarked shape: reference p_sampler, fixed seed, injected noise, 50 steps, <= 1e-3 max-abs."""
    _slow_unless_asked(name == "c4_inpaint", "C4 on the 25-steps/s fp32 engine (19 s); C4 is gated by default on f16x3 (1e-4) and f16x2 (5e-4)")
    fx = _load(golden_dir, name)
    first, traj = _loop_case(fx, torch.float32)
    scale = fx["first_out"].abs().max().item()
    e_first = (first - fx["first_out"]).abs().max().item()
    print(f"{name} fp32: first forward max|d| {e_first:.3e} (scale {scale:.2f})")
    _record(name, "fp32_first_forward", max_abs=e_first, scale=scale)
    assert e_first <= 2e-4 * scale
    for n in sorted(fx["traj"].keys()):
        ma, rms = _dist(traj[n], fx["traj"][n])
        print(f"{name} fp32: latent after step {n:2d}: max|d| {ma:.3e} rms {rms:.3e}")
        _record(name, f"fp32_step{n}", max_abs=ma, rms=rms)
        assert ma <= 1e-3
    ma, rms = _dist(traj["final"], fx["final"])
    print(f"{name} fp32: FINAL latent ({fx['steps']} steps): max|d| {ma:.3e} rms {rms:.3e}")
    _record(name, "fp32_final", max_abs=ma, rms=rms)
    assert ma <= 1e-3
    # the loop entry point gives the same bits as the segmented run above
    inp = fx["inpainting"]
    if not inp:
        arch, m = _model(inp, torch.float32)
        B, lat, steps = fx["B"], fx["lat"], fx["steps"]
        full, pooled, image = k22.make_conditioning(arch, B, seed=2)
        g = torch.Generator().manual_seed(42)
        x_T = torch.randn(B, 4, lat, lat, generator=g)
        noise_seq = torch.randn(steps, B, 4, lat, lat, generator=g)
        d = k22.create_gaussian_diffusion(**dict(k22.DIFFUSION_CONFIG_2_1, timestep_respacing=str(steps)))
        whole = d.p_sample_loop(m, (B, 4, lat, lat), model_kwargs=dict(full_emb=full.cuda(), pooled_emb=pooled.cuda(), image_emb=image.cuda()),
                                guidance_scale=fx["guidance"], noise=x_T.cuda(), noise_seq=noise_seq.cuda()).cpu()
        assert torch.equal(whole, traj["final"])
        # ... and so does the WHOLE loop replayed as one hipGraph (k22_unet_sample_loop: 50 forwards + 50 sampler steps, one launch)
        m.del_cache()
        one_graph = d.p_sample_loop(m, (B, 4, lat, lat), model_kwargs=dict(full_emb=full.cuda(), pooled_emb=pooled.cuda(), image_emb=image.cuda()),
                                    guidance_scale=fx["guidance"], noise=x_T.cuda(), noise_seq=noise_seq.cuda(), whole_loop_graph=True).cpu()
        assert torch.equal(one_graph, traj["final"])


@pytest.mark.parametrize("name", ["c2_text2img", "c4_inpaint"])
def test_full_size_p_sampler_split_precision_gate(golden_dir, name):
    """The north-star gate on the engine mode built to carry it at 16-bit MFMA rate (round 4): backend_dtype="f16x3" - fp32 tensors, every
    MFMA operand an fp16 (hi, lo) pair, three v_mfma_f32_32x32x16_f16 per product (include/k22.h: K22_F16X3).  Reference p_sampler
    (kandinsky2_1_model.py:245-257 -> gaussian_diffusion.py:384-475), fixed seed, injected noise, 50 steps: <= 1e-3 max-abs on the final
    latent and on every stored step, at C2 (the benchmarked shape) and C4 (inpainting bs 4).  Asserted at 1e-4: the mode is fp32-class
    (measured 3.5e-6 at C2; oracle/drift_ablation.py's emulation 'w:x3w/g+a+s:x3' predicts 2.1e-6), a tenth of the gate still catches a
    broken lo half (losing either cross term puts it at ~1e-3)."""
    fx = _load(golden_dir, name)
    first, traj = _loop_case(fx, k22.F16X3)
    scale = fx["first_out"].abs().max().item()
    e_first = (first - fx["first_out"]).abs().max().item()
    print(f"{name} f16x3: first forward max|d| {e_first:.3e} = {e_first / scale:.3e} of scale {scale:.2f}")
    _record(name, "f16x3_first_forward", max_abs=e_first, scale=scale, rel=e_first / scale)
    assert e_first <= 2e-5 * scale
    for n in sorted(fx["traj"].keys()):
        ma, rms = _dist(traj[n], fx["traj"][n])
        print(f"{name} f16x3: latent after step {n:2d}: max|d| {ma:.3e} rms {rms:.3e}")
        _record(name, f"f16x3_step{n}", max_abs=ma, rms=rms)
        assert ma <= 1e-4
    ma, rms = _dist(traj["final"], fx["final"])
    print(f"{name} f16x3: FINAL latent ({fx['steps']} steps): max|d| {ma:.3e} rms {rms:.3e}")
    _record(name, "f16x3_final", max_abs=ma, rms=rms)
    assert ma <= 1e-4 and rms <= 2e-5
    if not fx["inpainting"]:
        # the whole loop as one hipGraph replay gives the same bits as the segmented run
        arch, m = _model(False, k22.F16X3)
        B, lat, steps = fx["B"], fx["lat"], fx["steps"]
        full, pooled, image = k22.make_conditioning(arch, B, seed=2)
        g = torch.Generator().manual_seed(42)
        x_T = torch.randn(B, 4, lat, lat, generator=g)
        noise_seq = torch.randn(steps, B, 4, lat, lat, generator=g)
        d = k22.create_gaussian_diffusion(**dict(k22.DIFFUSION_CONFIG_2_1, timestep_respacing=str(steps)))
        one_graph = d.p_sample_loop(m, (B, 4, lat, lat), model_kwargs=dict(full_emb=full.cuda(), pooled_emb=pooled.cuda(), image_emb=image.cuda()),
                                    guidance_scale=fx["guidance"], noise=x_T.cuda(), noise_seq=noise_seq.cuda(), whole_loop_graph=True).cpu()
        assert torch.equal(one_graph, traj["final"])


@pytest.mark.parametrize("name", ["c2_text2img", "c4_inpaint"])
def test_full_size_p_sampler_asymmetric_split_gate(golden_dir, name):
    """Round 5: the north-star gate on the ASYMMETRIC split engine, backend_dtype="f16x2" (include/k22.h: K22_F16X2) - weights
    always as fp16 (hi, lo) pairs, the activation operand at fp16 precision where oracle/drift_ablation.py says its rounding is cheap (two
    MFMAs per product; attention one), the full split where it is not (skip connections, the out head, the top level under the default
    plan).  Asserted at 5e-4 max-abs on the final latent and on every stored step (measured at C2: 2.7e-4 / 5.4e-5 rms - the ablation's
    prediction for this plan is 5.6e-5 rms, tests/golden/drift_ablation_x2.json), i.e. HALF the north star's 1e-3."""
    fx = _load(golden_dir, name)
    first, traj = _loop_case(fx, k22.F16X2)
    scale = fx["first_out"].abs().max().item()
    e_first = (first - fx["first_out"]).abs().max().item()
    print(f"{name} f16x2: first forward max|d| {e_first:.3e} = {e_first / scale:.3e} of scale {scale:.2f}")
    _record(name, "f16x2_first_forward", max_abs=e_first, scale=scale, rel=e_first / scale)
    assert e_first <= 3e-4 * scale
    for n in sorted(fx["traj"].keys()):
        ma, rms = _dist(traj[n], fx["traj"][n])
        print(f"{name} f16x2: latent after step {n:2d}: max|d| {ma:.3e} rms {rms:.3e}")
        _record(name, f"f16x2_step{n}", max_abs=ma, rms=rms)
        assert ma <= 5e-4
    ma, rms = _dist(traj["final"], fx["final"])
    print(f"{name} f16x2: FINAL latent ({fx['steps']} steps): max|d| {ma:.3e} rms {rms:.3e}")
    _record(name, "f16x2_final", max_abs=ma, rms=rms)
    assert ma <= 5e-4 and rms <= 1e-4


@pytest.mark.parametrize("name", ["c2_text2img", "c4_inpaint"])
def test_full_size_p_sampler_bf16_measured_bound(golden_dir, name):
    """The benchmarked dtype at the benchmarked shape: distance of the bf16 engine's latents from the fp32 reference,
    measured, reported, and bounded at 2x the value observed with the shipped tile table."""
    fx = _load(golden_dir, name)
    first, traj = _loop_case(fx, torch.bfloat16)
    scale = fx["first_out"].abs().max().item()
    e_first = (first - fx["first_out"]).abs().max().item()
    print(f"{name} bf16: first forward max|d| {e_first:.3e} = {e_first / scale:.3e} of scale {scale:.2f}")
    _record(name, "bf16_first_forward", max_abs=e_first, scale=scale, rel=e_first / scale)
    bound = BF16_BOUNDS[name]
    worst = (0.0, 0.0)
    for n in sorted(fx["traj"].keys()):
        ma, rms = _dist(traj[n], fx["traj"][n])
        worst = (max(worst[0], ma), max(worst[1], rms))
        print(f"{name} bf16: latent after step {n:2d}: max|d| {ma:.3e} rms {rms:.3e}")
        _record(name, f"bf16_step{n}", max_abs=ma, rms=rms)
    ma, rms = _dist(traj["final"], fx["final"])
    worst = (max(worst[0], ma), max(worst[1], rms))
    print(f"{name} bf16: FINAL latent ({fx['steps']} steps): max|d| {ma:.3e} rms {rms:.3e}  (latent range [-1, 1])")
    _record(name, "bf16_final", max_abs=ma, rms=rms)
    assert torch.isfinite(traj["final"]).all()
    assert e_first <= bound["first_forward_rel"] * scale
    assert worst[0] <= bound["traj"][0] and worst[1] <= bound["traj"][1]


@pytest.mark.parametrize("name", ["c2_text2img", "c4_inpaint"])
def test_full_size_p_sampler_fp16_measured_bound(golden_dir, name):
    """backend_dtype=torch.float16 - the reference's own reduced-precision mode (use_fp16=True + convert_to_fp16(),
    kandinsky2_1_model.py:92-97) and the cheapest engine mode whose 50-step final latent stays within ~3e-3 of the fp32 reference
    (oracle/drift_ablation.py: bf16 WEIGHT rounding alone costs 1.4e-2, so no bf16 mode can): same kernels, bytes and MFMA rate as bf16."""
    _slow_unless_asked(name == "c4_inpaint", "the fp16 engine at C4 (13 s); its C2 bound and the bf16 engine's C4 bound run by default")
    fx = _load(golden_dir, name)
    first, traj = _loop_case(fx, torch.float16)
    scale = fx["first_out"].abs().max().item()
    e_first = (first - fx["first_out"]).abs().max().item()
    print(f"{name} fp16: first forward max|d| {e_first:.3e} = {e_first / scale:.3e} of scale {scale:.2f}")
    _record(name, "fp16_first_forward", max_abs=e_first, scale=scale, rel=e_first / scale)
    bound = FP16_BOUNDS[name]
    worst = (0.0, 0.0)
    for n in sorted(fx["traj"].keys()):
        ma, rms = _dist(traj[n], fx["traj"][n])
        worst = (max(worst[0], ma), max(worst[1], rms))
        print(f"{name} fp16: latent after step {n:2d}: max|d| {ma:.3e} rms {rms:.3e}")
        _record(name, f"fp16_step{n}", max_abs=ma, rms=rms)
    ma, rms = _dist(traj["final"], fx["final"])
    print(f"{name} fp16: FINAL latent ({fx['steps']} steps): max|d| {ma:.3e} rms {rms:.3e}  (latent range [-1, 1])")
    _record(name, "fp16_final", max_abs=ma, rms=rms)
    assert torch.isfinite(traj["final"]).all()
    assert e_first <= bound["first_forward_rel"] * scale
    assert worst[0] <= bound["traj"][0] and worst[1] <= bound["traj"][1]
    assert ma <= bound["final"][0] and rms <= bound["final"][1]


@pytest.mark.parametrize("dtype_name,backend", [("bf16", torch.bfloat16), ("fp16", torch.float16)])
def test_reduced_precision_engines_are_closer_to_fp32_than_the_references_own_mode(golden_dir, dtype_name, backend):
    """The yardstick at the benchmarked shape (VERDICT r2 #1a): the REFERENCE's own reduced-precision mode (use_fp16=True +
    convert_to_fp16(), kandinsky2_1_model.py:92-97, run with fp16 - what it ships - and with bf16 storage) against its own fp32 mode,
    C2, 50 steps, same injected noise (oracle/ref_fp16_drift.py -> tests/golden/ref_{bf16,fp16}_drift_c2.json: final latent 2.9e-2 /
    4.7e-3 max-abs).  The HIP engine of the same storage type must be at least as close to the fp32 reference as that - at the first
    forward and at the final latent."""
    fx = _load(golden_dir, "c2_text2img")
    yp = os.path.join(golden_dir, f"ref_{dtype_name}_drift_c2.json")
    if not os.path.exists(yp):
        pytest.skip("yardstick not generated")
    y = json.load(open(yp))
    first, traj = _loop_case(fx, backend)
    e_first = (first - fx["first_out"]).abs().max().item()
    ma, rms = _dist(traj["final"], fx["final"])
    yf = y["steps"][str(fx["steps"])]
    print(f"c2 {dtype_name}: engine first forward {e_first:.3e} vs reference-{dtype_name} {y['first_forward']['max_abs']:.3e}; "
          f"final latent {ma:.3e} / rms {rms:.3e} vs reference-{dtype_name} {yf['max_abs']:.3e} / {yf['rms']:.3e}")
    _record("c2_text2img", f"{dtype_name}_vs_reference_{dtype_name}_mode", engine_final=ma, reference_mode_final=yf["max_abs"],
            engine_rms=rms, reference_mode_rms=yf["rms"])
    assert e_first <= y["first_forward"]["max_abs"] and ma <= yf["max_abs"] and rms <= yf["rms"]


def _compact_err(out, c):
    """max-abs distance on the stored sub-grid, row band and column band of a compact fixture (make_golden._compact)."""
    s = c["stride"]
    e = (out[..., ::s, ::s] - c["sub"]).abs().max().item()
    e = max(e, (out[..., c["r0"]: c["r0"] + c["rows"].shape[-2], :] - c["rows"]).abs().max().item())
    e = max(e, (out[..., :, c["c0"]: c["c0"] + c["cols"].shape[-1]] - c["cols"]).abs().max().item())
    return e


@pytest.mark.parametrize("backend", [torch.float32, torch.bfloat16, "f16x3"])
def test_c3_forward_1024px_batch8(golden_dir, backend):
    """C3's per-GPU shape (1024x1024, 4 images -> CFG batch 8 at 128x128 latents): one forward against the reference."""
    fx = _load(golden_dir, "c3_forward")
    arch, m = _model(False, backend)
    B, h, w = fx["B"], fx["h"], fx["w"]
    full, pooled, image = k22.make_conditioning(arch, B, seed=2)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 4, h, w, generator=g)
    out = m(x.cuda(), fx["t"].cuda(), full_emb=full.cuda(), pooled_emb=pooled.cuda(), image_emb=image.cuda()).cpu()
    scale = fx["absmax"]
    err = _compact_err(out, fx["forward_compact"])
    print(f"c3_forward {backend}: max|d| {err:.3e} = {err / scale:.3e} of scale {scale:.2f}")
    _record("c3_forward", {torch.float32: "fp32", torch.bfloat16: "bf16"}.get(backend, backend), max_abs=err, scale=scale, rel=err / scale)
    assert err <= (2e-4 if backend == torch.float32 else (2e-5 if backend == "f16x3" else BF16_BOUNDS["c3_forward"]["first_forward_rel"])) * scale


@pytest.mark.parametrize("backend,tol,tol_rms", [(torch.float32, 1e-3, 1e-5), ("f16x3", 1e-4, 1e-5), ("f16x2", 3.5e-3, 2e-4)])
def test_c3_shard_10_step_loop(golden_dir, backend, tol, tol_rms):
    """Round 5: the sampler at the C3 per-GPU shape (1024x1024, 4 images per GPU -> CFG batch [8,4,128,128]) pinned at LOOP
    level: 10 steps of the reference p_sampler (create_model + SpacedDiffusion.p_sample_loop_progressive with injected noise,
    oracle/make_golden.py --only c3loop), dynamic threshold over 65 536 values per image (gaussian_diffusion.py:284-294).  fp32 / f16x3: the
    north-star class bounds.  f16x2: a 10-step schedule takes steps five times coarser than the 50-step one the 5e-4 gate is stated on, and its
    first step multiplies the eps error by sqrt(1/abar - 1) = 14.5 before the clamp removes it from all but a few pixels - a heavy-tailed
    distance (measured: after step 1 1.2e-3 max-abs at 4.6e-5 rms - the same ratio, 23, shows at C2 -, after step 5 1.6e-3 / 1.0e-4, FINAL
    5.0e-4 / 6.7e-5; fp32 engine 1.3e-5, f16x3 6.4e-6) - so its bounds here are 2x those measurements."""
    fx = _load(golden_dir, "c3_loop")
    first, traj = _loop_case(fx, backend)
    scale = fx["first_out"].abs().max().item()
    e_first = (first - fx["first_out"]).abs().max().item()
    print(f"c3_loop {backend}: first forward {e_first / scale:.3e} of scale")
    assert e_first <= (3e-4 if backend == "f16x2" else 2e-4) * scale
    for n in sorted(fx["traj"].keys()):
        ma, rms = _dist(traj[n], fx["traj"][n])
        print(f"c3_loop {backend}: latent after step {n}: max|d| {ma:.3e} rms {rms:.3e}")
        _record("c3_loop", f"{backend}_step{n}".replace("torch.", ""), max_abs=ma, rms=rms)
        assert ma <= tol and rms <= tol_rms
    ma, rms = _dist(traj["final"], fx["final"])
    print(f"c3_loop {backend}: FINAL latent ({fx['steps']} steps): max|d| {ma:.3e} rms {rms:.3e}")
    _record("c3_loop", f"{backend}_final".replace("torch.", ""), max_abs=ma, rms=rms)
    assert ma <= tol and rms <= tol_rms
    assert ma <= 1e-3   # the north star's gate is stated on the FINAL latent: it holds for every gate-carrying engine here too (f16x2 measured 5.0e-4)


@pytest.mark.parametrize("backend,tol,tol_rms", [("f16x3", 1e-4, 1e-5), ("f16x2", 5e-4, 1e-4)])
def test_c3_shard_50_step_loop(golden_dir, backend, tol, tol_rms):
    """Round 6 ("there is no 50-step loop golden at the C3 shape"): the FULL 50-step schedule - the one the north star's
    1e-3 gate is stated on - at C3's resolution (1024x1024 -> 128x128 latents, 65 536 values under the dynamic threshold), one image of the shard
    (CFG batch [2,4,128,128]; the images of a shard are independent): the reference's create_model + SpacedDiffusion.p_sample_loop_progressive with
    injected noise (oracle/make_golden.py --only c3loop50; ~2 h of CPU), latent after step 25 and FINAL latent.  Both gate-carrying engines must
    hold the gate itself here - and the 5e-4 this repository asserts for f16x2 at C2: measured f16x2 3.0e-4 / 5.0e-5 rms final (3.6e-4 after step 25),
    f16x3 2.8e-6 / 6.3e-7."""
    fx = _load(golden_dir, "c3_loop50")
    assert fx["steps"] == 50 and fx["lat"] == 128 and fx["B"] == 2
    _, traj = _loop_case(fx, backend)
    for n in sorted(fx["traj"].keys()):
        ma, rms = _dist(traj[n], fx["traj"][n])
        print(f"c3_loop50 {backend}: latent after step {n}: max|d| {ma:.3e} rms {rms:.3e}")
        _record("c3_loop50", f"{backend}_step{n}", max_abs=ma, rms=rms)
        assert ma <= 2 * tol and rms <= tol_rms      # mid-loop latents carry the heavy tail the clamp has not yet removed (see the 10-step test)
    ma, rms = _dist(traj["final"], fx["final"])
    print(f"c3_loop50 {backend}: FINAL latent (50 steps): max|d| {ma:.3e} rms {rms:.3e}")
    _record("c3_loop50", f"{backend}_final", max_abs=ma, rms=rms)
    assert ma <= tol and rms <= tol_rms
    assert ma <= 1e-3


def test_bf16_bits_do_not_depend_on_the_tuner(golden_dir):
    """Fixed-seed reproducibility of the product dtype: at a shape covered by the shipped tile table, an engine that may
    measure tile configurations (autotune on) and one that may not (autotune off) run the SAME configurations and give
    bit-identical outputs - and nothing is timed on the device for that shape."""
    from kandinsky2_amd import _lib
    fx = _load(golden_dir, "c2_text2img")
    L = _lib.lib()
    if L.k22_tile_table_size() == 0:
        pytest.skip("no tile table shipped / loaded")
    arch, sd = _state_dict(False)
    B, lat, bs = fx["B"], fx["lat"], fx["bs"]
    full, pooled, image = k22.make_conditioning(arch, B, seed=2)
    g = torch.Generator().manual_seed(42)
    x_T = torch.randn(B, 4, lat, lat, generator=g)
    x = torch.cat([x_T[:bs], x_T[:bs]], 0).cuda()
    kw = dict(full_emb=full.cuda(), pooled_emb=pooled.cuda(), image_emb=image.cuda())
    outs, reports = [], []
    measured_before = L.k22_tile_table_measured()
    for autotune in ("1", "0"):
        os.environ["K22_AUTOTUNE"] = autotune
        try:
            m = k22.Text2ImUNetHIP(arch, backend_dtype=torch.bfloat16, use_graph=True)
            m.load_state_dict(sd)
            m = m.to("cuda").eval()
            outs.append(m(x, fx["first_ts"].float().cuda(), **kw).cpu())
            reports.append(m.tuning_report())
            del m
        finally:
            os.environ.pop("K22_AUTOTUNE", None)
    assert L.k22_tile_table_measured() == measured_before, "the C2 shape must be fully covered by the shipped tile table"
    assert reports[0] == reports[1]
    assert torch.equal(outs[0], outs[1])
