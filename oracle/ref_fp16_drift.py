"""Yardstick for "what reduced-precision storage costs": the REFERENCE's own fp16 mode against its own fp32 mode.

    python oracle/ref_fp16_drift.py [--lat 32] [--steps 10] [--dtype fp16|bf16] [--out file.json]   (build container only: imports /root/reference)

Runs Kandinsky 2.1's reference UNet (create_model, 1.23 B parameters, seeded weights) through the verbatim model_fn +
SpacedDiffusion.p_sample_loop_progressive (kandinsky2_1_model.py:222-257, gaussian_diffusion.py:426-475) twice on the same
injected noise: once with use_fp16=False and once with use_fp16=True + convert_to_fp16() (what Kandinsky2_1.__init__ does for
the shipped configuration, kandinsky2_1_model.py:92-97: fp16 conv / attention weights and activations, fp32 GroupNorm, softmax,
time embedding and sampler state - SURVEY Appendix A), on the CPU.  Prints max-abs / rms distance of the first UNet output and
of the latent after every step, and writes tests/golden/ref_fp16_drift.json.  The bf16 engine's measured distance from the
fp32 reference (tests/test_full_size_gpu.py) is judged against these numbers in DESIGN.md section 3: fp16 has 3 more
mantissa bits than bf16, so the reference's own drift x 8 is the scale bf16 storage is expected to cost.
"""
import argparse
import copy
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import kandinsky2_amd as k22  # noqa: E402
from oracle import ref_loader  # noqa: E402


def build(use_fp16, sd, low=torch.float16):
    """use_fp16=True: the reference's reduced-precision mode.  low=torch.bfloat16 runs the SAME mode with bf16 in place of fp16: the
    modules convert_to_fp16() casts (unet.py:566-572, text2im_model2_1.py:49-55) are re-cast from the fp32 weights to bf16 and
    model.dtype - the activation type of the torso (unet.py:602) - is set to bf16; GroupNorm / softmax / time embedding / sampler
    stay fp32 exactly as in the fp16 mode.  That is the yardstick for a bf16 engine at the benchmarked shape (VERDICT r2 #1a)."""
    mc = ref_loader.ref("model.model_creation")
    cfg = dict(copy.deepcopy(k22.MODEL_CONFIG_2_1), up=False, inpainting=False, use_fp16=use_fp16)
    m = mc.create_model(**cfg).eval()
    m.load_state_dict(sd, strict=True)
    if use_fp16:
        m.convert_to_fp16()
        if low != torch.float16:
            for name, p in m.named_parameters():
                if p.dtype == torch.float16:
                    p.data = sd[name].to(low)
            m.dtype = low
    return m


def run(model, lat, steps, guidance=4.0):
    gd = ref_loader.ref("model.gaussian_diffusion")
    mc = ref_loader.ref("model.model_creation")
    arch = k22.make_arch(k22.MODEL_CONFIG_2_1)
    full, pooled, image = k22.make_conditioning(arch, 2, seed=2)
    g = torch.Generator().manual_seed(42)
    x_T = torch.randn(2, 4, lat, lat, generator=g)
    noise_seq = torch.randn(steps, 2, 4, lat, lat, generator=g)
    dt = model.dtype
    kw = dict(full_emb=full.to(dt), pooled_emb=pooled.to(dt), image_emb=image.to(dt))
    diff = mc.create_gaussian_diffusion(**dict(k22.DIFFUSION_CONFIG_2_1, timestep_respacing=str(steps)))
    first = {}

    def model_fn(x_t, ts, **kwargs):
        half = x_t[: len(x_t) // 2]
        combined = torch.cat([half, half], dim=0)
        model_out = model(combined, ts, **kwargs)
        first.setdefault("out", model_out.float().clone())
        eps, rest = model_out[:, :4], model_out[:, 4:]
        cond_eps, uncond_eps = torch.split(eps, len(eps) // 2, dim=0)
        half_eps = uncond_eps + guidance * (cond_eps - uncond_eps)
        eps = torch.cat([half_eps, half_eps], dim=0)
        return torch.cat([eps, rest], dim=1)

    it = iter(list(noise_seq))
    orig = gd.th.randn_like
    gd.th.randn_like = lambda t: next(it).to(t)
    traj = []
    try:
        model.del_cache()
        for out in diff.p_sample_loop_progressive(model_fn, tuple(x_T.shape), device="cpu", noise=x_T.clone(), progress=False,
                                                  model_kwargs=kw, init_step=None, denoised_fn=lambda x: x.clamp(-2, 2)):
            traj.append(out["sample"].float().clone())
            print(f"  step {len(traj)}/{steps}", flush=True)
        model.del_cache()
    finally:
        gd.th.randn_like = orig
    return first["out"], traj


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--lat", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--out", default="")
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    if a.threads:
        torch.set_num_threads(a.threads)
    low = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    arch = k22.make_arch(k22.MODEL_CONFIG_2_1)
    sd = k22.init_unet_state_dict(arch, seed=0)
    with torch.no_grad():
        f32_first, f32_traj = run(build(False, sd), a.lat, a.steps)
        f16_first, f16_traj = run(build(True, sd, low), a.lat, a.steps)
    scale = f32_first.abs().max().item()
    rep = {"config": f"reference Text2ImUNet 1.23B, CFG batch 2x4x{a.lat}x{a.lat}, {a.steps}-step p_sampler, CPU, seeded weights; reduced-precision "
                     f"mode = use_fp16 + convert_to_fp16() with {a.dtype} storage, against the reference's own fp32 mode",
           "first_forward": {"max_abs": (f16_first - f32_first).abs().max().item(), "scale": scale,
                             "rel": (f16_first - f32_first).abs().max().item() / scale}, "steps": {}}
    print(f"first forward: {a.dtype}-vs-fp32 max|d| {rep['first_forward']['max_abs']:.3e} = {rep['first_forward']['rel']:.3e} of scale {scale:.2f}")
    for n, (p, q) in enumerate(zip(f16_traj, f32_traj), 1):
        d = p - q
        rep["steps"][str(n)] = {"max_abs": d.abs().max().item(), "rms": d.pow(2).mean().sqrt().item()}
        print(f"latent after step {n:2d}: {a.dtype}-vs-fp32 max|d| {d.abs().max().item():.3e} rms {d.pow(2).mean().sqrt().item():.3e}")
    with open(a.out or os.path.join(ROOT, "tests", "golden", "ref_fp16_drift.json"), "w") as f:
        json.dump(rep, f, indent=1)
