#!/usr/bin/env python
"""Split-precision (K22_F16X3) engine on the GPU: tiny + C2 first forward against the committed reference goldens, the C2 50-step
p_sampler final latent (BASELINE.json's 1e-3 gate), steps/s and the per-class device time.  Developer tool (tools/gpu_x3.sh)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kandinsky2_amd as k22  # noqa: E402
from kandinsky2_amd import _lib  # noqa: E402

dev = "cuda"
dts = sys.argv[1].split(",") if len(sys.argv) > 1 else ["f16x3"]
DT = {"f16x3": k22.F16X3, "fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}
fx = torch.load(os.path.join(ROOT, "tests", "golden", "c2_text2img.pt"), weights_only=False)
arch = k22.make_arch(k22.MODEL_CONFIG_2_1)
sd = k22.init_unet_state_dict(arch, seed=0)
B, lat, steps, bs = fx["B"], fx["lat"], fx["steps"], fx["bs"]
full, pooled, image = k22.make_conditioning(arch, B, seed=2)
kw = dict(full_emb=full.to(dev), pooled_emb=pooled.to(dev), image_emb=image.to(dev))
g = torch.Generator().manual_seed(42)
x_T = torch.randn(B, 4, lat, lat, generator=g).to(dev)
noise_seq = torch.randn(steps, B, 4, lat, lat, generator=g).to(dev)
d = k22.create_gaussian_diffusion(**dict(k22.DIFFUSION_CONFIG_2_1, timestep_respacing=str(steps)))
for name in dts:
    m = k22.Text2ImUNetHIP(arch, backend_dtype=DT[name], use_graph=True)
    m.load_state_dict(sd)
    m = m.to(dev)
    m.prepare(free_params=True)
    t0 = time.time()
    first = m(torch.cat([x_T[:bs], x_T[:bs]], 0), fx["first_ts"].float().to(dev), **kw)
    torch.cuda.synchronize()
    t_first = time.time() - t0
    d0 = (first.cpu() - fx["first_out"]).abs().max().item()
    rep = {"dtype": name, "first_forward_rel": d0 / fx["first_out"].abs().max().item(), "first_call_s": round(t_first, 1),
           "tile_configs_measured": _lib.lib().k22_tile_table_measured()}
    out = None
    for it in range(2):
        m.del_cache()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = d.p_sample_loop(m, (B, 4, lat, lat), model_kwargs=kw, guidance_scale=fx["guidance"], noise=x_T, noise_seq=noise_seq, whole_loop_graph=True)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    dd = (out.cpu() - fx["final"]).float()
    rep.update(steps_per_s=round(steps / el, 2), final_latent_max_abs=float(f"{dd.abs().max().item():.3e}"),
               final_latent_rms=float(f"{dd.pow(2).mean().sqrt().item():.3e}"))
    prof = m.profile(3)
    rep["by_class_ms"] = {k: round(v["ms"], 4) for k, v in prof.items()}
    rep["launches"] = {k: v["launches"] for k, v in prof.items()}
    print(json.dumps(rep), flush=True)
    if "--tuning" in sys.argv:
        print(m.tuning_report(), flush=True)
    del m
    torch.cuda.empty_cache()
