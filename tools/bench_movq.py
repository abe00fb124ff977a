#!/usr/bin/env python
"""MoVQ decode time at the BASELINE.json image sizes (once per image; reported next to the UNet step rate)."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kandinsky2_amd as k22

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=768)
ap.add_argument("--bs", type=int, default=1)
ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
arch = k22.MoVQArch(k22.MOVQ_CONFIG_2_1["ddconfig"])
m = k22.MoVQDecoderHIP(backend_dtype={"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[a.dtype])
m.load_state_dict(k22.init_movq_state_dict(arch, seed=0), strict=True)
m = m.to("cuda")
lat = a.size // 8
z = torch.randn(a.bs, 4, lat, lat, device="cuda")
out = m.decode(z)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    out, u8 = m.decode(z, return_uint8=True)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
gflop = {768: 4885.5, 1024: 9647.4}.get(a.size, 4885.5 * (a.size / 768) ** 2) * a.bs
print(f"MoVQ decode {a.size}x{a.size} bs={a.bs} {a.dtype}: {ms:.2f} ms/image-batch  (~{gflop / ms:.0f} TFLOP/s, finite={bool(torch.isfinite(out).all())}, ws={m._ws.numel() / 2**20:.0f} MiB)")

# encoder (img2img / inpainting pre-step): once per call
e = k22.MoVQEncoderHIP(backend_dtype={"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[a.dtype])
e.load_state_dict(k22.init_movq_encoder_state_dict(arch, seed=0), strict=True)
e = e.to("cuda")
img = torch.randn(a.bs, 3, a.size, a.size, device="cuda") * 0.5
lat_out = e.encode(img)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    lat_out = e.encode(img)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print(f"MoVQ encode {a.size}x{a.size} bs={a.bs} {a.dtype}: {ms:.2f} ms/image-batch  (finite={bool(torch.isfinite(lat_out).all())}, ws={e._ws.numel() / 2**20:.0f} MiB)")
