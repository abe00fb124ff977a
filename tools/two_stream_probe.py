#!/usr/bin/env python
"""Probe (round 4): does running the two images of the CFG pair as two INDEPENDENT B = 1 chains on two streams beat one B = 2 chain?
The forward is a chain of ~400 dependent launches of which ~37 % (GroupNorm, the 20-us GEMMs, split-K finishes) are latency-bound."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kandinsky2_amd as k22

dev = "cuda"
arch = k22.make_arch(k22.MODEL_CONFIG_2_1)
sd = k22.init_unet_state_dict(arch, seed=0)
lat = 96
def mk():
    m = k22.Text2ImUNetHIP(arch, backend_dtype=torch.bfloat16, use_graph=True)
    m.load_state_dict(sd); m = m.to(dev); m.prepare(free_params=True)
    return m
m2 = mk()
full, pooled, image = [t.to(dev) for t in k22.make_conditioning(arch, 2, seed=2)]
x = torch.randn(2, 4, lat, lat, device=dev); t = torch.full((2,), 500.0, device=dev)
def run2(n):
    for _ in range(n): m2(x, t, full_emb=full, pooled_emb=pooled, image_emb=image)
run2(3); torch.cuda.synchronize()
t0 = time.perf_counter(); run2(20); torch.cuda.synchronize(); e2 = (time.perf_counter() - t0) / 20
print(f"one B=2 chain: {e2*1e3:.3f} ms per forward")
ma = k22.Text2ImUNetHIP(arch, backend_dtype=torch.bfloat16, use_graph=True); ma.prepare(arena=m2._arena)
mb = k22.Text2ImUNetHIP(arch, backend_dtype=torch.bfloat16, use_graph=True); mb.prepare(arena=m2._arena)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
xa, xb, ta = x[:1].contiguous(), x[1:].contiguous(), t[:1].contiguous()
kwa = dict(full_emb=full[:1].contiguous(), pooled_emb=pooled[:1].contiguous(), image_emb=image[:1].contiguous())
kwb = dict(full_emb=full[1:].contiguous(), pooled_emb=pooled[1:].contiguous(), image_emb=image[1:].contiguous())
def run1(n, both=True):
    for _ in range(n):
        with torch.cuda.stream(sa): ma(xa, ta, **kwa)
        if both:
            with torch.cuda.stream(sb): mb(xb, ta, **kwb)
run1(3); torch.cuda.synchronize()
t0 = time.perf_counter(); run1(20, both=False); torch.cuda.synchronize(); e1 = (time.perf_counter() - t0) / 20
print(f"one B=1 chain alone: {e1*1e3:.3f} ms per forward   (tile configs measured: {k22._lib.lib().k22_tile_table_measured()})")
t0 = time.perf_counter(); run1(20); torch.cuda.synchronize(); ep = (time.perf_counter() - t0) / 20
print(f"two B=1 chains on two streams: {ep*1e3:.3f} ms per pair   vs one B=2 chain {e2*1e3:.3f} ms")
