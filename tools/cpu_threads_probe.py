"""How many host threads should the CPU-oracle baseline use?  Times one 256x256 (latent 32x32) oracle step at several
thread counts (run on the GPU box; CPU only)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kandinsky2_amd as k22
from oracle import unet_ref
arch = k22.make_arch(k22.MODEL_CONFIG_2_1)
sd = k22.init_unet_state_dict(arch, seed=0)
B, lat = 2, 32
full, pooled, image = k22.make_conditioning(arch, B, seed=2)
x = torch.randn(B, 4, lat, lat)
print("cores", os.cpu_count())
for nt in (16, 32, 64, 128, 256):
    if nt > (os.cpu_count() or 1):
        continue
    torch.set_num_threads(nt)
    t0 = time.perf_counter()
    unet_ref.unet_forward(sd, arch, x, torch.full((B,), 500.0), full, pooled, image)
    print(nt, "threads:", round(time.perf_counter() - t0, 2), "s per 32x32-latent forward", flush=True)
