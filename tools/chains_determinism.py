"""Two-chain execution: bit-reproducibility of repeated forwards + steps/s, per engine type (developer tool)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kandinsky2_amd as k22
from kandinsky2_amd import _lib
names = sys.argv[1].split(",")
DT = {"f16x3": k22.F16X3, "fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}
for inpaint, B in ((False, 2), (True, 8)):
    arch = k22.make_arch(k22.MODEL_CONFIG_2_1, inpainting=inpaint)
    sd = k22.init_unet_state_dict(arch, seed=0)
    full, pooled, image = k22.make_conditioning(arch, B, seed=2)
    kw = dict(full_emb=full.cuda(), pooled_emb=pooled.cuda(), image_emb=image.cuda())
    if inpaint:
        kw.update(inpaint_image=torch.randn(B, 4, 96, 96, device="cuda"), inpaint_mask=torch.ones(B, 1, 96, 96, device="cuda"))
    x = torch.randn(B, 4, 96, 96, device="cuda"); t = torch.full((B,), 500.0, device="cuda")
    for n in names:
        m = k22.Text2ImUNetHIP(arch, backend_dtype=DT[n], use_graph=True)
        m.load_state_dict(sd); m = m.to("cuda"); m.prepare(free_params=True)
        outs = [m(x, t, **kw).clone() for _ in range(24)]
        torch.cuda.synchronize()
        bad = sum(1 for o in outs[1:] if not torch.equal(o, outs[0]))
        t0 = time.perf_counter()
        for _ in range(20): m(x, t, **kw)
        torch.cuda.synchronize()
        print(f"DET chains={os.environ.get('K22_CHAINS', '1')} {n} B={B} inpaint={inpaint}: {bad} of 23 repeats differ; {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per forward; "
              f"measured tile configs {_lib.lib().k22_tile_table_measured()}", flush=True)
        del m; torch.cuda.empty_cache()
