"""Debug: does any kernel of a forward write outside the workspace the engine asked for?  Guard regions before / after the workspace."""
import os, sys, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["K22_CHAINS"] = "1"
import kandinsky2_amd as k22
from kandinsky2_amd import _lib
L = _lib.lib()
G = 512 << 20
for name, dt in (("f16x3", k22.F16X3), ("bf16", torch.bfloat16), ("fp32", torch.float32)):
    for inpaint, B in ((True, 4), (False, 1), (False, 2)):
        arch = k22.make_arch(k22.MODEL_CONFIG_2_1, inpainting=inpaint)
        sd = k22.init_unet_state_dict(arch, seed=0)
        m = k22.Text2ImUNetHIP(arch, backend_dtype=dt, use_graph=False)
        m.load_state_dict(sd); m = m.to("cuda"); m.prepare(free_params=True)
        lat = 96
        nbytes = C.c_size_t()
        _lib.check(L.k22_unet_plan(m._handle, B, lat, lat, C.byref(nbytes)))
        buf = torch.full((G + nbytes.value + 512 + G,), 0xAB, dtype=torch.uint8, device="cuda")
        base = (buf.data_ptr() + G + 255) // 256 * 256
        _lib.check(L.k22_unet_bind(m._handle, base, nbytes.value))
        m._ws = buf; m._plan_key = (B, lat, lat); m._cond_key = None
        full, pooled, image = k22.make_conditioning(arch, B, seed=2)
        kw = dict(full_emb=full.cuda(), pooled_emb=pooled.cuda(), image_emb=image.cuda())
        if inpaint:
            kw.update(inpaint_image=torch.randn(B, 4, lat, lat, device="cuda"), inpaint_mask=torch.ones(B, 1, lat, lat, device="cuda"))
        x = torch.randn(B, 4, lat, lat, device="cuda")
        for _ in range(2):
            m(x, torch.full((B,), 500.0, device="cuda"), **kw)
        torch.cuda.synchronize()
        off0 = base - buf.data_ptr()
        pre = buf[:off0]; post = buf[off0 + nbytes.value:]
        bad_pre = (pre != 0xAB).nonzero().flatten(); bad_post = (post != 0xAB).nonzero().flatten()
        print(f"GUARD {name} inpaint={inpaint} B={B}: ws {nbytes.value} bytes; touched before: {bad_pre.numel()} (first {bad_pre[:3].tolist()}), after: {bad_post.numel()}"
              f" (first offsets past the end {bad_post[:4].tolist()}, last {bad_post[-2:].tolist() if bad_post.numel() else []})", flush=True)
        del m, buf
        torch.cuda.empty_cache()
