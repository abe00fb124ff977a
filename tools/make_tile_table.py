#!/usr/bin/env python
"""Measures the conv / GEMM tile configurations of the shapes BASELINE.json's configs run (and the small shapes smoke() and
the parity tests use) on THIS MI355X and writes the tile table the package ships (kandinsky-2_amd/tiles_gfx950.txt).

    K22_TILE_TABLE=0 python tools/make_tile_table.py gpurun_out/tiles_gfx950.txt

Run it after a kernel change that alters which configuration is fastest (or that changes the candidate set), copy the file
to kandinsky-2_amd/tiles_gfx950.txt and commit it: engines then resolve those problems from the table without timing
anything, which is what makes the bf16 bits of a fixed seed the same on every box (csrc/tuning.h).
"""
import os
import sys
import time

ENC_ONLY = "--encoders-only" in sys.argv     # keep the shipped table and add only the conditioning-tower shapes to it
SMALL_CONV = "--small-conv" in sys.argv      # keep the shipped table except the 16-bit 3x3 convolutions with M <= 1152 rows: re-measured here
                                             # (the weight-streaming kernel joined their candidate list)
X3_ONLY = "--x3-only" in sys.argv           # keep the shipped table and add only the split-precision (K22_F16X3) lines
FUSED_SKIP = "--fused-skip" in sys.argv     # keep the shipped table except the 3x3 convolutions WITH A FUSED 1x1 SKIP (key column 6 >= 100000): re-measured
                                            # (round 5: their second K loop became an NSK-deep LDS-DMA ring - the specialised kernels may win them now)
X2_ONLY = "--x2-only" in sys.argv           # keep the shipped table and add the asymmetric split's (K22_F16X2) own lines: measured with every
                                            # convolution of the plan at two MFMAs (K22_X2_PLAN=3) and without the fall-back to the x3 lines
if X2_ONLY:
    os.environ["K22_X2_OWN_LINES"] = "1"
    os.environ["K22_X2_PLAN"] = "3"
ADD_MISSING = "--add-missing" in sys.argv or "--fused-skip" in sys.argv   # keep the shipped table; run every UNet shape list (all engine types): whatever problem is not in the
                                            # table yet is measured and added (round 4: the half-batch problems of the two-chain execution)
if FUSED_SKIP:
    _here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _src = os.environ.get("K22_TILE_TABLE") or os.path.join(_here, "kandinsky-2_amd", "tiles_gfx950.txt")
    _tmp = "/tmp/k22_tiles_without_fused_skip.txt"
    with open(_src) as f, open(_tmp, "w") as g:
        kept = dropped = 0
        for line in f:
            v = line.split()
            if line.startswith("#") or len(v) < 7 or not (v[1] == "9" and int(v[5]) >= 100000):
                g.write(line); kept += 1
            else:
                dropped += 1
    print(f"--fused-skip: {kept} table lines kept, {dropped} dropped for re-measurement")
    os.environ["K22_TILE_TABLE"] = _tmp
elif SMALL_CONV:
    _here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _src = os.path.join(_here, "kandinsky-2_amd", "tiles_gfx950.txt")
    _tmp = "/tmp/k22_tiles_without_small_conv.txt"
    with open(_src) as f, open(_tmp, "w") as g:
        kept = dropped = 0
        for line in f:
            v = line.split()
            if line.startswith("#") or len(v) < 4 or not (v[0] == "0" and v[1] == "9" and int(v[2]) <= 1152):
                g.write(line); kept += 1
            else:
                dropped += 1
    print(f"--small-conv: {kept} table lines kept, {dropped} dropped for re-measurement")
    os.environ["K22_TILE_TABLE"] = _tmp
elif not ENC_ONLY and not X3_ONLY and not ADD_MISSING and not X2_ONLY and not FUSED_SKIP:
    os.environ["K22_TILE_TABLE"] = "0"      # start empty: everything below is measured here
os.environ.setdefault("K22_TUNE_REPS", "7")
os.environ.pop("K22_TUNE_CACHE", None)
os.environ["K22_AUTOTUNE"] = "1"

import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kandinsky2_amd as k22  # noqa: E402
from kandinsky2_amd import _lib  # noqa: E402


def unet(cfg, inpainting, shapes, dtypes=(torch.bfloat16, torch.float32)):
    arch = k22.make_arch(cfg, inpainting=inpainting)
    sd = k22.init_unet_state_dict(arch, seed=0)
    for dt in dtypes:
        m = k22.Text2ImUNetHIP(arch, backend_dtype=dt, use_graph=False)
        m.load_state_dict(sd)
        m = m.to("cuda")
        m.prepare(free_params=True)
        for (B, h, w) in shapes:
            t0 = time.time()
            full, pooled, image = k22.make_conditioning(arch, B, seed=2)
            kw = dict(full_emb=full.cuda(), pooled_emb=pooled.cuda(), image_emb=image.cuda())
            if inpainting:
                kw.update(inpaint_image=torch.zeros(B, 4, h, w, device="cuda"), inpaint_mask=torch.zeros(B, 1, h, w, device="cuda"))
            m.del_cache()
            m(torch.randn(B, 4, h, w, device="cuda"), torch.full((B,), 500.0, device="cuda"), **kw)
            torch.cuda.synchronize()
            print(f"unet {'inpaint' if inpainting else 'text2img'} C={arch.model_channels} {dt} B={B} {h}x{w}: "
                  f"{time.time() - t0:.1f} s, table = {_lib.lib().k22_tile_table_size()} entries", flush=True)
        del m
        torch.cuda.empty_cache()


def unet22(cfg, controlnet, shapes, dtypes=(torch.bfloat16, torch.float32)):
    arch = k22.make_arch22(cfg, controlnet=controlnet)
    sd = k22.init_unet22_state_dict(arch, seed=0)
    for dt in dtypes:
        m = k22.UNet2DConditionHIP(arch, backend_dtype=dt, use_graph=False)
        m.load_state_dict(sd)
        m = m.to("cuda")
        m.prepare(free_params=True)
        for (B, h, w) in shapes:
            ack = {"image_embeds": torch.randn(B, arch.image_dim, device="cuda")}
            if controlnet:
                ack["hint"] = torch.rand(B, 3, 8 * h, 8 * w, device="cuda")
            m.del_cache()
            m(torch.randn(B, 4, h, w, device="cuda"), 500, added_cond_kwargs=ack)
            torch.cuda.synchronize()
            print(f"unet22 {'controlnet ' if controlnet else ''}C={arch.model_channels} {dt} B={B} {h}x{w}: table = {_lib.lib().k22_tile_table_size()} entries", flush=True)
        del m
        torch.cuda.empty_cache()


def prior(hp, batches, dtypes=(torch.bfloat16, torch.float32)):
    sd = k22.init_prior_state_dict(hp, seed=0)
    for dt in dtypes:
        m = k22.PriorDiffusionModelHIP(hp, backend_dtype=dt)
        m.load_state_dict(sd)
        m = m.to("cuda")
        for N in batches:
            x, t = torch.randn(N, 768, device="cuda"), torch.full((N,), 500.0, device="cuda")
            te, tq = torch.randn(N, 768, device="cuda"), torch.randn(N, 77, 768, device="cuda")
            m.transformer(x, t, te, tq, torch.ones(N, 77, dtype=torch.bool, device="cuda"))
            torch.cuda.synchronize()
            print(f"prior width {hp['xf_width']} {dt} N={N}: table = {_lib.lib().k22_tile_table_size()} entries", flush=True)
        del m
        torch.cuda.empty_cache()


def encoders():
    """conditioning towers: production sizes at 2 / 4 / 8 sequences (bs 1, 2, 4) and 1 / 2 images, and the parity-test shapes"""
    cases = [(k22.CLIP_VITL14, dict(k22.XLMR_LARGE, vocab_size=4096), 1024, 768, (2, 4, 8), (1, 2), (torch.bfloat16, torch.float32)),
             (k22.tiny_clip_config(), k22.tiny_xlmr_config(), 128, 64, (1, 3, 4, 8), (1, 3), (torch.bfloat16, torch.float32))]
    for ccfg, xcfg, inf, outf, seqs, imgs, dtypes in cases:
        csd, xsd = k22.init_clip_state_dict(ccfg, seed=0), k22.init_multiclip_state_dict(xcfg, inf, outf, seed=0)
        for dt in dtypes:
            clip = k22.CLIPModelHIP(ccfg, backend_dtype=dt)
            clip.load_state_dict(csd)
            clip = clip.to("cuda")
            xl = k22.MultilingualCLIPHIP(xcfg, in_features=inf, out_features=outf, backend_dtype=dt)
            xl.load_state_dict(xsd)
            xl = xl.to("cuda")
            for n in seqs:
                tok = torch.zeros(n, 77, dtype=torch.long, device="cuda")
                tok[:, 0], tok[:, 1] = ccfg["vocab_size"] - 2, ccfg["vocab_size"] - 1
                clip.encode_text(tok)
                ids = torch.ones(n, 77, dtype=torch.long, device="cuda")
                ids[:, 0], ids[:, 1] = 0, 2
                xl(ids, ids.ne(1).long())
            r = ccfg["image_resolution"]
            for n in imgs:
                clip.encode_image(torch.zeros(n, 3, r, r, device="cuda"))
            torch.cuda.synchronize()
            print(f"encoders width {ccfg['transformer_width']}/{xcfg['hidden_size']} {dt}: table = {_lib.lib().k22_tile_table_size()} entries", flush=True)
            del clip, xl
            torch.cuda.empty_cache()


def vision_hf():
    """Kandinsky 2.2's image encoder (CLIPVisionModelWithProjection, CLIP ViT-bigG/14) at 1 / 2 images, and the parity-test tower"""
    for cfg, imgs, dtypes in ((k22.CLIP_BIGG_VISION, (1, 2), (torch.bfloat16, torch.float32)),
                              (k22.tiny_clip_vision_hf_config(), (1, 3), (torch.bfloat16, torch.float32))):
        sd = k22.init_clip_vision_hf_state_dict(cfg, seed=0)
        for dt in dtypes:
            m = k22.CLIPVisionModelWithProjectionHIP(cfg, backend_dtype=dt)
            m.load_state_dict(sd)
            m = m.to("cuda")
            for n in imgs:
                m(torch.zeros(n, 3, cfg["image_size"], cfg["image_size"], device="cuda"))
            torch.cuda.synchronize()
            print(f"clip vision (hf keys) width {cfg['hidden_size']} {dt}: table = {_lib.lib().k22_tile_table_size()} entries", flush=True)
            del m
            torch.cuda.empty_cache()


def main():
    out = next((a for a in sys.argv[1:] if not a.startswith("--")), "gpurun_out/tiles_gfx950.txt")
    quick = "--quick" in sys.argv
    if ENC_ONLY:
        n0 = _lib.lib().k22_tile_table_size()
        encoders()
        vision_hf()
        n = _lib.lib().k22_tile_table_save(out.encode())
        print(f"{n0} shipped + {n - n0} new = {n} entries -> {out}")
        return
    tiny = k22.tiny_model_config()
    if ADD_MISSING:
        n0 = _lib.lib().k22_tile_table_size()
        every = (torch.bfloat16, torch.float32, k22.F16X3)
        unet(tiny, False, [(2, 16, 16), (4, 8, 24), (4, 16, 16)], dtypes=every)
        unet(tiny, True, [(4, 16, 24), (2, 16, 16), (4, 16, 16)], dtypes=every)
        t22 = k22.tiny_unet22_config()
        unet22(t22, False, [(4, 16, 24), (4, 16, 16)])
        unet22(t22, True, [(4, 16, 24), (4, 16, 16)])
        unet(k22.MODEL_CONFIG_2_1, False, [(2, 32, 32), (2, 96, 96), (2, 64, 64), (4, 96, 96), (8, 96, 96), (2, 128, 128), (8, 128, 128)], dtypes=(torch.bfloat16,))
        unet(k22.MODEL_CONFIG_2_1, False, [(2, 32, 32), (2, 96, 96), (8, 128, 128)], dtypes=(torch.float32, k22.F16X3))
        unet(k22.MODEL_CONFIG_2_1, False, [(1, 96, 96), (1, 32, 32)], dtypes=(torch.bfloat16, torch.float32))   # the half-batch chains of Text2ImUNetHIP(chains=2)
        unet(tiny, False, [(1, 16, 16), (2, 8, 24)], dtypes=every + (k22.F16X2,))
        unet(tiny, True, [(2, 16, 24), (1, 16, 16)], dtypes=every + (k22.F16X2,))
        unet(k22.MODEL_CONFIG_2_1, True, [(8, 96, 96), (2, 96, 96)], dtypes=(torch.bfloat16,))
        unet(k22.MODEL_CONFIG_2_1, True, [(8, 96, 96)], dtypes=(torch.float32, k22.F16X3))
        unet22(k22.UNET_CONFIG_2_2, False, [(2, 96, 96), (2, 32, 32)], dtypes=(torch.bfloat16,))
        unet22(k22.UNET_CONFIG_2_2, False, [(2, 32, 32)], dtypes=(torch.float32,))
        unet22(k22.UNET_CONFIG_2_2, True, [(4, 96, 96)], dtypes=(torch.bfloat16,))
        n = _lib.lib().k22_tile_table_save(out.encode())
        print(f"{n0} shipped + {n - n0} new = {n} entries -> {out}")
        return
    if X2_ONLY:
        x2 = (k22.F16X2,)
        n0 = _lib.lib().k22_tile_table_size()
        unet(tiny, False, [(2, 16, 16), (4, 8, 24), (4, 16, 16)], dtypes=x2)
        unet(tiny, True, [(4, 16, 24), (2, 16, 16), (4, 16, 16)], dtypes=x2)
        unet(k22.MODEL_CONFIG_2_1, False, [(2, 96, 96), (2, 32, 32), (8, 128, 128), (1, 96, 96)], dtypes=x2)   # (1, ...): the half-batch chains
        unet(k22.MODEL_CONFIG_2_1, True, [(8, 96, 96)], dtypes=x2)
        n = _lib.lib().k22_tile_table_save(out.encode())
        print(f"{n0} shipped + {n - n0} new = {n} entries -> {out}")
        return
    if X3_ONLY:
        x3 = (k22.F16X3,)
        n0 = _lib.lib().k22_tile_table_size()
        unet(tiny, False, [(2, 16, 16), (4, 8, 24), (4, 16, 16)], dtypes=x3)
        unet(tiny, True, [(4, 16, 24), (2, 16, 16), (4, 16, 16)], dtypes=x3)
        unet(k22.MODEL_CONFIG_2_1, False, [(2, 32, 32), (2, 96, 96), (8, 128, 128), (1, 96, 96), (4, 96, 96)], dtypes=x3)
        unet(k22.MODEL_CONFIG_2_1, True, [(8, 96, 96)], dtypes=x3)
        n = _lib.lib().k22_tile_table_save(out.encode())
        print(f"{n0} shipped + {n - n0} new = {n} entries -> {out}")
        return
    if SMALL_CONV:
        b16 = (torch.bfloat16,)
        unet(tiny, False, [(2, 16, 16), (4, 8, 24), (4, 16, 16)], dtypes=b16)
        unet(tiny, True, [(4, 16, 24), (2, 16, 16), (4, 16, 16)], dtypes=b16)
        t22 = k22.tiny_unet22_config()
        unet22(t22, False, [(4, 16, 24), (4, 16, 16)], dtypes=b16)
        unet22(t22, True, [(4, 16, 24), (4, 16, 16)], dtypes=b16)
        unet(k22.MODEL_CONFIG_2_1, False, [(2, 32, 32), (2, 96, 96), (2, 64, 64), (4, 96, 96), (8, 96, 96), (2, 128, 128), (8, 128, 128)], dtypes=b16)
        unet(k22.MODEL_CONFIG_2_1, True, [(8, 96, 96), (2, 96, 96)], dtypes=b16)
        unet22(k22.UNET_CONFIG_2_2, False, [(2, 96, 96), (2, 32, 32)], dtypes=b16)
        unet22(k22.UNET_CONFIG_2_2, True, [(4, 96, 96)], dtypes=b16)
        n = _lib.lib().k22_tile_table_save(out.encode())
        print(f"{n} entries -> {out}")
        return
    # smoke() / parity-test shapes of the 1/3-width model
    unet(tiny, False, [(2, 16, 16), (4, 8, 24), (4, 16, 16)])
    unet(tiny, True, [(4, 16, 24), (2, 16, 16), (4, 16, 16)])
    t22 = k22.tiny_unet22_config()
    unet22(t22, False, [(4, 16, 24), (4, 16, 16)])
    unet22(t22, True, [(4, 16, 24), (4, 16, 16)])
    # BASELINE.json configs: C1 (256^2 bs 1), C2 (768^2 bs 1: the bench line), 512^2, C3 per-GPU (1024^2 bs 4), 768^2 bs 2 / bs 4
    if quick:
        unet(k22.MODEL_CONFIG_2_1, False, [(2, 96, 96)], dtypes=(torch.bfloat16,))
    else:
        # bf16 (product path): every configured shape; fp32 (parity path): the shapes the parity tests run
        unet(k22.MODEL_CONFIG_2_1, False, [(2, 32, 32), (2, 96, 96), (2, 64, 64), (4, 96, 96), (8, 96, 96), (2, 128, 128), (8, 128, 128)],
             dtypes=(torch.bfloat16,))
        unet(k22.MODEL_CONFIG_2_1, False, [(2, 32, 32), (2, 96, 96), (8, 128, 128)], dtypes=(torch.float32,))
        unet(k22.MODEL_CONFIG_2_1, True, [(8, 96, 96), (2, 96, 96)], dtypes=(torch.bfloat16,))   # C4 (inpainting 768^2 bs 4) and bs 1
        unet(k22.MODEL_CONFIG_2_1, True, [(8, 96, 96)], dtypes=(torch.float32,))
        # Kandinsky 2.2 head (32 context tokens): C2 shape, C5 (ControlNet-depth 768^2 bs 2), and the full-width parity shape
        unet22(k22.UNET_CONFIG_2_2, False, [(2, 96, 96), (2, 32, 32)], dtypes=(torch.bfloat16,))
        unet22(k22.UNET_CONFIG_2_2, False, [(2, 32, 32)], dtypes=(torch.float32,))
        unet22(k22.UNET_CONFIG_2_2, True, [(4, 96, 96)], dtypes=(torch.bfloat16,))
        prior(k22.tiny_prior_hparams(), [4])
        prior(k22.PRIOR_HPARAMS_2_1, [2, 4, 8], dtypes=(torch.bfloat16,))
        prior(k22.PRIOR_HPARAMS_2_1, [4], dtypes=(torch.float32,))
        encoders()
        vision_hf()
    n = _lib.lib().k22_tile_table_save(out.encode())
    print(f"{n} entries -> {out}")


if __name__ == "__main__":
    main()
