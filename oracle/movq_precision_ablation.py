"""TEST INFRASTRUCTURE / ANALYSIS ONLY - which part of the MoVQ decoder has to stay in fp32 for the uint8 image to stay within one
grey level of the reference?  Emulates the engine's 16-bit storage points on the CPU oracle (oracle/movq_ref.py): weights rounded once,
every convolution output (after its residual add) and every SpatialNorm(+swish) output rounded to the storage type, fp32 accumulation -
for the blocks BEFORE a cut index; the blocks from the cut on (and norm_out / conv_out) run in fp32.

    python oracle/movq_precision_ablation.py [--lat 32] [--dtype fp16|bf16]

Prints, per cut, the float error and the uint8 image statistics against the all-fp32 decode.  Not used by any test or product path.
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kandinsky2_amd as k22  # noqa: E402
from oracle import movq_ref as R  # noqa: E402


def decode(sd, sd_low, arch, quant, cut, rnd):
    """blocks [0, cut) in the low-precision emulation, [cut, end] in fp32; cut = len(blocks) + 1 puts norm_out / conv_out low too"""
    zq = quant
    blocks, _ = arch.blocks()
    low = cut > 0
    w = sd_low if low else sd
    r = rnd if low else (lambda t: t)
    h = r(R._conv(w, "decoder.conv_in", r(R._conv(w, "post_quant_conv", quant)), 1))
    for i, (kind, pfx, cin, cout) in enumerate(blocks):
        low = i < cut
        w = sd_low if low else sd
        r = rnd if low else (lambda t: t)
        if kind == "res":
            a = r(R._swish(R._snorm(w, pfx + ".norm1", h, zq)))
            a = r(R._conv(w, pfx + ".conv1", a, 1))
            a = r(R._swish(R._snorm(w, pfx + ".norm2", a, zq)))
            x = r(R._conv(w, pfx + ".nin_shortcut", h)) if cin != cout else h
            h = r(x + R._conv(w, pfx + ".conv2", a, 1))
        elif kind == "attn":
            n = r(R._snorm(w, pfx + ".norm", h, zq))
            q, k, v = r(R._conv(w, pfx + ".q", n)), r(R._conv(w, pfx + ".k", n)), r(R._conv(w, pfx + ".v", n))
            b, c, hh, ww = q.shape
            s = torch.bmm(q.reshape(b, c, -1).permute(0, 2, 1), k.reshape(b, c, -1)) * (int(c) ** (-0.5))
            p = r(F.softmax(s, dim=2))
            o = r(torch.bmm(v.reshape(b, c, -1), p.permute(0, 2, 1)).reshape(b, c, hh, ww))
            h = r(h + R._conv(w, pfx + ".proj_out", o))
        else:
            h = r(R._conv(w, pfx, F.interpolate(h, scale_factor=2.0, mode="nearest"), 1))
    low = cut > len(blocks)
    w = sd_low if low else sd
    r = rnd if low else (lambda t: t)
    h = r(R._swish(R._snorm(w, "decoder.norm_out", h, zq)))
    return R._conv(w, "decoder.conv_out", h, 1)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--lat", type=int, default=32)
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    T = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    rnd = lambda t: t.to(T).float()  # noqa: E731
    cfg = k22.MOVQ_CONFIG_2_1
    arch = k22.MoVQArch(cfg["ddconfig"], cfg["embed_dim"])
    sd = k22.init_movq_state_dict(arch, seed=0)
    sd_low = {k: (rnd(v) if v.dim() >= 2 and not k.endswith(("conv_y.weight", "conv_b.weight")) else v) for k, v in sd.items()}
    g = torch.Generator().manual_seed(5)
    z = torch.randn(1, 4, a.lat, a.lat, generator=g)
    blocks, _ = arch.blocks()
    with torch.no_grad():
        ref = R.movq_decode(sd, arch, z)
        u_ref = R.process_images_u8(ref).int()
        print(f"{a.dtype} storage emulation, {8 * a.lat} px; all-fp32 absmax {ref.abs().max():.3f}")
        print("cut (blocks below it are 16-bit)            | max|d| of scale | uint8 max diff | % bytes differ")
        for cut, label in [(len(blocks) + 1, "everything 16-bit"), (len(blocks), "norm_out + conv_out fp32"), (18, "+ level 0 (768 px at C2) fp32"),
                           (14, "+ level 1 fp32"), (10, "+ level 2 fp32"), (0, "all fp32")]:
            out = decode(sd, sd_low, arch, z, cut, rnd)
            du = (R.process_images_u8(out).int() - u_ref).abs()
            print(f"{cut:3d} {label:38s} | {((out - ref).abs().max() / ref.abs().max()).item():.3e} | {du.max().item():3d} | {(du > 0).float().mean().item() * 100:6.2f}")
