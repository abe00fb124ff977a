"""state_dict -> packed weight arena for the HIP engine.

Layouts (T = engine dtype: bf16, fp16 or fp32; for the split-precision engine "f16x3" the MFMA weights - conv3x3, conv1x1 / conv1d /
linear, qkv, encoder_kv - are x3 chunks (to_x3 below: fp16 hi / lo pairs of the fp32 weight x 2^8, 4 bytes per element) and the
emb_layers GEMV weights fp32; everything else fp32):
  conv3x3  [O,I,3,3] -> T [roundup(O,64)][ky][kx][I]   (K = tap-major then channel, contiguous)
  conv1x1 / conv1d k=1 / linear feeding the MFMA GEMM -> T [roundup(O,64)][I]
  AttentionBlock.qkv  rows re-ordered  head*192 + {q,k,v}*64 + d  ->  {q,k,v}*C + head*64 + d
  AttentionBlock.encoder_kv rows       head*128 + {k,v}*64 + d    ->  {k,v}*C + head*64 + d
     (the reference keeps per-head [q|k|v] interleaved, unet.py:296-300; the engine wants Q, K, V planes)
  all ResBlock emb_layers.1 concatenated in module order -> T [sum 2*Cout][time_embed_dim] (one GEMV/step)
  stem conv [O,I,3,3] -> fp32 [I*9][O]
  GroupNorm / LayerNorm affine, biases, time_embed.*, clip_to_seq, proj_n, img_layer: fp32 as is
  time_freqs: exp(-ln(10000) * arange(half)/half) exactly as nn.py:113-117 builds it.
The arena is ONE contiguous uint8 tensor (256-byte aligned entries): rank 0 packs, the other ranks
receive it with a single RCCL broadcast (parallel.py) and rebuild the same offset table from the shapes.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Tuple

import torch

from .arch import UNetArch


def _pad_rows(w: torch.Tensor, mult: int = 64) -> torch.Tensor:
    o = w.shape[0]
    op = (o + mult - 1) // mult * mult
    if op == o:
        return w
    pad = torch.zeros((op - o,) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
    return torch.cat([w, pad], 0)


def to_x3(w: torch.Tensor, scale: float = 256.0) -> torch.Tensor:
    """fp32 [..., K] (K % 8 == 0) -> the split-precision operand format of the K22_F16X3 arithmetic (csrc/common.h, "x3 chunks"):
    every group of 8 consecutive K elements becomes [hi0..hi7 | lo0..lo7] in fp16 with hi = rne(x * scale), lo = rne(x * scale - hi);
    returned as a float32-typed tensor of the SAME shape (4 bytes per element, bit pattern = the fp16 pairs)."""
    if w.shape[-1] % 8:
        raise ValueError("to_x3: the K dimension must be a multiple of 8")
    ws = w.float() * scale
    hi = ws.to(torch.float16)
    lo = (ws - hi.float()).to(torch.float16)
    sh = tuple(w.shape)
    g = sh[:-1] + (sh[-1] // 8, 8)
    return torch.stack([hi.reshape(g), lo.reshape(g)], dim=-2).reshape(sh[:-1] + (sh[-1] * 2,)).contiguous().view(torch.float32)


def packed_entries(arch: UNetArch, sd: Dict[str, torch.Tensor], tdtype, device) -> "OrderedDict[str, torch.Tensor]":
    f32 = torch.float32
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    head22 = arch.head == "2.2"
    if head22:
        # Kandinsky 2.2 UNet (diffusers keys): same blocks, another conditioning head; to_q / to_k / to_v are already planes
        from .unet22 import sd22_to_internal
        sd = sd22_to_internal(arch, sd)

    def get(name):
        return sd[name].detach().to(device=device, dtype=f32)

    x3 = tdtype in ("f16x3", "f16x2")   # split-precision engines: MFMA weights in x3 chunks, everything else fp32
    vdtype = f32 if x3 else tdtype   # weights read by the vector (non-MFMA) kernels

    def cast(w):
        return to_x3(w) if x3 else w.to(tdtype).contiguous()

    def conv3(name):
        w = get(name)  # [O,I,3,3]
        return cast(_pad_rows(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)))

    def mat(name):
        w = get(name)
        return cast(_pad_rows(w.reshape(w.shape[0], -1)))

    half = arch.model_channels // 2
    out["time_freqs"] = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=f32) / half).to(device)
    head = (["head22.ctx_proj", "head22.ctx_norm", "head22.emb_proj", "head22.emb_norm"] + [f"hint.{k}" for k in range(8 if arch.hint_channels else 0)]
            if head22 else ["clip_to_seq", "proj_n", "img_layer"])
    for n in ["time_embed.0", "time_embed.2"] + head:
        out[n + ".weight"] = get(n + ".weight").contiguous()
        out[n + ".bias"] = get(n + ".bias").contiguous()
    if not head22:
        out["ln_model_n.weight"] = get("ln_model_n.weight")
        out["ln_model_n.bias"] = get("ln_model_n.bias")
        out["to_model_dim_n.weight"] = mat("to_model_dim_n.weight")
        out["to_model_dim_n.bias"] = get("to_model_dim_n.bias")

    emb_w, emb_b = [], []
    for b in arch.blocks:
        if b[0] == "stem":
            pfx = b[1]
            w = get(pfx + ".weight")               # [O, I, 3, 3] -> [I*9][O]: the stem kernel reads one weight per thread, coalesced over O
            out[pfx + ".weight"] = w.reshape(w.shape[0], -1).t().contiguous()
            out[pfx + ".bias"] = get(pfx + ".bias")
        elif b[0] == "res":
            _, pfx, cin, cout, _ud = b
            for n in (".in_layers.0", ".out_layers.0"):
                out[pfx + n + ".weight"] = get(pfx + n + ".weight")
                out[pfx + n + ".bias"] = get(pfx + n + ".bias")
            for n in (".in_layers.2", ".out_layers.3"):
                out[pfx + n + ".weight"] = conv3(pfx + n + ".weight")
                out[pfx + n + ".bias"] = get(pfx + n + ".bias")
            if cin != cout:
                out[pfx + ".skip_connection.weight"] = mat(pfx + ".skip_connection.weight")
                out[pfx + ".skip_connection.bias"] = get(pfx + ".skip_connection.bias")
            emb_w.append(get(pfx + ".emb_layers.1.weight"))
            emb_b.append(get(pfx + ".emb_layers.1.bias"))
        else:
            _, pfx, c = b
            h = c // arch.num_head_channels
            out[pfx + ".norm.weight"] = get(pfx + ".norm.weight")
            out[pfx + ".norm.bias"] = get(pfx + ".norm.bias")
            if head22:   # [to_q | to_k | to_v] and [add_k_proj | add_v_proj]: head hh is rows hh*64 .. of each plane already
                wq, bq = get(pfx + ".qkv.weight"), get(pfx + ".qkv.bias")
                wk, bk = get(pfx + ".encoder_kv.weight"), get(pfx + ".encoder_kv.bias")
            else:
                wq = get(pfx + ".qkv.weight").reshape(h, 3, 64, c).permute(1, 0, 2, 3).reshape(3 * c, c)
                bq = get(pfx + ".qkv.bias").reshape(h, 3, 64).permute(1, 0, 2).reshape(-1)
                wk = get(pfx + ".encoder_kv.weight").reshape(h, 2, 64, arch.model_dim).permute(1, 0, 2, 3).reshape(2 * c, -1)
                bk = get(pfx + ".encoder_kv.bias").reshape(h, 2, 64).permute(1, 0, 2).reshape(-1)
            out[pfx + ".qkv.weight"] = cast(_pad_rows(wq))
            out[pfx + ".qkv.bias"] = bq.contiguous()
            out[pfx + ".encoder_kv.weight"] = cast(_pad_rows(wk))
            out[pfx + ".encoder_kv.bias"] = bk.contiguous()
            out[pfx + ".proj_out.weight"] = mat(pfx + ".proj_out.weight")
            out[pfx + ".proj_out.bias"] = get(pfx + ".proj_out.bias")
    out["emb_layers.weight"] = torch.cat(emb_w, 0).to(vdtype).contiguous()
    out["emb_layers.bias"] = torch.cat(emb_b, 0).contiguous()
    out["out.0.weight"] = get("out.0.weight")
    out["out.0.bias"] = get("out.0.bias")
    out["out.2.weight"] = conv3("out.2.weight")
    out["out.2.bias"] = get("out.2.bias")
    return out


def pack_arena(arch: UNetArch, sd: Dict[str, torch.Tensor], tdtype, device) -> Tuple[torch.Tensor, "OrderedDict[str, Tuple[int, int]]"]:
    """Returns (arena uint8 tensor on `device`, name -> (byte offset, byte size))."""
    entries = packed_entries(arch, sd, tdtype, device)
    table: "OrderedDict[str, Tuple[int, int]]" = OrderedDict()
    off = 0
    for name, t in entries.items():
        nbytes = t.numel() * t.element_size()
        table[name] = (off, nbytes)
        off += (nbytes + 255) // 256 * 256
    arena = torch.zeros(off + 256, dtype=torch.uint8, device=device)
    for name, t in entries.items():
        o, nbytes = table[name]
        arena[o:o + nbytes] = t.reshape(-1).view(torch.uint8)
    return arena, table
