"""CPU fp32 restatement of the Kandinsky-2.1 latent UNet forward (TEST INFRASTRUCTURE, see oracle/__init__.py).

Functional form over a reference-keyed state_dict; every function cites the reference lines it follows
(paths relative to /root/reference).  Pinned against the reference modules by oracle/make_golden.py.
"""
import math

import torch
import torch.nn.functional as F


def timestep_embedding(timesteps, dim, max_period=10000):
    """kandinsky2/model/nn.py:101-121"""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def group_norm32(x, w, b, swish):
    """GroupNorm32.forward, kandinsky2/model/nn.py:26-37 (32 groups, eps 1e-5, fp32)."""
    y = F.group_norm(x.float(), 32, w, b, eps=1e-5).to(x.dtype)
    return F.silu(y) if swish else y


def res_block(sd, pfx, x, emb, updown):
    """ResBlock.forward with use_scale_shift_norm, kandinsky2/model/unet.py:193-220."""
    h = group_norm32(x, sd[pfx + ".in_layers.0.weight"], sd[pfx + ".in_layers.0.bias"], True)
    if updown == 1:      # Downsample(use_conv=False) = AvgPool2d(2) on h and x, unet.py:157-164, 105-107
        h, x = F.avg_pool2d(h, 2), F.avg_pool2d(x, 2)
    elif updown == 2:    # Upsample(use_conv=False) = nearest x2, unet.py:67-77
        h, x = F.interpolate(h, scale_factor=2, mode="nearest"), F.interpolate(x, scale_factor=2, mode="nearest")
    h = F.conv2d(h, sd[pfx + ".in_layers.2.weight"], sd[pfx + ".in_layers.2.bias"], padding=1)
    e = F.linear(F.silu(emb), sd[pfx + ".emb_layers.1.weight"], sd[pfx + ".emb_layers.1.bias"])[..., None, None]
    scale, shift = torch.chunk(e, 2, dim=1)
    h = group_norm32(h, sd[pfx + ".out_layers.0.weight"], sd[pfx + ".out_layers.0.bias"], False) * (1 + scale) + shift
    h = F.conv2d(F.silu(h), sd[pfx + ".out_layers.3.weight"], sd[pfx + ".out_layers.3.bias"], padding=1)
    if (pfx + ".skip_connection.weight") in sd:
        x = F.conv2d(x, sd[pfx + ".skip_connection.weight"], sd[pfx + ".skip_connection.bias"])
    return x + h


def qkv_attention(qkv, encoder_kv, n_heads):
    """QKVAttention.forward einsum path, kandinsky2/model/unet.py:286-302, 333-340."""
    bs, width, length = qkv.shape
    ch = width // (3 * n_heads)
    q, k, v = qkv.reshape(bs * n_heads, ch * 3, length).split(ch, dim=1)
    ek, ev = encoder_kv.reshape(bs * n_heads, ch * 2, -1).split(ch, dim=1)
    k = torch.cat([ek, k], dim=-1)
    v = torch.cat([ev, v], dim=-1)
    scale = 1 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    w = torch.softmax(w.float(), dim=-1).type(w.dtype)
    a = torch.einsum("bts,bcs->bct", w, v)
    return a.reshape(bs, -1, length)


def attention_block(sd, pfx, x, encoder_out, head_ch=64):
    """AttentionBlock.forward, kandinsky2/model/unet.py:260-269."""
    b, c, hh, ww = x.shape
    n = group_norm32(x, sd[pfx + ".norm.weight"], sd[pfx + ".norm.bias"], False).view(b, c, -1)
    qkv = F.conv1d(n, sd[pfx + ".qkv.weight"], sd[pfx + ".qkv.bias"])
    ekv = F.conv1d(encoder_out, sd[pfx + ".encoder_kv.weight"], sd[pfx + ".encoder_kv.bias"])
    h = qkv_attention(qkv, ekv, c // head_ch)
    h = F.conv1d(h, sd[pfx + ".proj_out.weight"], sd[pfx + ".proj_out.bias"])
    return x + h.reshape(b, c, hh, ww)


def text_emb(sd, arch, full_emb, pooled_emb, image_emb):
    """Text2ImUNet.get_text_emb, pooling_type='from_model', kandinsky2/model/text2im_model2_1.py:57-80."""
    clip_seq = F.linear(image_emb, sd["clip_to_seq.weight"], sd["clip_to_seq.bias"]).reshape(
        image_emb.shape[0], arch.num_image_embs, arch.model_dim)
    xf_proj = F.linear(pooled_emb, sd["proj_n.weight"], sd["proj_n.bias"])
    xf_proj = F.layer_norm(xf_proj, (xf_proj.shape[-1],), sd["ln_model_n.weight"], sd["ln_model_n.bias"], 1e-5)
    xf_proj = xf_proj + F.linear(image_emb, sd["img_layer.weight"], sd["img_layer.bias"])
    xf_out = torch.cat((clip_seq, F.linear(full_emb, sd["to_model_dim_n.weight"], sd["to_model_dim_n.bias"])), dim=1)
    return xf_proj, xf_out.permute(0, 2, 1)


@torch.no_grad()
def unet_forward(sd, arch, x, timesteps, full_emb, pooled_emb, image_emb, inpaint_image=None, inpaint_mask=None):
    """Text2ImUNet.forward (text2im_model2_1.py:85-103) / InpaintText2ImUNet.forward (:146-155), fp32."""
    if arch.inpainting:
        if inpaint_image is None:
            inpaint_image = torch.zeros_like(x)
        if inpaint_mask is None:
            inpaint_mask = torch.zeros_like(x[:, :1])
        x = torch.cat([x, inpaint_image * inpaint_mask, inpaint_mask], dim=1)
    emb = timestep_embedding(timesteps, arch.model_channels)
    emb = F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    xf_proj, xf_out = text_emb(sd, arch, full_emb, pooled_emb, image_emb)
    emb = emb + xf_proj
    h = x.float()
    hs = []
    blocks = arch.blocks
    i = 0
    # input blocks: one hs entry per TimestepEmbedSequential (unet.py input_blocks)
    n_in = sum(1 for b in blocks if b[1].startswith("input_blocks."))
    cur_seq = None
    for b in blocks[:n_in]:
        seq = b[1].split(".")[1]
        if cur_seq is not None and seq != cur_seq:
            hs.append(h)
        cur_seq = seq
        h = _apply(sd, b, h, emb, xf_out, arch)
    hs.append(h)
    i = n_in
    while blocks[i][1].startswith("middle_block."):
        h = _apply(sd, blocks[i], h, emb, xf_out, arch)
        i += 1
    cur_seq = None
    for b in blocks[i:]:
        seq = b[1].split(".")[1]
        if seq != cur_seq:
            h = torch.cat([h, hs.pop()], dim=1)  # text2im_model2_1.py:99
            cur_seq = seq
        h = _apply(sd, b, h, emb, xf_out, arch)
    h = group_norm32(h, sd["out.0.weight"], sd["out.0.bias"], True)
    return F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)


def _apply(sd, b, h, emb, xf_out, arch):
    if b[0] == "stem":
        return F.conv2d(h, sd[b[1] + ".weight"], sd[b[1] + ".bias"], padding=1)
    if b[0] == "res":
        return res_block(sd, b[1], h, emb, b[4])
    return attention_block(sd, b[1], h, xf_out, arch.num_head_channels)
