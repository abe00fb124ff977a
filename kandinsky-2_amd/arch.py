"""Architecture walk of the Kandinsky-2.1 latent UNet (Text2ImUNet).

One function enumerates the blocks in module order; the drop-in nn.Module (unet.py), the weight
packer (pack.py), the seeded initialiser (weights.py) and the CPU oracle (oracle/unet_ref.py) all
consume it, and the C++ engine (csrc/engine.hip) repeats the same loops.

Mirrors the constructor loops of the reference (paths relative to /root/reference):
  UNetModel.__init__            kandinsky2/model/unet.py:371-563
  Text2ImUNet.__init__          kandinsky2/model/text2im_model2_1.py:14-47
  create_model (config schema)  kandinsky2/model/model_creation.py:9-83
"""
from __future__ import annotations

import copy
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import List, Tuple

# Same schema as the reference's CONFIG_2_1["model_config"] (kandinsky2/configs.py:125-149).
MODEL_CONFIG_2_1 = {
    "version": "2.1",
    "image_size": 64,
    "num_channels": 384,
    "num_res_blocks": 3,
    "channel_mult": "",
    "num_heads": 1,
    "num_head_channels": 64,
    "num_heads_upsample": -1,
    "attention_resolutions": "32,16,8",
    "dropout": 0,
    "model_dim": 768,
    "use_scale_shift_norm": True,
    "resblock_updown": True,
    "use_fp16": True,
    "cache_text_emb": True,
    "text_encoder_in_dim1": 1024,
    "text_encoder_in_dim2": 768,
    "image_encoder_in_dim": 768,
    "num_image_embs": 10,
    "pooling_type": "from_model",
    "in_channels": 4,
    "out_channels": 8,
    "use_flash_attention": False,
}

# Same schema as CONFIG_2_1["diffusion_config"] (kandinsky2/configs.py:150-162).
DIFFUSION_CONFIG_2_1 = {
    "learn_sigma": True,
    "sigma_small": False,
    "steps": 1000,
    "noise_schedule": "linear",
    "timestep_respacing": "",
    "use_kl": False,
    "predict_xstart": False,
    "rescale_timesteps": True,
    "rescale_learned_sigmas": True,
    "linear_start": 0.00085,
    "linear_end": 0.012,
}


def tiny_model_config() -> dict:
    """Same topology at 1/3 width (128 base channels): used by the CPU-sized parity tests."""
    c = copy.deepcopy(MODEL_CONFIG_2_1)
    c["num_channels"] = 128
    return c


@dataclass
class UNetArch:
    in_channels: int
    model_channels: int
    out_channels: int
    num_res_blocks: int
    channel_mult: Tuple[int, ...]
    attention_ds: Tuple[int, ...]
    num_head_channels: int
    model_dim: int
    text_dim1: int
    text_dim2: int
    image_dim: int
    num_image_embs: int
    text_ctx: int = 77
    inpainting: bool = False
    blocks: List[tuple] = field(default_factory=list)
    head: str = "2.1"          # "2.1": Text2ImUNet.get_text_emb (image + text tokens); "2.2": image-only head (unet22.py)
    hint_channels: int = 0     # 3 for the 2.2 ControlNet-depth UNet (hint conv stack, in_channels 8)

    @property
    def time_embed_dim(self) -> int:
        return 4 * self.model_channels

    @property
    def ctx_len(self) -> int:
        return self.num_image_embs + self.text_ctx


def make_arch(model_config: dict, inpainting: bool = False) -> UNetArch:
    mc = dict(model_config)
    if mc.get("pooling_type", "from_model") != "from_model":
        raise NotImplementedError("only pooling_type='from_model' (the shipped 2.1 config) is implemented")
    if not mc.get("use_scale_shift_norm", True) or not mc.get("resblock_updown", True):
        raise NotImplementedError("only use_scale_shift_norm=True, resblock_updown=True (the shipped 2.1 config)")
    cm = mc.get("channel_mult", "")
    if cm == "":  # model_creation.py:34-43
        cm = {256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}[mc["image_size"]]
    elif isinstance(cm, str):
        cm = tuple(int(v) for v in cm.split(","))
    att = tuple(mc["image_size"] // int(r) for r in mc["attention_resolutions"].split(","))  # model_creation.py:46-48
    in_ch = mc["in_channels"]
    if inpainting:  # InpaintText2ImUNet.__init__ (text2im_model2_1.py:136-144)
        in_ch = in_ch * 2 + 1
    a = UNetArch(
        in_channels=in_ch, model_channels=mc["num_channels"], out_channels=mc["out_channels"],
        num_res_blocks=mc["num_res_blocks"], channel_mult=tuple(cm), attention_ds=att,
        num_head_channels=mc["num_head_channels"], model_dim=mc["model_dim"],
        text_dim1=mc["text_encoder_in_dim1"], text_dim2=mc["text_encoder_in_dim2"],
        image_dim=mc.get("image_encoder_in_dim", 768), num_image_embs=mc.get("num_image_embs", 10),
        inpainting=inpainting,
    )
    a.blocks = _walk(a)
    return a


def _walk(a: UNetArch) -> List[tuple]:
    """[('stem', prefix, cin, cout) | ('res', prefix, cin, cout, updown) | ('attn', prefix, C)] in module
    order.  For output blocks `cin` is the concatenated width (h + skip), as in unet.py:519-532."""
    mc = a.model_channels
    out: List[tuple] = []
    ch = mc * a.channel_mult[0]
    out.append(("stem", "input_blocks.0.0", a.in_channels, ch))
    chans = [ch]
    ds, blk = 1, 1
    for level, mult in enumerate(a.channel_mult):
        for _ in range(a.num_res_blocks):
            out.append(("res", f"input_blocks.{blk}.0", ch, mc * mult, 0))
            ch = mc * mult
            if ds in a.attention_ds:
                out.append(("attn", f"input_blocks.{blk}.1", ch))
            chans.append(ch)
            blk += 1
        if level != len(a.channel_mult) - 1:
            out.append(("res", f"input_blocks.{blk}.0", ch, ch, 1))
            chans.append(ch)
            blk += 1
            ds *= 2
    out.append(("res", "middle_block.0", ch, ch, 0))
    out.append(("attn", "middle_block.1", ch))
    out.append(("res", "middle_block.2", ch, ch, 0))
    blk = 0
    for level, mult in list(enumerate(a.channel_mult))[::-1]:
        for i in range(a.num_res_blocks + 1):
            ich = chans.pop()
            out.append(("res", f"output_blocks.{blk}.0", ch + ich, mc * mult, 0))
            ch = mc * mult
            sub = 1
            if ds in a.attention_ds:
                out.append(("attn", f"output_blocks.{blk}.{sub}", ch))
                sub += 1
            if level and i == a.num_res_blocks:
                out.append(("res", f"output_blocks.{blk}.{sub}", ch, ch, 2))
                ds //= 2
            blk += 1
    return out


def param_shapes(a: UNetArch) -> "OrderedDict[str, tuple]":
    """state_dict key -> shape, identical to the reference Text2ImUNet's (checked against
    tests/golden/ref_unet_keys_*.json)."""
    if a.head == "2.2":   # the diffusers UNet2DConditionModel keys of the Kandinsky 2.2 decoder
        from .unet22 import param_shapes22
        return param_shapes22(a)
    mc, ted = a.model_channels, a.time_embed_dim
    p: "OrderedDict[str, tuple]" = OrderedDict()

    def lin(name, o, i):
        p[name + ".weight"] = (o, i)
        p[name + ".bias"] = (o,)

    lin("time_embed.0", ted, mc)
    lin("time_embed.2", ted, ted)
    for b in a.blocks:
        if b[0] == "stem":
            _, pfx, cin, cout = b
            p[pfx + ".weight"] = (cout, cin, 3, 3)
            p[pfx + ".bias"] = (cout,)
        elif b[0] == "res":
            _, pfx, cin, cout, _ud = b
            p[pfx + ".in_layers.0.weight"] = (cin,)
            p[pfx + ".in_layers.0.bias"] = (cin,)
            p[pfx + ".in_layers.2.weight"] = (cout, cin, 3, 3)
            p[pfx + ".in_layers.2.bias"] = (cout,)
            lin(pfx + ".emb_layers.1", 2 * cout, ted)
            p[pfx + ".out_layers.0.weight"] = (cout,)
            p[pfx + ".out_layers.0.bias"] = (cout,)
            p[pfx + ".out_layers.3.weight"] = (cout, cout, 3, 3)
            p[pfx + ".out_layers.3.bias"] = (cout,)
            if cin != cout:
                p[pfx + ".skip_connection.weight"] = (cout, cin, 1, 1)
                p[pfx + ".skip_connection.bias"] = (cout,)
        else:
            _, pfx, c = b
            p[pfx + ".norm.weight"] = (c,)
            p[pfx + ".norm.bias"] = (c,)
            p[pfx + ".qkv.weight"] = (3 * c, c, 1)
            p[pfx + ".qkv.bias"] = (3 * c,)
            p[pfx + ".encoder_kv.weight"] = (2 * c, a.model_dim, 1)
            p[pfx + ".encoder_kv.bias"] = (2 * c,)
            p[pfx + ".proj_out.weight"] = (c, c, 1)
            p[pfx + ".proj_out.bias"] = (c,)
    c0 = mc * a.channel_mult[0]
    p["out.0.weight"] = (c0,)
    p["out.0.bias"] = (c0,)
    p["out.2.weight"] = (a.out_channels, c0, 3, 3)
    p["out.2.bias"] = (a.out_channels,)
    lin("clip_to_seq", a.model_dim * a.num_image_embs, a.image_dim)
    lin("to_model_dim_n", a.model_dim, a.text_dim1)
    lin("proj_n", ted, a.text_dim2)
    p["ln_model_n.weight"] = (ted,)
    p["ln_model_n.bias"] = (ted,)
    lin("img_layer", ted, a.image_dim)
    return p
