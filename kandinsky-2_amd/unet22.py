"""Kandinsky 2.2 decoder UNet on the HIP engine: the module the reference INJECTS into the diffusers pipelines
(`UNet2DConditionModel.from_pretrained('kandinsky-community/kandinsky-2-2-decoder', subfolder='unet')` ->
`KandinskyV22Pipeline.from_pretrained(..., unet=self.unet)`, kandinsky2/kandinsky2_2_model.py:26-41), the ControlNet-depth
variant (notebooks/kandinsky2_2_controlnet.ipynb: `decoder(..., hint=hint, ...)`), and the `DDPMScheduler` step the pipeline
drives it with.

PARITY UNPINNED.  The 2.2 arithmetic lives in `diffusers`, which is not vendored in the reference tree, not pinned by its
setup.py and not installed here; no checkpoint config is readable offline.  What is built here follows the
diffusers source of the commit the reference's notebook installs (huggingface/diffusers e3d71ad8, July 2023) AS RECALLED -
UNet2DConditionModel with ResnetDownsampleBlock2D / SimpleCrossAttn{Down,Up}Block2D / UNetMidBlock2DSimpleCrossAttn,
AttnAddedKVProcessor, ImageProjection, ImageTimeEmbedding / ImageHintTimeEmbedding, DDPMScheduler(learned_range) - and is
checked against the CPU restatement in oracle/unet22_ref.py, which has the same provenance.  Both say so in their headers;
they must be re-pinned against real diffusers outputs when it is importable.

Structure: block for block the 2.2 decoder UNet IS the in-repo 2.1 UNet (same channel plan 384/768/1152/1536,
3 layers per block, scale-shift ResBlocks that resample inside, head dim 64, attention = self keys/values with the projected
context keys/values PREPENDED, out_channels 8) with another conditioning head:
  * context = 32 tokens LayerNorm(Linear(1280 -> 32*768)(image_embeds))      (encoder_hid_proj, "image_proj")
  * time embedding += LayerNorm(Linear(1280 -> 1536)(image_embeds))          (add_embedding, "image" / "image_hint")
  * no text branch; ControlNet-depth: hint [B,3,8h,8w] -> 8-conv stack -> [B,4,h,w], concatenated to the latent (in_channels 8).
So the engine runs the same kernels; this file maps the diffusers state_dict keys onto the engine's packed layout
(to_q / to_k / to_v are already the Q, K, V planes the engine wants: no per-head de-interleave as for 2.1).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from types import SimpleNamespace
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .arch import UNetArch, _walk
from .unet import Text2ImUNetHIP

# unet/config.json of kandinsky-community/kandinsky-2-2-decoder, the keys this engine reads [recalled, unverified]
UNET_CONFIG_2_2 = {
    "in_channels": 4, "out_channels": 8, "block_out_channels": (384, 768, 1152, 1536), "layers_per_block": 3,
    "attention_head_dim": 64, "cross_attention_dim": 768, "encoder_hid_dim": 1280, "encoder_hid_dim_type": "image_proj",
    "addition_embed_type": "image", "num_image_text_embeds": 32, "norm_num_groups": 32, "norm_eps": 1e-5,
    "resnet_time_scale_shift": "scale_shift", "flip_sin_to_cos": True, "freq_shift": 0,
    "down_block_types": ("ResnetDownsampleBlock2D", "SimpleCrossAttnDownBlock2D", "SimpleCrossAttnDownBlock2D", "SimpleCrossAttnDownBlock2D"),
    "mid_block_type": "UNetMidBlock2DSimpleCrossAttn",
    "up_block_types": ("SimpleCrossAttnUpBlock2D", "SimpleCrossAttnUpBlock2D", "SimpleCrossAttnUpBlock2D", "ResnetUpsampleBlock2D"),
}
HINT_CHANNELS = (3, 16, 16, 32, 32, 96, 96, 256, 4)       # input_hint_block: Conv2d k -> k+1 (3x3, pad 1), SiLU between
HINT_STRIDES = (1, 1, 2, 1, 2, 1, 2, 1)


def tiny_unet22_config() -> dict:
    """Same topology at 1/3 width (CPU-affordable parity tests)."""
    return dict(UNET_CONFIG_2_2, block_out_channels=(128, 256, 384, 512))


# diffusers `UNet2DConditionModel.__init__` defaults (models/unet_2d_condition.py, recalled): what a key absent from unet/config.json
# resolves to.  notebooks/lora_decoder.ipynb:3668-3670 shows the decoder's unet/config.json lacking `addition_time_embed_dim`,
# `transformer_layers_per_block`, `num_attention_heads`: all three default to values the SimpleCrossAttn / ResnetDownsample blocks never
# read (None / 1 / None -> heads = channels // attention_head_dim), so that log does not contradict the architecture built here.
UNET2D_DEFAULTS = {
    "in_channels": 4, "out_channels": 4, "center_input_sample": False, "flip_sin_to_cos": True, "freq_shift": 0,
    "down_block_types": ("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    "mid_block_type": "UNetMidBlock2DCrossAttn", "up_block_types": ("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    "only_cross_attention": False, "block_out_channels": (320, 640, 1280, 1280), "layers_per_block": 2, "downsample_padding": 1,
    "mid_block_scale_factor": 1, "act_fn": "silu", "norm_num_groups": 32, "norm_eps": 1e-5, "cross_attention_dim": 1280,
    "transformer_layers_per_block": 1, "encoder_hid_dim": None, "encoder_hid_dim_type": None, "attention_head_dim": 8, "num_attention_heads": None,
    "dual_cross_attention": False, "use_linear_projection": False, "class_embed_type": None, "addition_embed_type": None,
    "addition_time_embed_dim": None, "num_class_embeds": None, "upcast_attention": False, "resnet_time_scale_shift": "default",
    "resnet_skip_time_act": False, "resnet_out_scale_factor": 1.0, "time_embedding_type": "positional", "time_embedding_dim": None,
    "time_embedding_act_fn": None, "timestep_post_act": None, "time_cond_proj_dim": None, "conv_in_kernel": 3, "conv_out_kernel": 3,
    "projection_class_embeddings_input_dim": None, "class_embeddings_concat": False, "mid_block_only_cross_attention": None,
    "cross_attention_norm": None, "addition_embed_type_num_heads": 64, "num_image_text_embeds": 32,
}
_ENGINE_REQUIRES = {   # values of a resolved unet/config.json the HIP engine is built for (anything else raises, nothing is guessed)
    "act_fn": ("silu", "swish"), "norm_num_groups": (32,), "resnet_time_scale_shift": ("scale_shift",), "encoder_hid_dim_type": ("image_proj",),
    "addition_embed_type": ("image", "image_hint"), "mid_block_type": ("UNetMidBlock2DSimpleCrossAttn",), "only_cross_attention": (False,),
    "class_embed_type": (None,), "time_embedding_type": ("positional",), "flip_sin_to_cos": (True,), "freq_shift": (0,), "conv_in_kernel": (3,),
    "conv_out_kernel": (3,), "attention_head_dim": (64,), "mid_block_scale_factor": (1, 1.0), "resnet_out_scale_factor": (1, 1.0),
    "resnet_skip_time_act": (False,), "dual_cross_attention": (False,), "center_input_sample": (False,), "time_cond_proj_dim": (None,),
    "timestep_post_act": (None,), "time_embedding_act_fn": (None,), "time_embedding_dim": (None,), "cross_attention_norm": (None,),
    "mid_block_only_cross_attention": (None, False),
}


def resolve_unet22_config(config: Optional[dict] = None) -> dict:
    """unet/config.json (or None = the recalled UNET_CONFIG_2_2) -> the full key set with diffusers' defaults for absent keys."""
    given = {k: v for k, v in dict(config or UNET_CONFIG_2_2).items() if not k.startswith("_")}
    c = dict(UNET2D_DEFAULTS)
    c.update(given)
    c["_missing_keys"] = tuple(sorted(k for k in UNET2D_DEFAULTS if k not in given))
    return c


def make_arch22(config: Optional[dict] = None, controlnet: Optional[bool] = None, inpainting: Optional[bool] = None) -> UNetArch:
    """Architecture of the injected UNet2DConditionModel (kandinsky2_2_model.py:26-41) from its unet/config.json content: absent keys
    take diffusers' defaults, every value the engine was not built for raises.  controlnet / inpainting default to what the config
    says (addition_embed_type == "image_hint" / in_channels == 9); passing them overrides the config's in_channels as round 2 did."""
    c = resolve_unet22_config(config)
    for key, ok in _ENGINE_REQUIRES.items():
        v = c[key]
        if (tuple(v) if isinstance(v, list) else v) not in ok:
            raise NotImplementedError(f"unet config {key}={v!r}: the HIP engine is built for {ok} (the Kandinsky 2.2 decoder UNets)")
    if abs(float(c["norm_eps"]) - 1e-5) > 1e-12:
        raise NotImplementedError("unet config norm_eps: the engine's GroupNorm uses 1e-5")
    boc = tuple(c["block_out_channels"])
    mc = boc[0]
    if any(v % mc for v in boc):
        raise NotImplementedError("block_out_channels must be multiples of the first")
    types, ups = tuple(c["down_block_types"]), tuple(c["up_block_types"])
    mirror = {"ResnetDownsampleBlock2D": "ResnetUpsampleBlock2D", "SimpleCrossAttnDownBlock2D": "SimpleCrossAttnUpBlock2D"}
    if any(t not in mirror for t in types) or len(types) != len(boc) or ups != tuple(mirror[t] for t in reversed(types)):
        raise NotImplementedError("down / up block types: ResnetDownsampleBlock2D / SimpleCrossAttnDownBlock2D and their mirrored up blocks")
    if controlnet is None:
        controlnet = c["addition_embed_type"] == "image_hint"
    if inpainting is None:
        inpainting = (not controlnet) and c["in_channels"] == 9
    if controlnet and inpainting:
        raise NotImplementedError("controlnet + inpainting")
    if config is not None and "in_channels" in config and (controlnet, inpainting) == (False, False) and c["in_channels"] != 4:
        raise NotImplementedError(f"in_channels={c['in_channels']}: 4 (text2img), 8 (ControlNet-depth: latent + hint) or 9 (inpainting)")
    att = tuple(2 ** i for i, t in enumerate(types) if "CrossAttn" in t)
    a = UNetArch(
        in_channels=8 if controlnet else (9 if inpainting else 4), model_channels=mc, out_channels=c["out_channels"],
        num_res_blocks=c["layers_per_block"], channel_mult=tuple(v // mc for v in boc), attention_ds=att,
        num_head_channels=c["attention_head_dim"], model_dim=c["cross_attention_dim"], text_dim1=1, text_dim2=1,
        image_dim=c["encoder_hid_dim"], num_image_embs=c.get("num_image_text_embeds", 32), text_ctx=0, inpainting=inpainting,
        head="2.2", hint_channels=3 if controlnet else 0)
    a.blocks = _walk(a)
    return a


def names22(a: UNetArch) -> "OrderedDict[str, str]":
    """engine (2.1-style) block prefix -> diffusers module prefix, in module order."""
    L, n = len(a.channel_mult), a.num_res_blocks
    out: "OrderedDict[str, str]" = OrderedDict()
    out["input_blocks.0.0"] = "conv_in"
    blk = 1
    for lvl in range(L):
        for i in range(n):
            out[f"input_blocks.{blk}.0"] = f"down_blocks.{lvl}.resnets.{i}"
            out[f"input_blocks.{blk}.1"] = f"down_blocks.{lvl}.attentions.{i}"
            blk += 1
        if lvl != L - 1:
            out[f"input_blocks.{blk}.0"] = f"down_blocks.{lvl}.downsamplers.0"
            blk += 1
    out["middle_block.0"], out["middle_block.1"], out["middle_block.2"] = "mid_block.resnets.0", "mid_block.attentions.0", "mid_block.resnets.1"
    blk = 0
    ds = 2 ** (L - 1)
    for u, lvl in enumerate(range(L - 1, -1, -1)):
        for i in range(n + 1):
            out[f"output_blocks.{blk}.0"] = f"up_blocks.{u}.resnets.{i}"
            sub = 1
            if ds in a.attention_ds:
                out[f"output_blocks.{blk}.1"] = f"up_blocks.{u}.attentions.{i}"
                sub = 2
            if lvl and i == n:
                out[f"output_blocks.{blk}.{sub}"] = f"up_blocks.{u}.upsamplers.0"
            blk += 1
        if lvl:
            ds //= 2
    return out


_RES = {"in_layers.0": "norm1", "in_layers.2": "conv1", "emb_layers.1": "time_emb_proj", "out_layers.0": "norm2", "out_layers.3": "conv2",
        "skip_connection": "conv_shortcut"}


def param_shapes22(a: UNetArch) -> "OrderedDict[str, tuple]":
    """diffusers UNet2DConditionModel state_dict key -> shape for this architecture."""
    mc, ted, nm = a.model_channels, a.time_embed_dim, names22(a)
    p: "OrderedDict[str, tuple]" = OrderedDict()

    def lin(name, o, i):
        p[name + ".weight"] = (o, i)
        p[name + ".bias"] = (o,)

    def norm(name, c):
        p[name + ".weight"] = (c,)
        p[name + ".bias"] = (c,)

    lin("time_embedding.linear_1", ted, mc)
    lin("time_embedding.linear_2", ted, ted)
    lin("encoder_hid_proj.image_embeds", a.num_image_embs * a.model_dim, a.image_dim)
    norm("encoder_hid_proj.norm", a.model_dim)
    lin("add_embedding.image_proj", ted, a.image_dim)
    norm("add_embedding.image_norm", ted)
    if a.hint_channels:
        for k in range(8):
            p[f"add_embedding.input_hint_block.{2 * k}.weight"] = (HINT_CHANNELS[k + 1], HINT_CHANNELS[k], 3, 3)
            p[f"add_embedding.input_hint_block.{2 * k}.bias"] = (HINT_CHANNELS[k + 1],)
    for b in a.blocks:
        d = nm[b[1]]
        if b[0] == "stem":
            p[d + ".weight"] = (b[3], b[2], 3, 3)
            p[d + ".bias"] = (b[3],)
        elif b[0] == "res":
            _, _pfx, cin, cout, _ud = b
            norm(d + ".norm1", cin)
            p[d + ".conv1.weight"] = (cout, cin, 3, 3)
            p[d + ".conv1.bias"] = (cout,)
            lin(d + ".time_emb_proj", 2 * cout, ted)
            norm(d + ".norm2", cout)
            p[d + ".conv2.weight"] = (cout, cout, 3, 3)
            p[d + ".conv2.bias"] = (cout,)
            if cin != cout:
                p[d + ".conv_shortcut.weight"] = (cout, cin, 1, 1)
                p[d + ".conv_shortcut.bias"] = (cout,)
        else:
            c = b[2]
            norm(d + ".group_norm", c)
            for q in ("to_q", "to_k", "to_v"):
                lin(d + "." + q, c, c)
            lin(d + ".add_k_proj", c, a.model_dim)
            lin(d + ".add_v_proj", c, a.model_dim)
            lin(d + ".to_out.0", c, c)
    c0 = mc * a.channel_mult[0]
    norm("conv_norm_out", c0)
    p["conv_out.weight"] = (a.out_channels, c0, 3, 3)
    p["conv_out.bias"] = (a.out_channels,)
    return p


def sd22_to_internal(a: UNetArch, sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """diffusers-keyed state dict -> the engine packer's names (pack.packed_entries with qkv_planes=True): views / cats only."""
    nm = names22(a)
    out: Dict[str, torch.Tensor] = {}
    for k in ("weight", "bias"):
        out[f"time_embed.0.{k}"] = sd[f"time_embedding.linear_1.{k}"]
        out[f"time_embed.2.{k}"] = sd[f"time_embedding.linear_2.{k}"]
        out[f"head22.ctx_proj.{k}"] = sd[f"encoder_hid_proj.image_embeds.{k}"]
        out[f"head22.ctx_norm.{k}"] = sd[f"encoder_hid_proj.norm.{k}"]
        out[f"head22.emb_proj.{k}"] = sd[f"add_embedding.image_proj.{k}"]
        out[f"head22.emb_norm.{k}"] = sd[f"add_embedding.image_norm.{k}"]
        out[f"out.0.{k}"] = sd[f"conv_norm_out.{k}"]
        out[f"out.2.{k}"] = sd[f"conv_out.{k}"]
        if a.hint_channels:
            for i in range(8):
                out[f"hint.{i}.{k}"] = sd[f"add_embedding.input_hint_block.{2 * i}.{k}"]
    for b in a.blocks:
        pfx, d = b[1], nm[b[1]]
        if b[0] == "stem":
            out[pfx + ".weight"], out[pfx + ".bias"] = sd[d + ".weight"], sd[d + ".bias"]
        elif b[0] == "res":
            for ik, dk in _RES.items():
                if dk == "conv_shortcut" and b[2] == b[3]:
                    continue
                out[f"{pfx}.{ik}.weight"], out[f"{pfx}.{ik}.bias"] = sd[f"{d}.{dk}.weight"], sd[f"{d}.{dk}.bias"]
        else:
            out[pfx + ".norm.weight"], out[pfx + ".norm.bias"] = sd[d + ".group_norm.weight"], sd[d + ".group_norm.bias"]
            out[pfx + ".qkv.weight"] = torch.cat([sd[d + ".to_q.weight"], sd[d + ".to_k.weight"], sd[d + ".to_v.weight"]], 0)
            out[pfx + ".qkv.bias"] = torch.cat([sd[d + ".to_q.bias"], sd[d + ".to_k.bias"], sd[d + ".to_v.bias"]], 0)
            out[pfx + ".encoder_kv.weight"] = torch.cat([sd[d + ".add_k_proj.weight"], sd[d + ".add_v_proj.weight"]], 0)
            out[pfx + ".encoder_kv.bias"] = torch.cat([sd[d + ".add_k_proj.bias"], sd[d + ".add_v_proj.bias"]], 0)
            out[pfx + ".proj_out.weight"], out[pfx + ".proj_out.bias"] = sd[d + ".to_out.0.weight"], sd[d + ".to_out.0.bias"]
    return out


def init_unet22_state_dict(a: UNetArch, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Seeded random weights under the diffusers keys (no checkpoint is reachable offline)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in param_shapes22(a).items():
        leaf = name.rsplit(".", 1)[1]
        if any(t in name for t in (".norm1.", ".norm2.", "group_norm.", "conv_norm_out.", "encoder_hid_proj.norm.", "image_norm.")):
            t = torch.randn(shape, generator=g) * 0.1 + (1.0 if leaf == "weight" else 0.0)
        elif leaf == "bias":
            t = torch.randn(shape, generator=g) * 0.02
        else:
            fan_in = int(np.prod(shape[1:]))
            t = torch.randn(shape, generator=g) * (1.0 / math.sqrt(fan_in))
        sd[name] = t
    return sd


class UNet2DConditionHIP(Text2ImUNetHIP):
    """Drop-in for the `unet=` the reference passes to the diffusers Kandinsky 2.2 pipelines: diffusers state_dict keys, and the
    call the pipelines make:
        unet(sample=[2bs,C,h,w], timestep=t, encoder_hidden_states=None, added_cond_kwargs={"image_embeds": [2bs,1280]
             (, "hint": [2bs,3,8h,8w])}, return_dict=False)[0] -> [2bs,8,h,w]
    plus .config.in_channels, .dtype, .device, .to().  The conditioning (and the hint latent) is cached until del_cache(), or
    re-computed when a different image_embeds / hint tensor object is passed."""

    def __init__(self, arch: Optional[UNetArch] = None, backend_dtype: torch.dtype = torch.bfloat16, use_graph: bool = True,
                 meta_params: bool = False):
        super().__init__(arch or make_arch22(), backend_dtype=backend_dtype, use_graph=use_graph, cache_text_emb=True, meta_params=meta_params)
        a = self.arch
        self.config = SimpleNamespace(in_channels=a.in_channels, out_channels=a.out_channels, cross_attention_dim=a.model_dim,
                                      encoder_hid_dim=a.image_dim, addition_embed_type="image_hint" if a.hint_channels else "image")
        self._cond_src = self._hint_src = None

    @property
    def device(self):
        return self._arena.device if self._arena is not None else next(self.parameters()).device

    def del_cache(self):
        super().del_cache()
        self._cond_src = self._hint_src = None

    def set_condition(self, image_embeds, hint=None):
        B, a = self._plan_key[0], self.arch
        if tuple(image_embeds.shape) != (B, a.image_dim):
            raise ValueError(f"image_embeds must have shape {(B, a.image_dim)}; got {tuple(image_embeds.shape)}")
        if image_embeds.device.type != "cuda":
            raise RuntimeError("image_embeds must be on the GPU (no CPU fallback)")
        L = _lib.lib()
        e = image_embeds.detach().float().contiguous()
        _lib.check(L.k22_unet_set_condition(self._handle, None, None, e.data_ptr(), _lib.current_stream()))
        if a.hint_channels:
            H, W = self._plan_key[1], self._plan_key[2]
            if hint is None or tuple(hint.shape) != (B, a.hint_channels, 8 * H, 8 * W):
                raise ValueError(f"hint must have shape {(B, a.hint_channels, 8 * H, 8 * W)}")
            h = hint.detach().float().contiguous()
            _lib.check(L.k22_unet_set_hint(self._handle, h.data_ptr(), _lib.current_stream()))
        elif hint is not None:
            raise ValueError("this UNet has no hint input (build it with make_arch22(controlnet=True))")
        return self

    @torch.no_grad()
    def forward(self, sample, timestep, encoder_hidden_states=None, added_cond_kwargs=None, return_dict: bool = True, **_unused):
        if sample.device.type != "cuda":
            raise RuntimeError("UNet2DConditionHIP.forward: input must be on the GPU (no CPU fallback)")
        a = self.arch
        B, Cx, H, W = sample.shape
        want_c = 9 if a.inpainting else 4
        if Cx != want_c:
            raise ValueError(f"expected a {want_c}-channel sample" + (" (latents, masked-image latents, mask)" if a.inpainting else ""))
        if encoder_hidden_states is not None:
            raise ValueError("the Kandinsky 2.2 UNet takes encoder_hidden_states=None (image-only conditioning through added_cond_kwargs)")
        self._ensure_plan(B, H, W)
        ack = added_cond_kwargs or {}
        emb, hint = ack.get("image_embeds"), ack.get("hint")
        # the conditioning head is re-run whenever the tensors CHANGE, not only when they are other objects: diffusers-style callers
        # update buffers in place (key = storage pointer, version counter, shape)
        ident = lambda t: None if t is None else (t.data_ptr(), t._version, tuple(t.shape))  # noqa: E731
        if self._cond_key is None or (emb is not None and ident(emb) != self._cond_src) or (hint is not None and ident(hint) != self._hint_src):
            if emb is None:
                raise ValueError("added_cond_kwargs['image_embeds'] is required")
            self.set_condition(emb, hint)
            self._cond_key, self._cond_src, self._hint_src = True, ident(emb), ident(hint)
            self.cache = {"cached": True}
        t = torch.as_tensor(timestep, device=sample.device).float().reshape(-1)
        if t.numel() == 1:
            t = t.expand(B)
        if t.numel() != B:
            raise ValueError(f"timestep must be a scalar or hold one value per batch element ({B})")
        xf = sample.detach().float()
        img = msk = None
        if a.inpainting:
            img, msk = xf[:, 4:8].contiguous(), xf[:, 8:9].contiguous()
        xf = xf[:, :4].contiguous()
        out = torch.empty(B, a.out_channels, H, W, dtype=torch.float32, device=sample.device)
        _lib.check(_lib.lib().k22_unet_forward(self._handle, xf.data_ptr(), t.contiguous().data_ptr(), _lib.ptr(img), _lib.ptr(msk),
                                               out.data_ptr(), 1 if self.use_graph else 0, _lib.current_stream()))
        return SimpleNamespace(sample=out) if return_dict else (out,)


# diffusers `DDPMScheduler.__init__` defaults (schedulers/scheduling_ddpm.py of the commit the reference's notebook installs, recalled):
# what a key ABSENT from scheduler_config.json resolves to.
DDPM_SCHEDULER_DEFAULTS = {
    "num_train_timesteps": 1000, "beta_start": 0.0001, "beta_end": 0.02, "beta_schedule": "linear", "trained_betas": None,
    "variance_type": "fixed_small", "clip_sample": True, "prediction_type": "epsilon", "thresholding": False,
    "dynamic_thresholding_ratio": 0.995, "clip_sample_range": 1.0, "sample_max_value": 1.0, "timestep_spacing": "leading", "steps_offset": 0,
}
# scheduler/scheduler_config.json of kandinsky-community/kandinsky-2-2-decoder, the keys this build assumes it holds [RECALLED, UNVERIFIED;
# no checkpoint is reachable offline].  The one piece of in-tree evidence is notebooks/lora_decoder.ipynb:3661-3662: the 317-byte file
# loaded into DDPMScheduler reports `variance_type`, `clip_sample_range`, `timestep_spacing`, `trained_betas`, `sample_max_value`,
# `dynamic_thresholding_ratio` "not found in config ... initialized to default values" - so the decoder steps with FIXED_SMALL variance
# (the pipeline then drops the UNet's four variance channels), "leading" spacing, and a clip range of 1.0 if it clips at all.  Round 2
# hard-coded learned_range + clip +-2 (the 2.1 sampler's behaviour), which contradicts that log; it stays selectable
# (SCHEDULER_CONFIG_2_2_LEARNED_RANGE) because nothing here can be pinned.  `clip_sample` IS in the file but its value is not
# shown; False is what diffusers' own Kandinsky 2.2 test fixtures use.
SCHEDULER_CONFIG_2_2 = {"_class_name": "DDPMScheduler", "num_train_timesteps": 1000, "beta_schedule": "linear", "beta_start": 0.00085,
                        "beta_end": 0.012, "clip_sample": False, "prediction_type": "epsilon", "thresholding": False}
SCHEDULER_CONFIG_2_2_LEARNED_RANGE = dict(SCHEDULER_CONFIG_2_2, variance_type="learned_range", clip_sample=True, clip_sample_range=2.0)


class DDPMSchedulerHIP:
    """diffusers `DDPMScheduler` (schedulers/scheduling_ddpm.py, recalled; PARITY UNPINNED) driven by its config: constructor
    keywords and defaults are diffusers' own (DDPM_SCHEDULER_DEFAULTS), `from_config(dict)` takes the content of a
    scheduler_config.json and lists the keys it did not find - nothing about the checkpoint is hard-coded.  Built: linear /
    scaled_linear / squaredcos_cap_v2 betas, epsilon prediction, variance_type fixed_small / fixed_small_log / fixed_large /
    fixed_large_log / learned_range, clip_sample with clip_sample_range, leading / linspace / trailing spacing, steps_offset.
    Not built (raises): thresholding=True, v / sample prediction, variance_type "learned".
    step() runs k22_sampler_step (guidance, x0, clip, posterior mean, variance, ancestral noise in one launch); for the fixed
    variance types both log-variance bounds of the step table are set to the fixed value, so the UNet's variance channels cancel out
    of frac * max_log + (1 - frac) * min_log exactly as the pipeline's "drop the variance channels" rule prescribes.  For spaced
    timesteps the per-step quantities (alpha_prod_t / alpha_prod_t_prev, current_beta_t) are exactly improved-DDPM respacing
    (respace.py:83-97) over the retained timesteps, so the same step table serves both samplers."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", trained_betas=None,
                 variance_type="fixed_small", clip_sample=True, prediction_type="epsilon", thresholding=False,
                 dynamic_thresholding_ratio=0.995, clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading", steps_offset=0):
        if prediction_type != "epsilon":
            raise NotImplementedError("DDPMSchedulerHIP: prediction_type 'epsilon' only (the Kandinsky decoders)")
        if thresholding:
            raise NotImplementedError("DDPMSchedulerHIP: thresholding=True (per-sample dynamic thresholding) is not built")
        if variance_type not in ("fixed_small", "fixed_small_log", "fixed_large", "fixed_large_log", "learned_range"):
            raise NotImplementedError(f"DDPMSchedulerHIP: variance_type {variance_type!r}")
        if timestep_spacing not in ("leading", "linspace", "trailing"):
            raise ValueError(f"timestep_spacing {timestep_spacing!r}")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, trained_betas=trained_betas, variance_type=variance_type,
                                      clip_sample=bool(clip_sample), prediction_type=prediction_type, thresholding=False,
                                      dynamic_thresholding_ratio=dynamic_thresholding_ratio, clip_sample_range=float(clip_sample_range),
                                      sample_max_value=sample_max_value, timestep_spacing=timestep_spacing, steps_offset=int(steps_offset))
        self.num_train_timesteps = num_train_timesteps
        if trained_betas is not None:
            self.betas = np.asarray(trained_betas, dtype=np.float64)
        elif beta_schedule == "linear":
            self.betas = np.linspace(beta_start, beta_end, num_train_timesteps, dtype=np.float64)
        elif beta_schedule == "scaled_linear":
            self.betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float64) ** 2
        elif beta_schedule == "squaredcos_cap_v2":
            f = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2  # noqa: E731
            self.betas = np.array([min(1 - f((i + 1) / num_train_timesteps) / f(i / num_train_timesteps), 0.999) for i in range(num_train_timesteps)])
        else:
            raise NotImplementedError(f"beta_schedule {beta_schedule!r}")
        self.alphas_cumprod = np.cumprod(1.0 - self.betas)
        self.clip = float(clip_sample_range) if clip_sample else float("inf")
        self.init_noise_sigma = 1.0
        self.timesteps = None
        self.missing_keys: tuple = ()

    @classmethod
    def from_config(cls, config: dict):
        """`DDPMScheduler.from_config(scheduler_config.json)`: keys of the file override diffusers' defaults; like diffusers, the
        keys the file does not hold are reported (self.missing_keys) - they are what "initialized to default values" names."""
        known = {k: v for k, v in dict(config).items() if k in DDPM_SCHEDULER_DEFAULTS}
        unknown = [k for k in config if k not in DDPM_SCHEDULER_DEFAULTS and not k.startswith("_")]
        if unknown:
            raise ValueError(f"DDPMSchedulerHIP.from_config: keys this scheduler does not know: {unknown}")
        cname = dict(config).get("_class_name", "DDPMScheduler")
        if cname != "DDPMScheduler":
            raise NotImplementedError(f"scheduler class {cname!r}: only DDPMScheduler (what kandinsky2_2_model.py:29-41 loads) is built")
        sch = cls(**known)
        sch.missing_keys = tuple(sorted(k for k in DDPM_SCHEDULER_DEFAULTS if k not in known))
        return sch

    def set_timesteps(self, num_inference_steps: int, device="cuda"):
        T, c = self.num_train_timesteps, self.config
        if num_inference_steps > T:
            raise ValueError("num_inference_steps exceeds num_train_timesteps")
        ratio = T // num_inference_steps
        if c.timestep_spacing == "leading":
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].astype(np.int64) + c.steps_offset
        elif c.timestep_spacing == "linspace":
            ts = np.linspace(0, T - 1, num_inference_steps).round()[::-1].astype(np.int64)
        else:
            ts = np.round(np.arange(T, 0, -T / num_inference_steps)).astype(np.int64) - 1
        self.num_inference_steps = num_inference_steps
        ac = self.alphas_cumprod
        tab = np.zeros((num_inference_steps, 8), dtype=np.float32)
        for row, t in enumerate(ts):
            prev = t - ratio                                          # DDPMScheduler.previous_timestep
            a_t, a_prev = ac[t], (ac[prev] if prev >= 0 else 1.0)
            cur_alpha = a_t / a_prev
            cur_beta = 1.0 - cur_alpha
            var = max((1.0 - a_prev) / (1.0 - a_t) * cur_beta, 1e-20)  # _get_variance
            lo, hi = math.log(var), math.log(cur_beta)                 # learned_range: min_log, max_log
            if c.variance_type in ("fixed_small", "fixed_small_log"):
                hi = lo                                                # std = sqrt(var) (= exp(0.5 log var)) whatever the UNet predicts
            elif c.variance_type in ("fixed_large", "fixed_large_log"):
                lo = hi
            tab[row] = [math.sqrt(1.0 / a_t), math.sqrt(1.0 / a_t - 1.0), math.sqrt(a_prev) * cur_beta / (1.0 - a_t),
                        math.sqrt(cur_alpha) * (1.0 - a_prev) / (1.0 - a_t), lo, hi, 1.0 if t > 0 else 0.0, float(t)]
        self._table_host = tab
        self._table = torch.from_numpy(tab).to(device)
        self._row = {int(t): i for i, t in enumerate(ts)}
        self.timesteps = torch.from_numpy(ts.copy()).to(device)
        self._scratch = None
        return self

    def scale_model_input(self, sample, timestep=None):
        return sample

    @torch.no_grad()
    def add_noise(self, original_samples, noise, timesteps):
        """DDPMScheduler.add_noise: sqrt(alphas_cumprod[t]) * x + sqrt(1 - alphas_cumprod[t]) * noise (one timestep per call)."""
        t = int(torch.as_tensor(timesteps).reshape(-1)[0])
        x, nz = original_samples.detach().float().contiguous(), noise.detach().float().contiguous()
        if x.shape != nz.shape or x.dim() != 4:
            raise ValueError("add_noise: samples and noise must both be [N, C, h, w]")
        out = torch.empty_like(x)
        N, Cc, H, W = x.shape
        a = float(self.alphas_cumprod[t])
        _lib.check(_lib.lib().k22_blend_noised(None, x.data_ptr(), nz.data_ptr(), None, math.sqrt(a), math.sqrt(1.0 - a), out.data_ptr(),
                                               N, Cc, H * W, 0, _lib.current_stream()))
        return out

    @torch.no_grad()
    def step(self, model_output, timestep, sample, generator=None, noise: Optional[torch.Tensor] = None, return_dict: bool = True,
             guidance_scale: Optional[float] = None):
        """model_output [N,8,h,w] = (eps | variance channels: used by learned_range, ignored by the fixed variance types) already
        guided, sample [N,4,h,w] -> prev_sample.  With
        guidance_scale=..., model_output is the RAW UNet output of the CFG batch ordered [cond | uncond] and sample holds the
        duplicated halves (the fused call the 2.2 pipeline below makes)."""
        L = _lib.lib()
        N, C8, H, W = model_output.shape
        if C8 != 8 or tuple(sample.shape) != (N, 4, H, W):
            raise ValueError("model_output must be [N,8,h,w] (eps | variance) and sample [N,4,h,w]")
        row = self._row[int(timestep)]
        HW = H * W
        need = L.k22_sampler_scratch_bytes(N, HW)
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(need, dtype=torch.uint8, device=sample.device)
        if noise is not None:
            nz = noise
        else:   # diffusers' randn_tensor: a CPU generator draws on the CPU and the noise is moved to the sample's device
            gdev = generator.device if generator is not None else sample.device
            nz = torch.randn(sample.shape, generator=generator, device=gdev).to(sample.device)
        x, mo = sample.detach().float().contiguous(), model_output.detach().float().contiguous()
        out = torch.empty_like(x)
        fused = guidance_scale is not None
        _lib.check(L.k22_sampler_step(x.data_ptr(), mo.data_ptr(), nz.float().contiguous().data_ptr(), None, None, self._table.data_ptr(), row,
                                      float(guidance_scale) if fused else 1.0, 1 if fused else 0, -min(self.clip, 3.0e38), min(self.clip, 3.0e38), -1, 0.0,
                                      self._scratch.data_ptr(), out.data_ptr(), None, N, HW, _lib.current_stream()))
        return SimpleNamespace(prev_sample=out) if return_dict else (out,)
