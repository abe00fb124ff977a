"""Drop-in MoVQ *decoder* (the image_encoder.decode() half the sampling path uses) and *encoder* (image_encoder.encode(),
the img2img / inpainting pre-step) backed by the HIP engine.

Mirrors the reference interface for this path:
    Kandinsky2_1.image_encoder = MOVQ(**config["image_enc_params"]["params"])     (kandinsky2_1_model.py:107-113)
    samples = self.image_encoder.decode(samples / self.scale)                      (kandinsky2_1_model.py:288-289)
    process_images(samples)                                                        (kandinsky2/utils.py:57-70)
State-dict keys / shapes are those of the reference's MOVQ for `post_quant_conv` and `decoder.*`
(kandinsky2/vqgan/autoencoder.py:163-185, movq_modules.py:228-357); encoder / quantizer keys of a full MOVQ
checkpoint are accepted and ignored (load_state_dict(strict=False)).  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib

MOVQ_CONFIG_2_1 = {  # CONFIG_2_1["image_enc_params"]["params"] (kandinsky2/configs.py:68-90)
    "embed_dim": 4, "n_embed": 16384,
    "ddconfig": {"double_z": False, "z_channels": 4, "resolution": 256, "in_channels": 3, "out_ch": 3, "ch": 128,
                 "ch_mult": [1, 2, 2, 4], "num_res_blocks": 2, "attn_resolutions": [32], "dropout": 0.0},
}


class MoVQArch:
    def __init__(self, ddconfig: dict, embed_dim: int = 4):
        self.ch = ddconfig["ch"]
        self.ch_mult = tuple(ddconfig["ch_mult"])
        self.num_res_blocks = ddconfig["num_res_blocks"]
        self.z_channels = ddconfig["z_channels"]
        self.out_ch = ddconfig["out_ch"]
        self.zq_ch = embed_dim
        res = ddconfig["resolution"]
        n = len(self.ch_mult)
        # MOVQDecoder.__init__ tracks a nominal resolution: level i runs at resolution / 2^i (movq_modules.py:258, 301-303)
        self.attn_levels = [i for i in range(n) if (res // 2 ** i) in ddconfig["attn_resolutions"]]
        if self.z_channels != 4 or embed_dim != 4:
            raise NotImplementedError("the engine supports z_channels == embed_dim == 4")

    def blocks(self):
        """(kind, prefix, cin, cout) in forward order: 'res' | 'attn' | 'up' (Upsample conv)."""
        n = len(self.ch_mult)
        out = []
        block_in = self.ch * self.ch_mult[n - 1]
        out += [("res", "decoder.mid.block_1", block_in, block_in), ("attn", "decoder.mid.attn_1", block_in, block_in),
                ("res", "decoder.mid.block_2", block_in, block_in)]
        for lvl in reversed(range(n)):
            block_out = self.ch * self.ch_mult[lvl]
            for i in range(self.num_res_blocks + 1):
                out.append(("res", f"decoder.up.{lvl}.block.{i}", block_in, block_out))
                block_in = block_out
                if lvl in self.attn_levels:
                    out.append(("attn", f"decoder.up.{lvl}.attn.{i}", block_in, block_in))
            if lvl != 0:
                out.append(("up", f"decoder.up.{lvl}.upsample.conv", block_in, block_in))
        return out, block_in


def movq_param_shapes(a: MoVQArch) -> "OrderedDict[str, tuple]":
    """state_dict keys -> shapes of post_quant_conv + decoder, in the reference's registration order."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    zq = a.zq_ch

    def conv(name, o, i, k):
        s[name + ".weight"] = (o, i, k, k)
        s[name + ".bias"] = (o,)

    def snorm(name, c):
        s[name + ".norm_layer.weight"] = (c,)
        s[name + ".norm_layer.bias"] = (c,)
        conv(name + ".conv_y", c, zq, 1)
        conv(name + ".conv_b", c, zq, 1)

    conv("post_quant_conv", a.z_channels, zq, 1)
    blocks, last = a.blocks()
    conv("decoder.conv_in", blocks[0][2], a.z_channels, 3)
    for kind, pfx, cin, cout in blocks:
        if kind == "res":
            snorm(pfx + ".norm1", cin)
            conv(pfx + ".conv1", cout, cin, 3)
            snorm(pfx + ".norm2", cout)
            conv(pfx + ".conv2", cout, cout, 3)
            if cin != cout:
                conv(pfx + ".nin_shortcut", cout, cin, 1)
        elif kind == "attn":
            snorm(pfx + ".norm", cin)
            for n in ("q", "k", "v", "proj_out"):
                conv(pfx + "." + n, cin, cin, 1)
        else:
            conv(pfx, cout, cin, 3)
    snorm("decoder.norm_out", last)
    conv("decoder.conv_out", a.out_ch, last, 3)
    return s


def init_movq_state_dict(a: MoVQArch, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Seeded random weights (there are no checkpoints on the box): fan-in scaled convs, norms around 1, and
    conv_y biased to 1 so that SpatialNorm starts near a plain GroupNorm."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in movq_param_shapes(a).items():
        leaf = name.rsplit(".", 1)[1]
        if ".norm_layer." in name:
            t = torch.randn(shape, generator=g) * 0.1 + (1.0 if leaf == "weight" else 0.0)
        elif leaf == "bias":
            t = torch.randn(shape, generator=g) * 0.02 + (1.0 if ".conv_y." in name else 0.0)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = torch.randn(shape, generator=g) * (1.0 / math.sqrt(fan_in)) * (0.5 if (".conv_y." in name or ".conv_b." in name) else 1.0)
        sd[name] = t
    return sd


def _pad_rows(w: torch.Tensor, mult: int = 64) -> torch.Tensor:
    o = w.shape[0]
    op = (o + mult - 1) // mult * mult
    if op == o:
        return w
    return torch.cat([w, torch.zeros((op - o,) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)], 0)


def pack_movq_arena(a: MoVQArch, sd: Dict[str, torch.Tensor], tdtype: torch.dtype, device) -> Tuple[torch.Tensor, "OrderedDict[str, Tuple[int, int]]"]:
    f32 = torch.float32
    ent: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def get(name):
        return sd[name].detach().to(device=device, dtype=f32)

    for name, shape in movq_param_shapes(a).items():
        w = get(name)
        if name == "decoder.conv_in.weight":
            wp = torch.zeros(w.shape[0], 3, 3, 64, device=device)
            wp[..., : w.shape[1]] = w.permute(0, 2, 3, 1)
            ent[name] = _pad_rows(wp.reshape(w.shape[0], -1)).to(tdtype).contiguous()
        elif name.startswith("post_quant_conv") or ".conv_y." in name or ".conv_b." in name or ".norm_layer." in name or name.endswith(".bias"):
            ent[name] = w.reshape(w.shape[0], -1).contiguous() if w.dim() == 4 else w.contiguous()
        elif len(shape) == 4 and shape[2] == 3:
            ent[name] = _pad_rows(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)).to(tdtype).contiguous()
        else:  # 1x1 projections feeding the MFMA GEMM
            ent[name] = _pad_rows(w.reshape(w.shape[0], -1)).to(tdtype).contiguous()
    table: "OrderedDict[str, Tuple[int, int]]" = OrderedDict()
    off = 0
    for name, t in ent.items():
        nb = t.numel() * t.element_size()
        table[name] = (off, nb)
        off += (nb + 255) // 256 * 256
    arena = torch.zeros(off + 256, dtype=torch.uint8, device=device)
    for name, t in ent.items():
        o, nb = table[name]
        arena[o:o + nb] = t.reshape(-1).view(torch.uint8)
    return arena, table


class MoVQDecoderHIP(nn.Module):
    """MI355X-native `MOVQ.decode` (kandinsky2/vqgan/autoencoder.py:182-185)."""

    def __init__(self, ddconfig: Optional[dict] = None, n_embed: int = 16384, embed_dim: int = 4,
                 backend_dtype: torch.dtype = torch.bfloat16):
        super().__init__()
        self.arch = MoVQArch(ddconfig or MOVQ_CONFIG_2_1["ddconfig"], embed_dim)
        self.backend_dtype = backend_dtype
        from .unet import _register
        for name, shape in movq_param_shapes(self.arch).items():
            _register(self, name, nn.Parameter(torch.zeros(shape), requires_grad=False))
        self._handle = None
        self._arena = None
        self._ws = None
        self._plan_key = None

    def _release(self):
        if self._handle is not None:
            _lib.lib().k22_movq_destroy(self._handle)
            self._handle = None
        self._arena = self._ws = self._plan_key = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def load_state_dict(self, state_dict, strict: bool = False, **kw):
        own = set(movq_param_shapes(self.arch).keys())
        r = super().load_state_dict({k: v for k, v in state_dict.items() if k in own}, strict=False, **kw)
        self._release()
        if strict and r.missing_keys:
            raise RuntimeError(f"missing decoder keys: {r.missing_keys[:4]} ...")
        return r

    def _apply(self, fn, *args, **kwargs):
        r = super()._apply(fn, *args, **kwargs)
        self._release()
        return r

    def prepare(self):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("MoVQDecoderHIP runs on the GPU only (no CPU fallback): move it with .to('cuda')")
        L = _lib.lib()
        self._release()
        self._arena, table = pack_movq_arena(self.arch, self.state_dict(), self.backend_dtype, dev)
        a = self.arch
        cfg = _lib.K22MoVQConfig()
        cfg.dtype = _lib.dtype_code(self.backend_dtype)
        cfg.ch = a.ch
        cfg.n_levels = len(a.ch_mult)
        for i, v in enumerate(a.ch_mult):
            cfg.ch_mult[i] = v
        cfg.num_res_blocks = a.num_res_blocks
        cfg.attn_levels = sum(1 << i for i in a.attn_levels)
        cfg.z_channels = a.z_channels
        cfg.out_ch = a.out_ch
        base = self._arena.data_ptr()
        arr = (_lib.K22Weight * len(table))()
        self._names = []
        for i, (name, (off, _n)) in enumerate(table.items()):
            nb = name.encode()
            self._names.append(nb)
            arr[i].name = nb
            arr[i].ptr = base + off
        h = C.c_void_p()
        _lib.check(L.k22_movq_create(C.byref(cfg), arr, len(table), C.byref(h)))
        self._handle = h
        return self

    def _ensure_plan(self, B, h, w):
        if self._handle is None:
            self.prepare()
        if self._plan_key != (B, h, w):
            self._plan_key = None   # a failed plan / bind leaves the native engine without a plan: never skip re-planning after it
            nbytes = C.c_size_t()
            _lib.check(_lib.lib().k22_movq_plan(self._handle, B, h, w, C.byref(nbytes)))
            self._ws = torch.empty(nbytes.value + 256, dtype=torch.uint8, device=self._arena.device)
            al = (self._ws.data_ptr() + 255) // 256 * 256
            _lib.check(_lib.lib().k22_movq_bind(self._handle, al, nbytes.value))
            self._plan_key = (B, h, w)

    @torch.no_grad()
    def decode(self, quant: torch.Tensor, return_uint8: bool = False):
        """quant [B,4,h,w] -> image [B,3,8h,8w] float32 in about [-1,1] (and, with return_uint8, the NHWC uint8
        image of process_images)."""
        if quant.device.type != "cuda":
            raise RuntimeError("MoVQDecoderHIP.decode: input must be on the GPU (no CPU fallback)")
        B, Cz, h, w = quant.shape
        if Cz != 4:
            raise ValueError("expected a 4-channel latent")
        self._ensure_plan(B, h, w)
        z = quant.detach().float().contiguous()
        up = 1 << (len(self.arch.ch_mult) - 1)
        out = torch.empty(B, self.arch.out_ch, h * up, w * up, dtype=torch.float32, device=quant.device)
        u8 = torch.empty(B, h * up, w * up, self.arch.out_ch, dtype=torch.uint8, device=quant.device) if return_uint8 else None
        _lib.check(_lib.lib().k22_movq_decode(self._handle, z.data_ptr(), out.data_ptr(), _lib.ptr(u8), _lib.current_stream()))
        return (out, u8) if return_uint8 else out

    forward = decode


# ======================================================================================================================
# Encoder: MOVQ.encode = Encoder.forward + quant_conv (kandinsky2/vqgan/autoencoder.py:176-180, vqgan_blocks.py:253-367)
# ======================================================================================================================
def movq_encoder_blocks(a: MoVQArch):
    """(kind, prefix, cin, cout) in forward order: 'res' | 'attn' | 'down' (Downsample conv); and the final width."""
    out = []
    n = len(a.ch_mult)
    block_in = a.ch
    for lvl in range(n):
        block_out = a.ch * a.ch_mult[lvl]
        for i in range(a.num_res_blocks):
            out.append(("res", f"encoder.down.{lvl}.block.{i}", block_in, block_out))
            block_in = block_out
            if lvl in a.attn_levels:
                out.append(("attn", f"encoder.down.{lvl}.attn.{i}", block_in, block_in))
        if lvl != n - 1:
            out.append(("down", f"encoder.down.{lvl}.downsample.conv", block_in, block_in))
    out += [("res", "encoder.mid.block_1", block_in, block_in), ("attn", "encoder.mid.attn_1", block_in, block_in),
            ("res", "encoder.mid.block_2", block_in, block_in)]
    return out, block_in


def movq_encoder_param_shapes(a: MoVQArch, in_channels: int = 3) -> "OrderedDict[str, tuple]":
    """state_dict keys -> shapes of encoder.* + quant_conv.* as the reference's MOVQ registers them."""
    s: "OrderedDict[str, tuple]" = OrderedDict()

    def conv(name, o, i, k):
        s[name + ".weight"] = (o, i, k, k)
        s[name + ".bias"] = (o,)

    def norm(name, c):
        s[name + ".weight"] = (c,)
        s[name + ".bias"] = (c,)

    conv("encoder.conv_in", a.ch, in_channels, 3)
    blocks, last = movq_encoder_blocks(a)
    for kind, pfx, cin, cout in blocks:
        if kind == "res":
            norm(pfx + ".norm1", cin)
            conv(pfx + ".conv1", cout, cin, 3)
            norm(pfx + ".norm2", cout)
            conv(pfx + ".conv2", cout, cout, 3)
            if cin != cout:
                conv(pfx + ".nin_shortcut", cout, cin, 1)
        elif kind == "attn":
            norm(pfx + ".norm", cin)
            for n in ("q", "k", "v", "proj_out"):
                conv(pfx + "." + n, cin, cin, 1)
        else:
            conv(pfx, cout, cin, 3)
    norm("encoder.norm_out", last)
    conv("encoder.conv_out", a.z_channels, last, 3)
    conv("quant_conv", a.zq_ch, a.z_channels, 1)
    return s


def init_movq_encoder_state_dict(a: MoVQArch, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in movq_encoder_param_shapes(a).items():
        leaf = name.rsplit(".", 1)[1]
        if len(shape) == 1 and (".norm" in name):
            t = torch.randn(shape, generator=g) * 0.1 + (1.0 if leaf == "weight" else 0.0)
        elif leaf == "bias":
            t = torch.randn(shape, generator=g) * 0.02
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = torch.randn(shape, generator=g) * (1.0 / math.sqrt(fan_in))
        sd[name] = t
    return sd


def pack_movq_encoder_arena(a: MoVQArch, sd: Dict[str, torch.Tensor], tdtype: torch.dtype, device):
    f32 = torch.float32
    ent: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in movq_encoder_param_shapes(a).items():
        w = sd[name].detach().to(device=device, dtype=f32)
        if name == "encoder.conv_in.weight":      # Cin 3 -> 64 (the prepare kernel zero-extends the image the same way)
            wp = torch.zeros(w.shape[0], 3, 3, 64, device=device)
            wp[..., : w.shape[1]] = w.permute(0, 2, 3, 1)
            ent[name] = _pad_rows(wp.reshape(w.shape[0], -1)).to(tdtype).contiguous()
        elif name.startswith("quant_conv") or name.endswith(".bias") or len(shape) == 1:
            ent[name] = w.reshape(w.shape[0], -1).contiguous() if w.dim() == 4 else w.contiguous()
        elif shape[2] == 3:
            ent[name] = _pad_rows(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)).to(tdtype).contiguous()
        else:
            ent[name] = _pad_rows(w.reshape(w.shape[0], -1)).to(tdtype).contiguous()
    table: "OrderedDict[str, Tuple[int, int]]" = OrderedDict()
    off = 0
    for name, t in ent.items():
        nb = t.numel() * t.element_size()
        table[name] = (off, nb)
        off += (nb + 255) // 256 * 256
    arena = torch.zeros(off + 256, dtype=torch.uint8, device=device)
    for name, t in ent.items():
        o, nb = table[name]
        arena[o:o + nb] = t.reshape(-1).view(torch.uint8)
    return arena, table


class MoVQEncoderHIP(nn.Module):
    """MI355X-native `MOVQ.encode` (kandinsky2/vqgan/autoencoder.py:176-180): image [B,3,H,W] in [-1,1] -> latent [B,4,H/8,W/8]
    (before the pipeline multiplies by its latent scale, kandinsky2_1_model.py:467).  State-dict keys are the reference's
    `encoder.*` and `quant_conv.*`; the other keys of a full MOVQ checkpoint are ignored."""

    def __init__(self, ddconfig: Optional[dict] = None, n_embed: int = 16384, embed_dim: int = 4,
                 backend_dtype: torch.dtype = torch.bfloat16):
        super().__init__()
        dd = ddconfig or MOVQ_CONFIG_2_1["ddconfig"]
        self.arch = MoVQArch(dd, embed_dim)
        self.in_channels = dd.get("in_channels", 3)
        if self.in_channels != 3:
            raise NotImplementedError("the encoder engine takes 3-channel images")
        self.backend_dtype = backend_dtype
        from .unet import _register
        for name, shape in movq_encoder_param_shapes(self.arch).items():
            _register(self, name, nn.Parameter(torch.zeros(shape), requires_grad=False))
        self._handle = None
        self._arena = self._ws = self._plan_key = None

    def _release(self):
        if self._handle is not None:
            _lib.lib().k22_movq_destroy(self._handle)
            self._handle = None
        self._arena = self._ws = self._plan_key = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def load_state_dict(self, state_dict, strict: bool = False, **kw):
        own = set(movq_encoder_param_shapes(self.arch).keys())
        r = super().load_state_dict({k: v for k, v in state_dict.items() if k in own}, strict=False, **kw)
        self._release()
        if strict and r.missing_keys:
            raise RuntimeError(f"missing encoder keys: {r.missing_keys[:4]} ...")
        return r

    def _apply(self, fn, *args, **kwargs):
        r = super()._apply(fn, *args, **kwargs)
        self._release()
        return r

    def prepare(self):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("MoVQEncoderHIP runs on the GPU only (no CPU fallback): move it with .to('cuda')")
        L = _lib.lib()
        self._release()
        self._arena, table = pack_movq_encoder_arena(self.arch, self.state_dict(), self.backend_dtype, dev)
        a = self.arch
        cfg = _lib.K22MoVQConfig()
        cfg.dtype = _lib.dtype_code(self.backend_dtype)
        cfg.ch = a.ch
        cfg.n_levels = len(a.ch_mult)
        for i, v in enumerate(a.ch_mult):
            cfg.ch_mult[i] = v
        cfg.num_res_blocks = a.num_res_blocks
        cfg.attn_levels = sum(1 << i for i in a.attn_levels)
        cfg.z_channels = a.z_channels
        cfg.out_ch = a.out_ch
        base = self._arena.data_ptr()
        arr = (_lib.K22Weight * len(table))()
        self._names = []
        for i, (name, (off, _n)) in enumerate(table.items()):
            nb = name.encode()
            self._names.append(nb)
            arr[i].name = nb
            arr[i].ptr = base + off
        h = C.c_void_p()
        _lib.check(L.k22_movq_create(C.byref(cfg), arr, len(table), C.byref(h)))
        self._handle = h
        return self

    @torch.no_grad()
    def encode(self, image: torch.Tensor) -> torch.Tensor:
        if image.device.type != "cuda":
            raise RuntimeError("MoVQEncoderHIP.encode: input must be on the GPU (no CPU fallback)")
        B, Ci, H, W = image.shape
        if Ci != 3:
            raise ValueError("expected a 3-channel image")
        if self._handle is None:
            self.prepare()
        if self._plan_key != (B, H, W):
            self._plan_key = None   # a failed plan / bind leaves the native engine without a plan: never skip re-planning after it
            nbytes = C.c_size_t()
            _lib.check(_lib.lib().k22_movq_plan_encoder(self._handle, B, H, W, C.byref(nbytes)))
            self._ws = torch.empty(nbytes.value + 256, dtype=torch.uint8, device=self._arena.device)
            al = (self._ws.data_ptr() + 255) // 256 * 256
            _lib.check(_lib.lib().k22_movq_bind(self._handle, al, nbytes.value))
            self._plan_key = (B, H, W)
        x = image.detach().float().contiguous()
        down = 1 << (len(self.arch.ch_mult) - 1)
        out = torch.empty(B, 4, H // down, W // down, dtype=torch.float32, device=image.device)
        _lib.check(_lib.lib().k22_movq_encode(self._handle, x.data_ptr(), out.data_ptr(), _lib.current_stream()))
        return out

    forward = encode
