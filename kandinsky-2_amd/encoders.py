"""Conditioning encoders on the HIP encoder engine (csrc/encoder.hip; SURVEY 8f-3, a "next" row): the three transformer towers
Kandinsky 2.1 runs once per prompt / image before the prior and the denoising loop,

    CLIPVisionModelWithProjectionHIP   transformers' class of that name under its own keys: the CLIP ViT-bigG/14 image encoder Kandinsky 2.2 loads
                          (kandinsky2/kandinsky2_2_model.py:24) - 48 layers x 1664, 16 heads of 104 channels, 1280-d embedding.
    CLIPModelHIP          clip.load("ViT-L/14") as the reference uses it: the text tower walked by generate_clip_emb
                          (kandinsky2/kandinsky2_1_model.py:159-168) and encode_image (:177-181).  Parameters under the OpenAI
                          `clip.model.CLIP` state_dict keys, so that checkpoint loads unchanged.
    MultilingualCLIPHIP   kandinsky2/model/text_encoders.py:108-122 (transformers' XLMRobertaModel + masked mean +
                          LinearTransformation), same keys (`transformer.*`, `LinearTransformation.*`), same forward signature.
    TextEncoderHIP        kandinsky2/model/text_encoders.py:125-167 for model_name == "multiclip" (the 2.1 configuration).
    HIPConditioner        the conditioner object Kandinsky2_1HIP drives (pipeline.py), built from the modules above + the two
                          tokenizers: what Kandinsky2_1.encode_text / generate_clip_emb / encode_images do (:117-181).

Tokenisation (XLM-R sentencepiece, CLIP byte-pair encoding) and CLIP's image preprocessing are host-side string / PIL work and stay
with the tokenizer / preprocess objects the caller passes in, as in the reference.  Public tensors are fp32; the engine computes in
`backend_dtype` (bf16 MFMA product path, fp32 parity path).  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from .prior import _pad_rows
from .unet import _register

ENC_CLIP_TEXT, ENC_CLIP_VISION, ENC_XLMR = 0, 1, 2

# clip.model.CLIP(...) constructor arguments of ViT-L/14 (the CONFIG_2_1 "clip_name", kandinsky2/configs.py:65)
CLIP_VITL14 = {"embed_dim": 768, "image_resolution": 224, "vision_layers": 24, "vision_width": 1024, "vision_patch_size": 14,
               "context_length": 77, "vocab_size": 49408, "transformer_width": 768, "transformer_heads": 12, "transformer_layers": 12}
# config.json of xlm-roberta-large, the transformer inside M-CLIP/XLM-Roberta-Large-Vit-L-14 (text_enc_params, configs.py:89-94)
XLMR_LARGE = {"vocab_size": 250002, "hidden_size": 1024, "num_hidden_layers": 24, "num_attention_heads": 16, "intermediate_size": 4096,
              "max_position_embeddings": 514, "type_vocab_size": 1, "layer_norm_eps": 1e-5, "pad_token_id": 1, "hidden_act": "gelu"}


# vision_config (+ projection_dim) of the `image_encoder` sub-folder of kandinsky-community/kandinsky-2-2-prior, the
# CLIPVisionModelWithProjection Kandinsky2_2.__init__ loads (kandinsky2/kandinsky2_2_model.py:24): open_clip ViT-bigG/14 in transformers'
# layout [published architecture: 48 layers x 1664 wide, 16 heads of 104, MLP 8192, erf GELU, 14-px patches of a 224-px image, 1280-d output]
CLIP_BIGG_VISION = {"hidden_size": 1664, "intermediate_size": 8192, "num_hidden_layers": 48, "num_attention_heads": 16, "image_size": 224,
                    "patch_size": 14, "projection_dim": 1280, "hidden_act": "gelu", "layer_norm_eps": 1e-5}


def tiny_clip_vision_hf_config() -> dict:
    """Same structure at 2 layers x 832 wide: EIGHT HEADS OF 104 (the head width of bigG, which the 64-wide flash kernel does not take;
    832 = 13 x 64 keeps the GEMM K alignment) and an MLP that is not 4 x width."""
    return dict(CLIP_BIGG_VISION, hidden_size=832, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=8, image_size=56, projection_dim=64)


def tiny_clip_config() -> dict:
    """Same structure, 2 layers x 128 wide, 56-px images (golden fixtures / quick parity tests)."""
    return dict(CLIP_VITL14, embed_dim=64, image_resolution=56, vision_layers=2, vision_width=128, vocab_size=1000,
                transformer_width=128, transformer_heads=2, transformer_layers=2)


def tiny_xlmr_config() -> dict:
    return dict(XLMR_LARGE, vocab_size=1000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512)


# ---- parameter tables on the reference's state_dict keys ---------------------------------------------------------------------------
def _block(s, p, W):
    s[p + ".attn.in_proj_weight"] = (3 * W, W); s[p + ".attn.in_proj_bias"] = (3 * W,)
    s[p + ".attn.out_proj.weight"] = (W, W); s[p + ".attn.out_proj.bias"] = (W,)
    s[p + ".ln_1.weight"] = (W,); s[p + ".ln_1.bias"] = (W,)
    s[p + ".mlp.c_fc.weight"] = (4 * W, W); s[p + ".mlp.c_fc.bias"] = (4 * W,)
    s[p + ".mlp.c_proj.weight"] = (W, 4 * W); s[p + ".mlp.c_proj.bias"] = (W,)
    s[p + ".ln_2.weight"] = (W,); s[p + ".ln_2.bias"] = (W,)


def clip_param_shapes(cfg: dict) -> "OrderedDict[str, tuple]":
    """state_dict of clip.model.CLIP with a VisionTransformer (clip/model.py, OpenAI CLIP)."""
    E, Wt, Wv, p = cfg["embed_dim"], cfg["transformer_width"], cfg["vision_width"], cfg["vision_patch_size"]
    P = (cfg["image_resolution"] // p) ** 2
    s: "OrderedDict[str, tuple]" = OrderedDict()
    s["positional_embedding"] = (cfg["context_length"], Wt)
    s["text_projection"] = (Wt, E)
    s["logit_scale"] = ()
    s["visual.class_embedding"] = (Wv,)
    s["visual.positional_embedding"] = (P + 1, Wv)
    s["visual.proj"] = (Wv, E)
    s["visual.conv1.weight"] = (Wv, 3, p, p)
    s["visual.ln_pre.weight"] = (Wv,); s["visual.ln_pre.bias"] = (Wv,)
    for l in range(cfg["vision_layers"]):
        _block(s, f"visual.transformer.resblocks.{l}", Wv)
    s["visual.ln_post.weight"] = (Wv,); s["visual.ln_post.bias"] = (Wv,)
    for l in range(cfg["transformer_layers"]):
        _block(s, f"transformer.resblocks.{l}", Wt)
    s["token_embedding.weight"] = (cfg["vocab_size"], Wt)
    s["ln_final.weight"] = (Wt,); s["ln_final.bias"] = (Wt,)
    return s


def multiclip_param_shapes(cfg: dict, in_features=1024, out_features=768) -> "OrderedDict[str, tuple]":
    """state_dict of MultilingualCLIP (text_encoders.py:108-116): XLMRobertaModel under `transformer.` + LinearTransformation."""
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    s: "OrderedDict[str, tuple]" = OrderedDict()

    def lin(name, o, i):
        s[name + ".weight"] = (o, i); s[name + ".bias"] = (o,)

    e = "transformer.embeddings."
    s[e + "word_embeddings.weight"] = (cfg["vocab_size"], H)
    s[e + "position_embeddings.weight"] = (cfg["max_position_embeddings"], H)
    s[e + "token_type_embeddings.weight"] = (cfg["type_vocab_size"], H)
    s[e + "LayerNorm.weight"] = (H,); s[e + "LayerNorm.bias"] = (H,)
    for l in range(cfg["num_hidden_layers"]):
        p = f"transformer.encoder.layer.{l}."
        lin(p + "attention.self.query", H, H); lin(p + "attention.self.key", H, H); lin(p + "attention.self.value", H, H)
        lin(p + "attention.output.dense", H, H)
        s[p + "attention.output.LayerNorm.weight"] = (H,); s[p + "attention.output.LayerNorm.bias"] = (H,)
        lin(p + "intermediate.dense", I, H)
        lin(p + "output.dense", H, I)
        s[p + "output.LayerNorm.weight"] = (H,); s[p + "output.LayerNorm.bias"] = (H,)
    lin("transformer.pooler.dense", H, H)          # part of XLMRobertaModel's state_dict; MultilingualCLIP.forward never uses it
    lin("LinearTransformation", out_features, in_features)
    return s


def clip_vision_hf_param_shapes(cfg: dict) -> "OrderedDict[str, tuple]":
    """state_dict of transformers' CLIPVisionModelWithProjection (models/clip/modeling_clip.py); the non-persistent position_ids buffer
    of older versions is not a parameter and is ignored on load."""
    H, I, p = cfg["hidden_size"], cfg["intermediate_size"], cfg["patch_size"]
    P = (cfg["image_size"] // p) ** 2
    s: "OrderedDict[str, tuple]" = OrderedDict()
    e = "vision_model.embeddings."
    s[e + "class_embedding"] = (H,)
    s[e + "patch_embedding.weight"] = (H, 3, p, p)
    s[e + "position_embedding.weight"] = (P + 1, H)
    s["vision_model.pre_layrnorm.weight"] = (H,); s["vision_model.pre_layrnorm.bias"] = (H,)     # [sic]: transformers' spelling
    for l in range(cfg["num_hidden_layers"]):
        q = f"vision_model.encoder.layers.{l}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[q + f"self_attn.{nm}.weight"] = (H, H); s[q + f"self_attn.{nm}.bias"] = (H,)
        s[q + "layer_norm1.weight"] = (H,); s[q + "layer_norm1.bias"] = (H,)
        s[q + "mlp.fc1.weight"] = (I, H); s[q + "mlp.fc1.bias"] = (I,)
        s[q + "mlp.fc2.weight"] = (H, I); s[q + "mlp.fc2.bias"] = (H,)
        s[q + "layer_norm2.weight"] = (H,); s[q + "layer_norm2.bias"] = (H,)
    s["vision_model.post_layernorm.weight"] = (H,); s["vision_model.post_layernorm.bias"] = (H,)
    s["visual_projection.weight"] = (cfg["projection_dim"], H)
    return s


def _init(shapes, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in shapes.items():
        leaf = name.rsplit(".", 1)[-1]
        if name == "logit_scale":
            t = torch.tensor(math.log(1 / 0.07))
        elif "ln_" in name or "LayerNorm" in name or "layer_norm" in name or "layernorm" in name or "layrnorm" in name:
            t = torch.randn(shape, generator=g) * 0.1 + (1.0 if leaf == "weight" else 0.0)
        elif "embedding" in name:
            t = torch.randn(shape, generator=g) * 0.3
        elif leaf in ("bias", "in_proj_bias"):
            t = torch.randn(shape, generator=g) * 0.02
        elif name in ("text_projection", "visual.proj"):
            t = torch.randn(shape, generator=g) * shape[0] ** -0.5
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = torch.randn(shape, generator=g) * fan_in ** -0.5
        sd[name] = t
    return sd


def init_clip_state_dict(cfg: dict, seed: int = 0):
    return _init(clip_param_shapes(cfg), seed)


def init_clip_vision_hf_state_dict(cfg: dict, seed: int = 0):
    return _init(clip_vision_hf_param_shapes(cfg), seed)


def init_multiclip_state_dict(cfg: dict, in_features=1024, out_features=768, seed: int = 0):
    return _init(multiclip_param_shapes(cfg, in_features, out_features), seed)


# ---- arenas: reference keys -> the engine's names and layouts (include/k22.h, "conditioning encoders") -----------------------------
def _finish_arena(ent, device):
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


def _clip_layers(ent, sd, src, n_layers, tdtype, f):
    for l in range(n_layers):
        p, q = f"{src}.resblocks.{l}", f"layers.{l}"
        ent[q + ".qkv.weight"] = _pad_rows(f(sd[p + ".attn.in_proj_weight"])).to(tdtype).contiguous()
        ent[q + ".qkv.bias"] = f(sd[p + ".attn.in_proj_bias"]).contiguous()
        ent[q + ".proj.weight"] = _pad_rows(f(sd[p + ".attn.out_proj.weight"])).to(tdtype).contiguous()
        ent[q + ".proj.bias"] = f(sd[p + ".attn.out_proj.bias"]).contiguous()
        ent[q + ".fc.weight"] = _pad_rows(f(sd[p + ".mlp.c_fc.weight"])).to(tdtype).contiguous()
        ent[q + ".fc.bias"] = f(sd[p + ".mlp.c_fc.bias"]).contiguous()
        ent[q + ".out.weight"] = _pad_rows(f(sd[p + ".mlp.c_proj.weight"])).to(tdtype).contiguous()
        ent[q + ".out.bias"] = f(sd[p + ".mlp.c_proj.bias"]).contiguous()
        for k in ("ln_1", "ln_2"):
            ent[f"{q}.{k}.weight"] = f(sd[f"{p}.{k}.weight"]).contiguous()
            ent[f"{q}.{k}.bias"] = f(sd[f"{p}.{k}.bias"]).contiguous()


def pack_clip_text_arena(cfg, sd, tdtype, device):
    f = lambda t: t.detach().to(device=device, dtype=torch.float32)  # noqa: E731
    ent: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    ent["token_embedding"] = f(sd["token_embedding.weight"]).contiguous()
    ent["positional_embedding"] = f(sd["positional_embedding"]).contiguous()
    _clip_layers(ent, sd, "transformer", cfg["transformer_layers"], tdtype, f)
    ent["ln_final.weight"] = f(sd["ln_final.weight"]).contiguous(); ent["ln_final.bias"] = f(sd["ln_final.bias"]).contiguous()
    ent["head.weight"] = f(sd["text_projection"]).t().contiguous()          # x @ text_projection == Linear(weight = text_projection^T)
    return _finish_arena(ent, device)


def pack_clip_vision_arena(cfg, sd, tdtype, device):
    f = lambda t: t.detach().to(device=device, dtype=torch.float32)  # noqa: E731
    Wv, p = cfg["vision_width"], cfg["vision_patch_size"]
    K = 3 * p * p
    Kp = (K + 63) // 64 * 64
    ent: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    w = torch.zeros(Wv, Kp, device=device)
    w[:, :K] = f(sd["visual.conv1.weight"]).reshape(Wv, K)
    ent["patch.weight"] = _pad_rows(w).to(tdtype).contiguous()
    ent["class_embedding"] = f(sd["visual.class_embedding"]).contiguous()
    ent["positional_embedding"] = f(sd["visual.positional_embedding"]).contiguous()
    for k in ("ln_pre", "ln_post"):
        ent[k + ".weight"] = f(sd[f"visual.{k}.weight"]).contiguous(); ent[k + ".bias"] = f(sd[f"visual.{k}.bias"]).contiguous()
    _clip_layers(ent, sd, "visual.transformer", cfg["vision_layers"], tdtype, f)
    ent["head.weight"] = f(sd["visual.proj"]).t().contiguous()
    return _finish_arena(ent, device)


def pack_clip_vision_hf_arena(cfg, sd, tdtype, device):
    """transformers CLIPVisionModelWithProjection keys -> the encoder engine's names (q / k / v rows stacked [q | k | v], heads
    contiguous inside each, exactly as the separate q_proj / k_proj / v_proj weights already are)."""
    f = lambda t: t.detach().to(device=device, dtype=torch.float32)  # noqa: E731
    H, p = cfg["hidden_size"], cfg["patch_size"]
    K = 3 * p * p
    Kp = (K + 63) // 64 * 64
    e = "vision_model.embeddings."
    ent: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    w = torch.zeros(H, Kp, device=device)
    w[:, :K] = f(sd[e + "patch_embedding.weight"]).reshape(H, K)
    ent["patch.weight"] = _pad_rows(w).to(tdtype).contiguous()
    ent["class_embedding"] = f(sd[e + "class_embedding"]).contiguous()
    ent["positional_embedding"] = f(sd[e + "position_embedding.weight"]).contiguous()
    ent["ln_pre.weight"] = f(sd["vision_model.pre_layrnorm.weight"]).contiguous(); ent["ln_pre.bias"] = f(sd["vision_model.pre_layrnorm.bias"]).contiguous()
    ent["ln_post.weight"] = f(sd["vision_model.post_layernorm.weight"]).contiguous(); ent["ln_post.bias"] = f(sd["vision_model.post_layernorm.bias"]).contiguous()
    for l in range(cfg["num_hidden_layers"]):
        s_, q = f"vision_model.encoder.layers.{l}.", f"layers.{l}"
        a = s_ + "self_attn."
        ent[q + ".qkv.weight"] = _pad_rows(torch.cat([f(sd[a + "q_proj.weight"]), f(sd[a + "k_proj.weight"]), f(sd[a + "v_proj.weight"])], 0)).to(tdtype).contiguous()
        ent[q + ".qkv.bias"] = torch.cat([f(sd[a + "q_proj.bias"]), f(sd[a + "k_proj.bias"]), f(sd[a + "v_proj.bias"])], 0).contiguous()
        ent[q + ".proj.weight"] = _pad_rows(f(sd[a + "out_proj.weight"])).to(tdtype).contiguous()
        ent[q + ".proj.bias"] = f(sd[a + "out_proj.bias"]).contiguous()
        ent[q + ".fc.weight"] = _pad_rows(f(sd[s_ + "mlp.fc1.weight"])).to(tdtype).contiguous()
        ent[q + ".fc.bias"] = f(sd[s_ + "mlp.fc1.bias"]).contiguous()
        ent[q + ".out.weight"] = _pad_rows(f(sd[s_ + "mlp.fc2.weight"])).to(tdtype).contiguous()
        ent[q + ".out.bias"] = f(sd[s_ + "mlp.fc2.bias"]).contiguous()
        ent[q + ".ln_1.weight"] = f(sd[s_ + "layer_norm1.weight"]).contiguous(); ent[q + ".ln_1.bias"] = f(sd[s_ + "layer_norm1.bias"]).contiguous()
        ent[q + ".ln_2.weight"] = f(sd[s_ + "layer_norm2.weight"]).contiguous(); ent[q + ".ln_2.bias"] = f(sd[s_ + "layer_norm2.bias"]).contiguous()
    ent["head.weight"] = f(sd["visual_projection.weight"]).contiguous()
    return _finish_arena(ent, device)


def pack_multiclip_arena(cfg, sd, tdtype, device):
    f = lambda t: t.detach().to(device=device, dtype=torch.float32)  # noqa: E731
    e = "transformer.embeddings."
    ent: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    ent["token_embedding"] = f(sd[e + "word_embeddings.weight"]).contiguous()
    ent["positional_embedding"] = f(sd[e + "position_embeddings.weight"]).contiguous()
    ent["token_type_embedding"] = f(sd[e + "token_type_embeddings.weight"])[0].contiguous()    # token_type_ids are all zero
    ent["embeddings_ln.weight"] = f(sd[e + "LayerNorm.weight"]).contiguous(); ent["embeddings_ln.bias"] = f(sd[e + "LayerNorm.bias"]).contiguous()
    for l in range(cfg["num_hidden_layers"]):
        p, q = f"transformer.encoder.layer.{l}.", f"layers.{l}"
        a = p + "attention.self."
        ent[q + ".qkv.weight"] = _pad_rows(torch.cat([f(sd[a + "query.weight"]), f(sd[a + "key.weight"]), f(sd[a + "value.weight"])], 0)).to(tdtype).contiguous()
        ent[q + ".qkv.bias"] = torch.cat([f(sd[a + "query.bias"]), f(sd[a + "key.bias"]), f(sd[a + "value.bias"])], 0).contiguous()
        ent[q + ".proj.weight"] = _pad_rows(f(sd[p + "attention.output.dense.weight"])).to(tdtype).contiguous()
        ent[q + ".proj.bias"] = f(sd[p + "attention.output.dense.bias"]).contiguous()
        ent[q + ".ln_1.weight"] = f(sd[p + "attention.output.LayerNorm.weight"]).contiguous()
        ent[q + ".ln_1.bias"] = f(sd[p + "attention.output.LayerNorm.bias"]).contiguous()
        ent[q + ".fc.weight"] = _pad_rows(f(sd[p + "intermediate.dense.weight"])).to(tdtype).contiguous()
        ent[q + ".fc.bias"] = f(sd[p + "intermediate.dense.bias"]).contiguous()
        ent[q + ".out.weight"] = _pad_rows(f(sd[p + "output.dense.weight"])).to(tdtype).contiguous()
        ent[q + ".out.bias"] = f(sd[p + "output.dense.bias"]).contiguous()
        ent[q + ".ln_2.weight"] = f(sd[p + "output.LayerNorm.weight"]).contiguous()
        ent[q + ".ln_2.bias"] = f(sd[p + "output.LayerNorm.bias"]).contiguous()
    ent["head.weight"] = f(sd["LinearTransformation.weight"]).contiguous()
    ent["head.bias"] = f(sd["LinearTransformation.bias"]).contiguous()
    return _finish_arena(ent, device)


class _Engine:
    """One K22Encoder handle: arena + config + a plan per batch size (<= 8 rows per call; larger batches run in chunks)."""

    def __init__(self, ecfg: dict, arena: torch.Tensor, table, backend_dtype):
        self.cfg, self.arena = ecfg, arena
        c = _lib.K22EncoderConfig()
        c.dtype = _lib.dtype_code(backend_dtype)
        for k, v in ecfg.items():
            setattr(c, k, v)
        base = arena.data_ptr()
        arr = (_lib.K22Weight * len(table))()
        self._names = []
        for i, (name, (off, _n)) in enumerate(table.items()):
            nb = name.encode()
            self._names.append(nb)
            arr[i].name = nb
            arr[i].ptr = base + off
        h = C.c_void_p()
        _lib.check(_lib.lib().k22_encoder_create(C.byref(c), arr, len(table), C.byref(h)))
        self.handle, self._plan_B, self._ws = h, None, None

    def close(self):
        if self.handle is not None:
            _lib.lib().k22_encoder_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _plan(self, B):
        if self._plan_B != B:
            self._plan_B = None
            nbytes = C.c_size_t()
            _lib.check(_lib.lib().k22_encoder_plan(self.handle, B, C.byref(nbytes)))
            self._ws = torch.empty(nbytes.value + 256, dtype=torch.uint8, device=self.arena.device)
            al = (self._ws.data_ptr() + 255) // 256 * 256
            _lib.check(_lib.lib().k22_encoder_bind(self.handle, al, nbytes.value))
            self._plan_B = B

    def run(self, tokens=None, key_valid=None, image=None, want_seq=True):
        dev = self.arena.device
        N = (image if tokens is None else tokens).shape[0]
        c = self.cfg
        seqs, pools = [], []
        for s in range(0, N, 8):
            B = min(8, N - s)
            self._plan(B)
            tk = None if tokens is None else tokens[s:s + B].to(device=dev, dtype=torch.int32).contiguous()
            kv = None if key_valid is None else key_valid[s:s + B].to(device=dev, dtype=torch.float32).contiguous()
            im = None if image is None else image[s:s + B].to(device=dev, dtype=torch.float32).contiguous()
            seq = torch.empty(B, c["n_ctx"], c["width"], dtype=torch.float32, device=dev) if (want_seq and image is None) else None
            pooled = torch.empty(B, c["out_dim"], dtype=torch.float32, device=dev)
            _lib.check(_lib.lib().k22_encoder_forward(self.handle, _lib.ptr(tk), _lib.ptr(kv), _lib.ptr(im), _lib.ptr(seq), pooled.data_ptr(),
                                                      _lib.current_stream()))
            seqs.append(seq); pools.append(pooled)
        return (None if seqs[0] is None else torch.cat(seqs, 0)), torch.cat(pools, 0)


class _HIPModule(nn.Module):
    def __init__(self, shapes, backend_dtype):
        super().__init__()
        self.backend_dtype = backend_dtype
        for name, shape in shapes.items():
            _register(self, name, nn.Parameter(torch.zeros(shape), requires_grad=False))
        self._engines: Dict[str, _Engine] = {}

    def _release(self):
        for e in self._engines.values():
            e.close()
        self._engines = {}

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self._release()
        return r

    def _apply(self, fn, *args, **kwargs):
        r = super()._apply(fn, *args, **kwargs)
        self._release()
        return r

    def _device(self):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError(f"{type(self).__name__} runs on the GPU only (no CPU fallback): move it with .to('cuda')")
        return dev

    @property
    def dtype(self):
        return torch.float32        # public tensors; the engine's arithmetic type is backend_dtype


class CLIPModelHIP(_HIPModule):
    """clip.model.CLIP (ViT image tower) for the calls the reference makes: encode_image, encode_text, and the text-tower walk of
    generate_clip_emb exposed as `encode_text_with_sequence(tokens) -> (txt_feat, txt_feat_seq)`."""

    def __init__(self, config: Optional[dict] = None, backend_dtype: torch.dtype = torch.bfloat16):
        self.config = dict(config or CLIP_VITL14)
        c = self.config
        if c["transformer_width"] != 64 * c["transformer_heads"] or c["vision_width"] % 64:
            raise ValueError("CLIPModelHIP: 64 channels per attention head (ViT-L/14: 768 / 12, 1024 / 16)")
        super().__init__(clip_param_shapes(c), backend_dtype)
        self.context_length, self.vocab_size, self.input_resolution = c["context_length"], c["vocab_size"], c["image_resolution"]

    def _engine(self, which):
        if which not in self._engines:
            dev, c, sd = self._device(), self.config, self.state_dict()
            if which == "text":
                arena, table = pack_clip_text_arena(c, sd, self.backend_dtype, dev)
                ecfg = dict(kind=ENC_CLIP_TEXT, width=c["transformer_width"], layers=c["transformer_layers"], heads=c["transformer_heads"],
                            n_ctx=c["context_length"], vocab=c["vocab_size"], out_dim=c["embed_dim"], image_size=0, patch=0, max_pos=0,
                            pad_id=0, ln_eps=1e-5)
            else:
                arena, table = pack_clip_vision_arena(c, sd, self.backend_dtype, dev)
                g = c["image_resolution"] // c["vision_patch_size"]
                ecfg = dict(kind=ENC_CLIP_VISION, width=c["vision_width"], layers=c["vision_layers"], heads=c["vision_width"] // 64,
                            n_ctx=g * g + 1, vocab=0, out_dim=c["embed_dim"], image_size=c["image_resolution"], patch=c["vision_patch_size"],
                            max_pos=0, pad_id=0, ln_eps=1e-5)
            self._engines[which] = _Engine(ecfg, arena, table, self.backend_dtype)
        return self._engines[which]

    def _check_tokens(self, text):
        if text.dim() != 2 or text.shape[1] != self.context_length:
            raise ValueError(f"CLIP tokens must be [n, {self.context_length}], got {tuple(text.shape)}")
        if text.numel() and (int(text.min()) < 0 or int(text.max()) >= self.vocab_size):
            raise ValueError("CLIP token id outside the vocabulary")

    @torch.no_grad()
    def encode_text_with_sequence(self, text):
        """-> (txt_feat [n, embed_dim], txt_feat_seq [n, 77, width]) = (x[arange, text.argmax(-1)] @ text_projection, ln_final(x))."""
        self._check_tokens(text)
        seq, pooled = self._engine("text").run(tokens=text)
        return pooled, seq

    @torch.no_grad()
    def encode_text(self, text):
        return self.encode_text_with_sequence(text)[0]

    @torch.no_grad()
    def encode_image(self, image):
        r = self.input_resolution
        if image.dim() != 4 or tuple(image.shape[1:]) != (3, r, r):
            raise ValueError(f"CLIP image batch must be [n, 3, {r}, {r}] (preprocessed), got {tuple(image.shape)}")
        return self._engine("vision").run(image=image)[1]


class CLIPVisionModelWithProjectionHIP(_HIPModule):
    """transformers' `CLIPVisionModelWithProjection` - the `image_encoder` Kandinsky2_2.__init__ loads from the prior repository
    (kandinsky2/kandinsky2_2_model.py:24; CLIP ViT-bigG/14) and diffusers' prior pipeline calls as
    `image_encoder(pixel_values).image_embeds` (mix_images / interpolate, the zero-image negative embedding) - on the encoder engine.
    Parameters under transformers' own state_dict keys, so `load_state_dict(CLIPVisionModelWithProjection.state_dict())` works unchanged.
    16 heads of 104 channels: attention runs on the engine's generic-head-width kernel, everything else on the shared Linear / LayerNorm
    path.  forward(pixel_values [n,3,224,224], already preprocessed) -> namespace(image_embeds [n, projection_dim])."""

    def __init__(self, config: Optional[dict] = None, backend_dtype: torch.dtype = torch.bfloat16):
        self.config_dict = dict(config or CLIP_BIGG_VISION)
        c = self.config_dict
        if c["hidden_size"] % c["num_attention_heads"] or c["hidden_size"] % 64 or c["intermediate_size"] % 64:
            raise ValueError("CLIPVisionModelWithProjectionHIP: hidden_size % heads == 0, hidden_size and intermediate_size multiples of 64")
        if c.get("hidden_act", "quick_gelu") not in ("gelu", "quick_gelu"):
            raise NotImplementedError("hidden_act: gelu (bigG) or quick_gelu (OpenAI CLIP)")
        super().__init__(clip_vision_hf_param_shapes(c), backend_dtype)
        self.config = type("Config", (), {"image_size": c["image_size"], "projection_dim": c["projection_dim"], "hidden_size": c["hidden_size"]})()

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        sd = {k: v for k, v in state_dict.items() if not k.endswith("position_ids")}        # buffer of older transformers versions
        return super().load_state_dict(sd, strict=strict, **kw)

    def _engine(self):
        if "vision" not in self._engines:
            dev, c = self._device(), self.config_dict
            arena, table = pack_clip_vision_hf_arena(c, self.state_dict(), self.backend_dtype, dev)
            g = c["image_size"] // c["patch_size"]
            ecfg = dict(kind=ENC_CLIP_VISION, width=c["hidden_size"], layers=c["num_hidden_layers"], heads=c["num_attention_heads"], n_ctx=g * g + 1,
                        vocab=0, out_dim=c["projection_dim"], image_size=c["image_size"], patch=c["patch_size"], max_pos=0, pad_id=0,
                        ln_eps=float(c.get("layer_norm_eps", 1e-5)), mlp_dim=c["intermediate_size"],
                        hidden_act=1 if c.get("hidden_act", "quick_gelu") == "gelu" else 0)
            self._engines["vision"] = _Engine(ecfg, arena, table, self.backend_dtype)
        return self._engines["vision"]

    @torch.no_grad()
    def forward(self, pixel_values, **_unused):
        r = self.config_dict["image_size"]
        if pixel_values.dim() != 4 or tuple(pixel_values.shape[1:]) != (3, r, r):
            raise ValueError(f"pixel_values must be [n, 3, {r}, {r}] (preprocessed), got {tuple(pixel_values.shape)}")
        from types import SimpleNamespace
        return SimpleNamespace(image_embeds=self._engine().run(image=pixel_values)[1])


class MultilingualCLIPHIP(_HIPModule):
    """MultilingualCLIP (text_encoders.py:108-122): forward(input_ids, attention_mask) -> (LinearTransformation(masked mean), embs)."""

    def __init__(self, config: Optional[dict] = None, in_features=1024, out_features=768, backend_dtype: torch.dtype = torch.bfloat16):
        self.config = dict(config or XLMR_LARGE)
        c = self.config
        if c["hidden_size"] != 64 * c["num_attention_heads"] or c["intermediate_size"] != 4 * c["hidden_size"] or in_features != c["hidden_size"]:
            raise ValueError("MultilingualCLIPHIP: 64 channels per head, intermediate = 4 x hidden, in_features = hidden (xlm-roberta-large)")
        if c.get("hidden_act", "gelu") != "gelu" or c.get("position_embedding_type", "absolute") != "absolute":
            raise NotImplementedError("MultilingualCLIPHIP: erf-GELU, absolute position embeddings (xlm-roberta-large)")
        self.out_features = out_features
        self._arena_table = None
        super().__init__(multiclip_param_shapes(c, in_features, out_features), backend_dtype)

    def _engine_for(self, n_ctx):
        """one engine per sequence length (the reference pads to 77), all on one packed arena"""
        if self._arena_table is None:
            self._arena_table = pack_multiclip_arena(self.config, self.state_dict(), self.backend_dtype, self._device())
        arena, table = self._arena_table
        key = f"xlmr{n_ctx}"
        if key not in self._engines:
            c = self.config
            ecfg = dict(kind=ENC_XLMR, width=c["hidden_size"], layers=c["num_hidden_layers"], heads=c["num_attention_heads"], n_ctx=n_ctx,
                        vocab=c["vocab_size"], out_dim=self.out_features, image_size=0, patch=0, max_pos=c["max_position_embeddings"],
                        pad_id=c["pad_token_id"], ln_eps=float(c["layer_norm_eps"]))
            self._engines[key] = _Engine(ecfg, arena, table, self.backend_dtype)
        return self._engines[key]

    def _release(self):
        super()._release()
        self._arena_table = None

    @torch.no_grad()
    def forward(self, input_ids, attention_mask):
        c = self.config
        if input_ids.dim() != 2 or attention_mask.shape != input_ids.shape:
            raise ValueError("MultilingualCLIPHIP: input_ids and attention_mask must both be [n, tokens]")
        n_ctx = input_ids.shape[1]
        if n_ctx + c["pad_token_id"] + 1 > c["max_position_embeddings"]:
            raise ValueError("MultilingualCLIPHIP: sequence longer than the position-embedding table")
        if input_ids.numel() and (int(input_ids.min()) < 0 or int(input_ids.max()) >= c["vocab_size"]):
            raise ValueError("XLM-R token id outside the vocabulary")
        embs, pooled = self._engine_for(n_ctx).run(tokens=input_ids, key_valid=attention_mask)
        return pooled, embs


class TextEncoderHIP(nn.Module):
    """TextEncoder (text_encoders.py:125-167) for the configuration Kandinsky 2.1 ships: model_name == "multiclip"."""

    def __init__(self, model_path="", model_name="multiclip", *, xlmr_config: Optional[dict] = None, state_dict=None,
                 backend_dtype: torch.dtype = torch.bfloat16, **kwargs):
        super().__init__()
        if model_name != "multiclip":
            raise NotImplementedError("TextEncoderHIP: only model_name='multiclip' (CONFIG_2_1 text_enc_params) is built")
        self.model_name = model_name
        self.model = MultilingualCLIPHIP(xlmr_config, backend_dtype=backend_dtype, **kwargs)
        if state_dict is not None:
            self.model.load_state_dict(state_dict, strict=False)       # as the reference (text_encoders.py:140-142)
        self.model.eval()

    def forward(self, tokens, mask=None):
        pooled_out, full_out = self.model(input_ids=tokens, attention_mask=mask)
        return full_out, pooled_out


class HIPConditioner:
    """The conditioner of pipeline.Kandinsky2_1HIP on the HIP towers: what Kandinsky2_1.encode_text (:117-131),
    generate_clip_emb up to the prior call (:133-170) and encode_images (:177-181) compute.
    tokenizer1: the XLM-R tokenizer (transformers AutoTokenizer call signature); tokenizer2: the reference's CustomizedTokenizer
    (padded_tokens_and_mask); preprocess: clip's PIL -> tensor transform (needed only for PIL images)."""

    def __init__(self, text_encoder: TextEncoderHIP, tokenizer1, tokenizer2, clip_model: CLIPModelHIP, preprocess=None, text_ctx=77):
        self.text_encoder, self.tokenizer1, self.tokenizer2 = text_encoder, tokenizer1, tokenizer2
        self.clip_model, self.preprocess, self.text_ctx = clip_model, preprocess, text_ctx

    @torch.no_grad()
    def encode_text(self, prompt, batch_size, device):
        enc = self.tokenizer1([prompt] * batch_size + [""] * batch_size, max_length=77, padding="max_length", truncation=True,
                              return_attention_mask=True, add_special_tokens=True, return_tensors="pt")
        full_emb, pooled_emb = self.text_encoder(tokens=enc["input_ids"].to(device), mask=enc["attention_mask"].to(device))
        return full_emb, pooled_emb

    @torch.no_grad()
    def clip_text(self, prompts, negative_prompt, device):
        tok, mask = self.tokenizer2.padded_tokens_and_mask(list(prompts), self.text_ctx)
        cf_token, cf_mask = self.tokenizer2.padded_tokens_and_mask([negative_prompt], self.text_ctx)
        if cf_token.shape != tok.shape:
            cf_token, cf_mask = cf_token.expand(tok.shape[0], -1), cf_mask.expand(tok.shape[0], -1)
        tok, mask = torch.cat([tok, cf_token], 0).to(device), torch.cat([mask, cf_mask], 0).to(device)
        txt_feat, txt_feat_seq = self.clip_model.encode_text_with_sequence(tok)
        return txt_feat, txt_feat_seq, mask

    @torch.no_grad()
    def encode_image(self, image, device):
        if not torch.is_tensor(image):
            if self.preprocess is None:
                raise ValueError("HIPConditioner: a PIL image needs the clip `preprocess` transform")
            image = self.preprocess(image).unsqueeze(0)
        return self.clip_model.encode_image(image.to(device))

    def zero_image_emb(self, device):
        r = self.clip_model.input_resolution
        return self.encode_image(torch.zeros(1, 3, r, r), device)
