"""CPU suite for the conditioning encoders: the oracle restatements against fixtures made from the REFERENCE's
MultilingualCLIP (on the installed transformers) and from transformers' CLIP port (oracle/make_golden_encoders.py), the state-dict
key tables against the reference's, the arena layouts, and the host-side guards."""
import json
import math
import os

import pytest
import torch

import kandinsky2_amd as k22
from kandinsky2_amd import encoders
from oracle import encoders_ref

SIZES = ["tiny", "full"]


def _fx(golden_dir, name):
    p = os.path.join(golden_dir, name + ".pt")
    if not os.path.exists(p):
        pytest.skip(f"{name}.pt not generated")
    return torch.load(p, weights_only=False)


def _rel(a, b):
    return (a - b).abs().max().item() / b.abs().max().item()


@pytest.mark.parametrize("size", SIZES)
def test_multiclip_oracle_matches_reference_golden(golden_dir, size):
    fx = _fx(golden_dir, f"enc_multiclip_{size}")
    m = fx["meta"]
    sd = k22.init_multiclip_state_dict(m["cfg"], m["in_features"], m["out_features"], seed=m["seed_w"])
    pooled, embs = encoders_ref.multiclip_forward(sd, m["cfg"], fx["input_ids"], fx["attention_mask"])
    assert _rel(embs, fx["embs"]) <= 2e-5 and _rel(pooled, fx["pooled"]) <= 2e-5


@pytest.mark.parametrize("size", SIZES)
def test_clip_oracle_matches_golden(golden_dir, size):
    fx = _fx(golden_dir, f"enc_clip_{size}")
    cfg = fx["meta"]["cfg"]
    sd = k22.init_clip_state_dict(cfg, seed=fx["meta"]["seed_w"])
    feat, seq = encoders_ref.clip_text_forward(sd, cfg, fx["tokens"])
    img = encoders_ref.clip_image_forward(sd, cfg, fx["image"])
    assert _rel(seq, fx["txt_feat_seq"]) <= 2e-5 and _rel(feat, fx["txt_feat"]) <= 2e-5 and _rel(img, fx["img_feat"]) <= 2e-5


@pytest.mark.parametrize("name", ["enc_clipvision_hf_tiny", "enc_clipvision_bigg"])
def test_clip_vision_hf_oracle_matches_transformers_golden(golden_dir, name):
    """oracle/encoders_ref.clip_vision_hf_forward against the fixture transformers' own CLIPVisionModelWithProjection produced
    (kandinsky2_2_model.py:24 loads that class as the 2.2 image encoder: CLIP ViT-bigG/14)."""
    fx = _fx(golden_dir, name)
    cfg = fx["meta"]["cfg"]
    if cfg["num_hidden_layers"] > 8 and not os.environ.get("K22_SLOW_CPU_TESTS"):
        # the 1.8 B-parameter tower takes ~2 min on 8 cores (weights + forward): checked by make_golden_encoders.py when the fixture is made
        assert fx["image_embeds"].shape == (1, 1280) and cfg["hidden_size"] // cfg["num_attention_heads"] == 104
        return
    sd = k22.init_clip_vision_hf_state_dict(cfg, seed=fx["meta"]["seed_w"])
    assert _rel(encoders_ref.clip_vision_hf_forward(sd, cfg, fx["image"]), fx["image_embeds"]) <= 2e-5


def test_clip_bigg_vision_is_the_published_tower():
    shapes = k22.clip_vision_hf_param_shapes(k22.CLIP_BIGG_VISION)
    n = sum(math.prod(s) for s in shapes.values())
    assert 1.84e9 < n < 1.85e9                                   # ViT-bigG/14 vision tower + 1664 -> 1280 projection
    assert shapes["vision_model.encoder.layers.47.mlp.fc1.weight"] == (8192, 1664) and shapes["visual_projection.weight"] == (1280, 1664)
    assert shapes["vision_model.embeddings.position_embedding.weight"] == (257, 1664)
    m = k22.CLIPVisionModelWithProjectionHIP(k22.tiny_clip_vision_hf_config())
    with pytest.raises(RuntimeError):                            # no CPU fallback
        m(torch.zeros(1, 3, 56, 56))


def test_multiclip_state_dict_keys_match_reference(golden_dir):
    """ref_multiclip_keys.json: state_dict of the reference's MultilingualCLIP built on xlm-roberta-large's config."""
    p = os.path.join(golden_dir, "ref_multiclip_keys.json")
    if not os.path.exists(p):
        pytest.skip("ref_multiclip_keys.json not generated")
    ref = {k: tuple(v) for k, v in json.load(open(p)).items()}
    ours = {k: tuple(v) for k, v in k22.multiclip_param_shapes(k22.XLMR_LARGE).items()}
    assert ours == ref


def test_parameter_counts_are_the_published_models():
    n_clip = sum(math.prod(s) for s in k22.clip_param_shapes(k22.CLIP_VITL14).values())
    n_xlmr = sum(math.prod(s) for k, s in k22.multiclip_param_shapes(k22.XLMR_LARGE).items() if not k.startswith("LinearTransformation"))
    assert n_clip == 427_616_513          # OpenAI CLIP ViT-L/14
    assert n_xlmr == 559_890_432          # xlm-roberta-large (with pooler)


def test_arena_layouts():
    cfg = k22.tiny_clip_config()
    sd = k22.init_clip_state_dict(cfg, seed=1)
    arena, table = encoders.pack_clip_text_arena(cfg, sd, torch.float32, "cpu")
    W, E = cfg["transformer_width"], cfg["embed_dim"]

    def get(name, shape, dtype=torch.float32):
        o, nb = table[name]
        return arena[o:o + nb].view(dtype).reshape(shape)

    assert torch.equal(get("head.weight", (E, W)), sd["text_projection"].t())
    assert torch.equal(get("layers.1.qkv.weight", (3 * W, W)), sd["transformer.resblocks.1.attn.in_proj_weight"])
    assert all(o % 256 == 0 for o, _ in table.values())
    arena, table = encoders.pack_clip_vision_arena(cfg, sd, torch.float32, "cpu")
    Wv, p = cfg["vision_width"], cfg["vision_patch_size"]
    K = 3 * p * p
    Kp = (K + 63) // 64 * 64
    pw = get("patch.weight", (Wv, Kp))
    assert torch.equal(pw[:, :K], sd["visual.conv1.weight"].reshape(Wv, K)) and (pw[:, K:] == 0).all()
    xc = k22.tiny_xlmr_config()
    xsd = k22.init_multiclip_state_dict(xc, 128, 64, seed=2)
    arena, table = encoders.pack_multiclip_arena(xc, xsd, torch.float32, "cpu")
    H = xc["hidden_size"]
    qkv = get("layers.0.qkv.weight", (3 * H, H))
    a = "transformer.encoder.layer.0.attention.self."
    assert torch.equal(qkv, torch.cat([xsd[a + "query.weight"], xsd[a + "key.weight"], xsd[a + "value.weight"]], 0))
    assert torch.equal(get("token_type_embedding", (H,)), xsd["transformer.embeddings.token_type_embeddings.weight"][0])


def test_encoders_fail_loudly_without_gpu_and_validate_inputs():
    m = k22.CLIPModelHIP(k22.tiny_clip_config())
    m.load_state_dict(k22.init_clip_state_dict(k22.tiny_clip_config()))
    with pytest.raises(RuntimeError):
        m.encode_text(torch.zeros(1, 77, dtype=torch.long))
    with pytest.raises(ValueError):
        m.encode_text(torch.zeros(1, 76, dtype=torch.long))
    with pytest.raises(ValueError):
        m.encode_text(torch.full((1, 77), 5000, dtype=torch.long))
    with pytest.raises(ValueError):
        m.encode_image(torch.zeros(1, 3, 32, 32))
    x = k22.MultilingualCLIPHIP(k22.tiny_xlmr_config(), in_features=128, out_features=64)
    with pytest.raises(ValueError):
        x(torch.zeros(2, 77, dtype=torch.long), torch.ones(2, 76))
    with pytest.raises(RuntimeError):
        x(torch.zeros(2, 77, dtype=torch.long), torch.ones(2, 77))
    with pytest.raises(NotImplementedError):
        k22.TextEncoderHIP(model_name="T5EncoderModel")
    with pytest.raises(ValueError):
        k22.MultilingualCLIPHIP(dict(k22.tiny_xlmr_config(), intermediate_size=256), in_features=128, out_features=64)


def test_encoder_c_abi_argument_checks():
    """k22_encoder_create / plan / forward reject bad configurations with an error code and message (host logic, no GPU work)."""
    import ctypes as C
    from kandinsky2_amd import _lib
    L = _lib.lib()

    def plan(B=2, **kw):
        c = _lib.K22EncoderConfig()
        vals = dict(dtype=_lib.K22_BF16, kind=encoders.ENC_XLMR, width=128, layers=2, heads=2, n_ctx=77, vocab=1000, out_dim=64, image_size=0,
                    patch=0, max_pos=514, pad_id=1, ln_eps=1e-5)
        vals.update(kw)
        for k, v in vals.items():
            setattr(c, k, v)
        h, n = C.c_void_p(), C.c_size_t()
        assert L.k22_encoder_create(C.byref(c), None, 0, C.byref(h)) == 0
        try:
            rc = L.k22_encoder_plan(h, B, C.byref(n))
            fwd = L.k22_encoder_forward(h, None, None, None, None, None, None)
            return rc, (L.k22_last_error() or b"").decode(), fwd
        finally:
            L.k22_encoder_destroy(h)

    assert plan(B=9)[0] != 0 and plan(B=0)[0] != 0
    assert plan(heads=3)[0] != 0 and plan(width=96, heads=1)[0] != 0
    assert plan(kind=encoders.ENC_CLIP_VISION, image_size=56, patch=0, n_ctx=17)[0] != 0          # no division by a zero patch
    assert plan(kind=encoders.ENC_CLIP_VISION, image_size=56, patch=14, n_ctx=18)[0] != 0         # n_ctx != 16 patches + class token
    assert plan(kind=7)[0] != 0 and plan(max_pos=1)[0] != 0
    rc, msg, fwd = plan()                                             # a valid configuration without weights: named, not dereferenced
    assert rc != 0 and fwd != 0
    c = _lib.K22EncoderConfig()
    c.dtype = 5
    h = C.c_void_p()
    assert L.k22_encoder_create(C.byref(c), None, 0, C.byref(h)) != 0 and L.k22_encoder_create(None, None, 0, C.byref(h)) != 0
    assert L.k22_blend_noised(None, None, None, None, 1.0, 0.0, None, 1, 4, 16, 0, None) != 0
