"""TEST INFRASTRUCTURE ONLY - CPU restatement (PyTorch fp32) of the conditioning encoders of Kandinsky 2.1:

    multiclip_forward   MultilingualCLIP.forward (kandinsky2/model/text_encoders.py:108-122): transformers' XLMRobertaModel
                        (modeling_xlm_roberta.py: XLMRobertaEmbeddings with create_position_ids_from_input_ids, 24 x
                        [self-attention -> SelfOutput(dense, LayerNorm(x + .)) -> Intermediate(dense, erf GELU) ->
                        Output(dense, LayerNorm(x + .))]), masked mean over tokens, LinearTransformation.
    clip_text_forward   the text tower as Kandinsky2_1.generate_clip_emb walks it (kandinsky2/kandinsky2_1_model.py:159-168) over
                        OpenAI clip's CLIP (clip/model.py: token_embedding + positional_embedding, Transformer of
                        ResidualAttentionBlock(ln_1, nn.MultiheadAttention with the causal mask, ln_2, c_fc -> QuickGELU -> c_proj),
                        ln_final, x[arange, argmax(tokens)] @ text_projection).
    clip_image_forward  clip_model.encode_image (kandinsky2_1_model.py:177-181; clip/model.py VisionTransformer.forward: conv1,
                        class_embedding, positional_embedding, ln_pre, Transformer, ln_post(x[:, 0]) @ proj).

    clip_vision_hf_forward  transformers' CLIPVisionModelWithProjection.forward(...).image_embeds (models/clip/modeling_clip.py:
                        CLIPVisionEmbeddings -> pre_layrnorm -> CLIPEncoderLayer x N [layer_norm1, CLIPAttention (q * hd^-0.5), layer_norm2,
                        CLIPMLP fc1 -> act -> fc2] -> post_layernorm(x[:, 0]) -> visual_projection) under transformers' own keys: the
                        CLIP ViT-bigG/14 image encoder Kandinsky2_2.__init__ loads (kandinsky2/kandinsky2_2_model.py:24).  Pinned against
                        the installed transformers class itself (the very dependency the reference calls) on seeded weights.

PINNING (oracle/make_golden_encoders.py, fixtures tests/golden/enc_*.pt):
  * multiclip_forward is checked against the REFERENCE'S OWN MultilingualCLIP class (imported from /root/reference, running the
    installed transformers' XLMRobertaModel - the very dependency the reference calls) on seeded weights: pinned.
  * the OpenAI `clip` package is a dependency of the reference that is neither vendored nor installed here (requirements: the git
    URL of openai/CLIP, unpinned).  clip_text_forward / clip_image_forward restate clip/model.py and are checked against
    transformers' CLIPTextModelWithProjection / CLIPVisionModelWithProjection (hidden_act="quick_gelu") - the published port of the
    same checkpoint family, loaded with the same seeded weights through the OpenAI -> HF key map of convert_clip_original_pytorch_
    to_hf.py.  That pins the arithmetic to an independent implementation of the same model, not to the clip package itself.
State-dict keys are the reference's (OpenAI clip / MultilingualCLIP), so these functions also check the engine's key mapping.
"""
import torch
import torch.nn.functional as F


def _ln(sd, name, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def _mha(x, w_qkv, b_qkv, w_o, b_o, heads, add_mask=None):
    """softmax(q k^T / sqrt(d) + mask) v over `heads` heads; w_qkv rows are [q | k | v]."""
    B, T, W = x.shape
    d = W // heads
    q, k, v = F.linear(x, w_qkv, b_qkv).split(W, dim=-1)
    sp = lambda t: t.reshape(B, T, heads, d).permute(0, 2, 1, 3)  # noqa: E731
    s = torch.matmul(sp(q), sp(k).transpose(-1, -2)) * d ** -0.5
    if add_mask is not None:
        s = s + add_mask
    o = torch.matmul(torch.softmax(s, dim=-1), sp(v)).permute(0, 2, 1, 3).reshape(B, T, W)
    return F.linear(o, w_o, b_o)


def _clip_block(sd, p, x, heads, mask):
    h = _ln(sd, p + ".ln_1", x)
    x = x + _mha(h, sd[p + ".attn.in_proj_weight"], sd[p + ".attn.in_proj_bias"], sd[p + ".attn.out_proj.weight"], sd[p + ".attn.out_proj.bias"],
                 heads, mask)
    h = _lin(sd, p + ".mlp.c_fc", _ln(sd, p + ".ln_2", x))
    h = h * torch.sigmoid(1.702 * h)                       # QuickGELU
    return x + _lin(sd, p + ".mlp.c_proj", h)


@torch.no_grad()
def clip_text_forward(sd, cfg, tok):
    """-> (txt_feat [n, embed_dim], txt_feat_seq [n, ctx, width])"""
    n_ctx = cfg["context_length"]
    x = sd["token_embedding.weight"][tok.long()] + sd["positional_embedding"]
    mask = torch.full((n_ctx, n_ctx), float("-inf")).triu_(1)
    for l in range(cfg["transformer_layers"]):
        x = _clip_block(sd, f"transformer.resblocks.{l}", x, cfg["transformer_heads"], mask)
    x = _ln(sd, "ln_final", x)
    return x[torch.arange(x.shape[0]), tok.argmax(dim=-1)] @ sd["text_projection"], x


@torch.no_grad()
def clip_image_forward(sd, cfg, image):
    Wv, p = cfg["vision_width"], cfg["vision_patch_size"]
    x = F.conv2d(image.float(), sd["visual.conv1.weight"], stride=p)                      # [n, W, g, g]
    x = x.reshape(x.shape[0], Wv, -1).permute(0, 2, 1)
    x = torch.cat([sd["visual.class_embedding"].expand(x.shape[0], 1, Wv), x], dim=1) + sd["visual.positional_embedding"]
    x = _ln(sd, "visual.ln_pre", x)
    for l in range(cfg["vision_layers"]):
        x = _clip_block(sd, f"visual.transformer.resblocks.{l}", x, Wv // 64, None)
    return _ln(sd, "visual.ln_post", x[:, 0]) @ sd["visual.proj"]


@torch.no_grad()
def multiclip_forward(sd, cfg, input_ids, attention_mask):
    """-> (LinearTransformation(masked mean) [n, out], embs = last_hidden_state [n, T, hidden])"""
    pad, eps, heads = cfg["pad_token_id"], cfg["layer_norm_eps"], cfg["num_attention_heads"]
    e = "transformer.embeddings."
    ids = input_ids.long()
    nonpad = ids.ne(pad).int()
    pos = (torch.cumsum(nonpad, dim=1) * nonpad).long() + pad                               # create_position_ids_from_input_ids
    x = sd[e + "word_embeddings.weight"][ids] + sd[e + "token_type_embeddings.weight"][0] + sd[e + "position_embeddings.weight"][pos]
    x = _ln(sd, e + "LayerNorm", x, eps)
    am = attention_mask.float()
    add = (1.0 - am)[:, None, None, :] * torch.finfo(torch.float32).min                      # get_extended_attention_mask
    for l in range(cfg["num_hidden_layers"]):
        p = f"transformer.encoder.layer.{l}."
        a = p + "attention.self."
        w = torch.cat([sd[a + "query.weight"], sd[a + "key.weight"], sd[a + "value.weight"]], 0)
        b = torch.cat([sd[a + "query.bias"], sd[a + "key.bias"], sd[a + "value.bias"]], 0)
        h = _mha(x, w, b, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"], heads, add)
        x = _ln(sd, p + "attention.output.LayerNorm", h + x, eps)
        h = _lin(sd, p + "output.dense", F.gelu(_lin(sd, p + "intermediate.dense", x)))
        x = _ln(sd, p + "output.LayerNorm", h + x, eps)
    pooled = (x * am.unsqueeze(2)).sum(dim=1) / am.sum(dim=1)[:, None]
    return _lin(sd, "LinearTransformation", pooled), x


@torch.no_grad()
def clip_vision_hf_forward(sd, cfg, pixel_values):
    """-> image_embeds [n, projection_dim]"""
    H, p, heads = cfg["hidden_size"], cfg["patch_size"], cfg["num_attention_heads"]
    eps = cfg.get("layer_norm_eps", 1e-5)
    e = "vision_model.embeddings."
    x = F.conv2d(pixel_values.float(), sd[e + "patch_embedding.weight"], stride=p)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([sd[e + "class_embedding"].expand(x.shape[0], 1, H), x], dim=1) + sd[e + "position_embedding.weight"]
    x = _ln(sd, "vision_model.pre_layrnorm", x, eps)
    for l in range(cfg["num_hidden_layers"]):
        q = f"vision_model.encoder.layers.{l}."
        a = q + "self_attn."
        w = torch.cat([sd[a + "q_proj.weight"], sd[a + "k_proj.weight"], sd[a + "v_proj.weight"]], 0)
        b = torch.cat([sd[a + "q_proj.bias"], sd[a + "k_proj.bias"], sd[a + "v_proj.bias"]], 0)
        x = x + _mha(_ln(sd, q + "layer_norm1", x, eps), w, b, sd[a + "out_proj.weight"], sd[a + "out_proj.bias"], heads)
        h = _lin(sd, q + "mlp.fc1", _ln(sd, q + "layer_norm2", x, eps))
        h = F.gelu(h) if cfg.get("hidden_act", "quick_gelu") == "gelu" else h * torch.sigmoid(1.702 * h)
        x = x + _lin(sd, q + "mlp.fc2", h)
    return F.linear(_ln(sd, "vision_model.post_layernorm", x[:, 0], eps), sd["visual_projection.weight"])
