"""GPU parity of the conditioning-encoder engine (k22_encoder_* through CLIPModelHIP / MultilingualCLIPHIP / TextEncoderHIP /
HIPConditioner) against fixtures made from the REFERENCE's MultilingualCLIP (running transformers' XLMRobertaModel) and from
transformers' CLIP port holding the same weights (oracle/make_golden_encoders.py; oracle/encoders_ref.py explains the pinning).

Tolerances: fp32 engine 2e-4 of the output scale (exact-fp32 MFMA, different summation order); bf16 engine 3e-2 of the scale on the
tiny towers, 5e-2 on the 24-layer production towers (bf16 weights and GEMM operands, fp32 residual stream and LayerNorm; the
reference itself runs these towers in fp16 when use_fp16 is set, kandinsky2_1_model.py:60-62).
"""
import os

import pytest
import torch

import kandinsky2_amd as k22

pytestmark = pytest.mark.gpu

BACKENDS = [(torch.float32, 2e-4, 2e-4), (torch.bfloat16, 3e-2, 5e-2)]


def _fx(golden_dir, name):
    p = os.path.join(golden_dir, name + ".pt")
    if not os.path.exists(p):
        pytest.skip(f"{name}.pt not generated")
    return torch.load(p, weights_only=False)


def _rel(a, b):
    return (a.float().cpu() - b).abs().max().item() / b.abs().max().item()


@pytest.mark.parametrize("size", ["tiny", "full"])
@pytest.mark.parametrize("backend,tol_tiny,tol_full", BACKENDS)
def test_multiclip_vs_reference_golden(golden_dir, size, backend, tol_tiny, tol_full):
    fx = _fx(golden_dir, f"enc_multiclip_{size}")
    m = fx["meta"]
    te = k22.TextEncoderHIP(model_name="multiclip", xlmr_config=m["cfg"], in_features=m["in_features"], out_features=m["out_features"],
                            state_dict=k22.init_multiclip_state_dict(m["cfg"], m["in_features"], m["out_features"], seed=m["seed_w"]),
                            backend_dtype=backend).to("cuda")
    full_out, pooled_out = te(tokens=fx["input_ids"].cuda(), mask=fx["attention_mask"].cuda())
    e1, e2 = _rel(full_out, fx["embs"]), _rel(pooled_out, fx["pooled"])
    print(f"multiclip {size} {backend}: embs {e1:.3e} pooled {e2:.3e} of scale (reference MultilingualCLIP golden)")
    tol = tol_tiny if size == "tiny" else tol_full
    assert full_out.shape == fx["embs"].shape and pooled_out.shape == fx["pooled"].shape and e1 <= tol and e2 <= tol
    # a second call with another batch size re-plans; rows are independent of their batch neighbours
    f2, p2 = te(tokens=fx["input_ids"][:1].cuda(), mask=fx["attention_mask"][:1].cuda())
    assert _rel(f2, fx["embs"][:1]) <= tol and _rel(p2, fx["pooled"][:1]) <= tol


@pytest.mark.parametrize("size", ["tiny", "full"])
@pytest.mark.parametrize("backend,tol_tiny,tol_full", BACKENDS)
def test_clip_towers_vs_golden(golden_dir, size, backend, tol_tiny, tol_full):
    fx = _fx(golden_dir, f"enc_clip_{size}")
    cfg = fx["meta"]["cfg"]
    m = k22.CLIPModelHIP(cfg, backend_dtype=backend)
    m.load_state_dict(k22.init_clip_state_dict(cfg, seed=fx["meta"]["seed_w"]))
    m = m.to("cuda")
    feat, seq = m.encode_text_with_sequence(fx["tokens"].cuda())
    img = m.encode_image(fx["image"].cuda())
    e = _rel(seq, fx["txt_feat_seq"]), _rel(feat, fx["txt_feat"]), _rel(img, fx["img_feat"])
    print(f"clip {size} {backend}: txt_feat_seq {e[0]:.3e} txt_feat {e[1]:.3e} img_feat {e[2]:.3e} of scale")
    tol = tol_tiny if size == "tiny" else tol_full
    assert max(e) <= tol
    assert _rel(m.encode_text(fx["tokens"][1:2].cuda()), fx["txt_feat"][1:2]) <= tol


def test_more_than_eight_rows_run_in_chunks(golden_dir):
    fx = _fx(golden_dir, "enc_clip_tiny")
    cfg = fx["meta"]["cfg"]
    m = k22.CLIPModelHIP(cfg, backend_dtype=torch.float32)
    m.load_state_dict(k22.init_clip_state_dict(cfg, seed=fx["meta"]["seed_w"]))
    m = m.to("cuda")
    tok = fx["tokens"].repeat(4, 1)[:11]
    feat, seq = m.encode_text_with_sequence(tok.cuda())
    assert feat.shape[0] == 11 and _rel(feat, fx["txt_feat"].repeat(4, 1)[:11]) <= 2e-4 and _rel(seq, fx["txt_feat_seq"].repeat(4, 1, 1)[:11]) <= 2e-4


class _Tok1:
    """stands in for the XLM-R AutoTokenizer call: returns the fixture's ids for [prompt]*bs + [""]*bs"""

    def __init__(self, ids, mask):
        self.ids, self.mask = ids, mask

    def __call__(self, texts, **kw):
        assert kw["max_length"] == 77 and kw["padding"] == "max_length" and kw["return_tensors"] == "pt"
        rows = [0 if t else 1 for t in texts]                      # row 0: full-length prompt, row 1: the empty prompt
        return {"input_ids": self.ids[rows].long(), "attention_mask": self.mask[rows].long()}


class _Tok2:
    def __init__(self, tok):
        self.tok = tok

    def padded_tokens_and_mask(self, texts, ctx):
        rows = [0 if t else 1 for t in texts]
        t = self.tok[rows].long()
        return t, t.ne(0)


def test_hip_conditioner_calls(golden_dir):
    """HIPConditioner (the conditioner interface of pipeline.Kandinsky2_1HIP): encode_text / clip_text / zero_image_emb on the HIP
    towers, with stand-in tokenizers that return the fixtures' token rows."""
    fxm, fxc = _fx(golden_dir, "enc_multiclip_tiny"), _fx(golden_dir, "enc_clip_tiny")
    mm, cfgc = fxm["meta"], fxc["meta"]["cfg"]
    te = k22.TextEncoderHIP(xlmr_config=mm["cfg"], in_features=mm["in_features"], out_features=mm["out_features"],
                            state_dict=k22.init_multiclip_state_dict(mm["cfg"], mm["in_features"], mm["out_features"], seed=mm["seed_w"]),
                            backend_dtype=torch.float32).to("cuda")
    clip = k22.CLIPModelHIP(cfgc, backend_dtype=torch.float32)
    clip.load_state_dict(k22.init_clip_state_dict(cfgc, seed=fxc["meta"]["seed_w"]))
    cond = k22.HIPConditioner(te, _Tok1(fxm["input_ids"], fxm["attention_mask"]), _Tok2(fxc["tokens"]), clip.to("cuda"))
    full, pooled = cond.encode_text("a cat", 2, "cuda")
    assert full.shape == (4, 77, 128) and _rel(full[0:1], fxm["embs"][0:1]) <= 2e-4 and _rel(full[3:4], fxm["embs"][1:2]) <= 2e-4
    assert _rel(pooled[1:2], fxm["pooled"][0:1]) <= 2e-4
    feat, seq, mask = cond.clip_text(["a cat"], "", "cuda")
    assert feat.shape == (2, 64) and mask.shape == (2, 77) and _rel(feat, fxc["txt_feat"][:2]) <= 2e-4 and _rel(seq, fxc["txt_feat_seq"][:2]) <= 2e-4
    z = cond.zero_image_emb("cuda")
    assert z.shape == (1, 64) and torch.isfinite(z).all()
