"""Times the conditioning-encoder towers at their production sizes (once-per-prompt work, outside bench.py's per-step metric):
CLIP ViT-L/14 text tower and XLM-R-large MultilingualCLIP on the CFG pair of one prompt (2 x 77 tokens), CLIP image tower on one
224-px image.  Prints ms per call and the weight-stream rate (these towers are weight-bandwidth bound at 154 token rows).

    python tools/bench_encoders.py [--dtype bf16|fp32] [--reps 20]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kandinsky2_amd as k22  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[a.dtype]
    es = 4 if a.dtype == "fp32" else 2
    out = {}
    xcfg = dict(k22.XLMR_LARGE, vocab_size=32768)     # the vocabulary only feeds a gather: 32 k rows keep the host-side init short
    te = k22.TextEncoderHIP(xlmr_config=xcfg, state_dict=k22.init_multiclip_state_dict(xcfg, seed=0), backend_dtype=dt).to("cuda")
    ids = torch.randint(3, 32768, (2, 77), device="cuda")
    ids[:, 0] = 0; ids[0, 30] = 2; ids[0, 31:] = 1; ids[1, 1] = 2; ids[1, 2:] = 1
    am = ids.ne(1).long()
    ms = timed(lambda: te(tokens=ids, mask=am), a.reps)
    wb = 24 * 12 * 1024 * 1024 * es
    out["xlmr_large_2x77"] = {"ms": round(ms, 3), "linear_weight_GB_per_s": round(wb / ms / 1e6, 1)}
    clip = k22.CLIPModelHIP(k22.CLIP_VITL14, backend_dtype=dt)
    clip.load_state_dict(k22.init_clip_state_dict(k22.CLIP_VITL14, seed=0))
    clip = clip.to("cuda")
    tok = torch.zeros(2, 77, dtype=torch.long, device="cuda")
    tok[:, 0] = 49406; tok[0, 1:9] = 1000; tok[0, 9] = 49407; tok[1, 1] = 49407
    ms = timed(lambda: clip.encode_text_with_sequence(tok), a.reps)
    out["clip_text_2x77"] = {"ms": round(ms, 3), "linear_weight_GB_per_s": round(12 * 12 * 768 * 768 * es / ms / 1e6, 1)}
    img = torch.randn(1, 3, 224, 224, device="cuda")
    ms = timed(lambda: clip.encode_image(img), a.reps)
    out["clip_image_1x224"] = {"ms": round(ms, 3), "linear_weight_GB_per_s": round(24 * 12 * 1024 * 1024 * es / ms / 1e6, 1)}
    del clip, te
    torch.cuda.empty_cache()
    # Kandinsky 2.2's image encoder: CLIP ViT-bigG/14 (kandinsky2_2_model.py:24), 1.85 B parameters, 16 heads of 104 channels
    c = k22.CLIP_BIGG_VISION
    big = k22.CLIPVisionModelWithProjectionHIP(c, backend_dtype=dt)
    big.load_state_dict(k22.init_clip_vision_hf_state_dict(c, seed=0))
    big = big.to("cuda")
    ms = timed(lambda: big(img).image_embeds, max(3, a.reps // 4))
    wbig = c["num_hidden_layers"] * (4 * c["hidden_size"] ** 2 + 2 * c["hidden_size"] * c["intermediate_size"]) * es
    out["clip_bigg_image_1x224"] = {"ms": round(ms, 3), "linear_weight_GB_per_s": round(wbig / ms / 1e6, 1)}
    out["dtype"] = a.dtype
    print(json.dumps(out))
