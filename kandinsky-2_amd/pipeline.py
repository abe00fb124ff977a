"""Drop-in driver: the reference's `Kandinsky2_1` (kandinsky2/kandinsky2_1_model.py:20-548) and `get_kandinsky2`
(kandinsky2/__init__.py:164-192) with the three hot-path models - diffusion prior, latent UNet + sampler loop, MoVQ - on the
HIP engines.  Method names, argument names, defaults and the order of operations are the reference's:

    model = get_kandinsky2("cuda", task_type="text2img", cache_dir=..., model_version="2.1", conditioner=...)
    images = model.generate_text2img("red cat", num_steps=50, batch_size=1, guidance_scale=4, h=768, w=768,
                                     sampler="p_sampler", prior_cf_scale=4, prior_steps="5")

chains  prior (PriorDiffusionModelHIP) -> CFG batch -> p_sampler / DDIM / PLMS loop on Text2ImUNetHIP (fused sampler step)
-> MoVQDecoderHIP.decode(samples / scale) -> crop -> uint8 (process_images) entirely on the GPU.

The conditioning encoders (XLM-R text encoder, CLIP text and image towers; kandinsky2_1_model.py:117-181) are NOT part of
the hot path: they run once per prompt in PyTorch and are reached through a `conditioner` object:

    conditioner.encode_text(prompt, batch_size, device) -> full_emb [2bs,77,1024], pooled_emb [2bs,768]   (rows [prompt]*bs + [""]*bs)
    conditioner.clip_text(prompts, negative_prompt, device) -> txt_feat [2n,768], txt_feat_seq [2n,77,768], mask [2n,77] bool
    conditioner.encode_image(image, device) -> [1,768]           conditioner.zero_image_emb(device) -> [1,768]

`ReferenceConditioner` wraps the reference's own encoder modules (whoever has `clip`, the tokenizers and the checkpoints
builds them exactly as Kandinsky2_1.__init__ does); `SeededConditioner` produces deterministic N(0,1) embeddings of the right
shapes from a hash of the prompt (benchmarks and parity tests: no checkpoint or tokenizer is reachable offline).
There is no CPU fallback anywhere below.
"""
from __future__ import annotations

import copy
import hashlib
import os
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from .arch import DIFFUSION_CONFIG_2_1, MODEL_CONFIG_2_1, make_arch
from .diffusion import DDIMSamplerHIP, PLMSSamplerHIP, create_gaussian_diffusion
from .movq import MOVQ_CONFIG_2_1, MoVQDecoderHIP, MoVQEncoderHIP
from .prior import PRIOR_DIFFUSION_2_1, PRIOR_HPARAMS_2_1, PriorDiffusionModelHIP
from .unet import Text2ImUNetHIP
from . import prestep

# the reference's CONFIG_2_1 (kandinsky2/configs.py:64-158), restricted to the keys the sampling path reads
CONFIG_2_1 = {
    "clip_name": "ViT-L/14",
    "clip_image_size": 224,
    "tokenizer_name": "",
    "image_enc_params": {"name": "MOVQ", "scale": 1, "ckpt_path": "", "params": copy.deepcopy(MOVQ_CONFIG_2_1)},
    "text_enc_params": {"model_path": "", "model_name": "multiclip", "in_features": 1024, "out_features": 768},
    "prior": {"clip_mean_std_path": "ViT-L-14_stats.th",
              "params": {"model": {"type": "prior", "diffusion_sampler": "uniform", "hparams": copy.deepcopy(PRIOR_HPARAMS_2_1)},
                         "diffusion": copy.deepcopy(PRIOR_DIFFUSION_2_1)}},
    "model_config": copy.deepcopy(MODEL_CONFIG_2_1),
    "diffusion_config": copy.deepcopy(DIFFUSION_CONFIG_2_1),
}


def _load(obj):
    """a checkpoint path (torch.load, as the reference) or an already loaded state dict"""
    if isinstance(obj, (str, os.PathLike)):
        return torch.load(obj, map_location="cpu")
    return obj


def process_images(batch: torch.Tensor, output_type: str = "pil"):
    """kandinsky2/utils.py:57-70 for a float batch [B,3,H,W]; `batch` may already be the uint8 NHWC image the MoVQ engine's
    fused epilogue wrote (same arithmetic: ((x + 1) * 127.5).round().clamp(0, 255))."""
    if batch.dtype != torch.uint8:
        batch = ((batch + 1) * 127.5).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1)
    if output_type == "tensor":
        return batch
    arr = batch.to("cpu").numpy()
    if output_type == "uint8":
        return arr
    from PIL import Image
    return [Image.fromarray(arr[i]) for i in range(arr.shape[0])]


def prepare_image(image, w=512, h=512) -> torch.Tensor:
    """kandinsky2/utils.py:34-40 for a PIL image; a tensor [1,3,h,w] already in [-1,1] passes through."""
    if torch.is_tensor(image):
        if image.dim() != 4 or image.shape[1] != 3:
            raise ValueError("image tensor must be [1, 3, h, w] in [-1, 1]")
        return image.float()
    from PIL import Image
    pil = image.resize((w, h), resample=Image.BICUBIC, reducing_gap=1)
    arr = np.array(pil.convert("RGB")).astype(np.float32) / 127.5 - 1
    return torch.from_numpy(np.transpose(arr, [2, 0, 1])).unsqueeze(0)


class SeededConditioner:
    """Deterministic stand-in for the conditioning encoders: N(0,1) embeddings seeded by a hash of the prompt."""

    def __init__(self, text_dim1=1024, text_dim2=768, clip_dim=768, clip_xf_width=768, text_ctx=77, seed=0):
        self.d1, self.d2, self.cd, self.cw, self.ctx, self.seed = text_dim1, text_dim2, clip_dim, clip_xf_width, text_ctx, seed

    def _gen(self, tag: str, prompt: str) -> torch.Generator:
        h = hashlib.sha256(f"{self.seed}|{tag}|{prompt}".encode()).digest()
        return torch.Generator().manual_seed(int.from_bytes(h[:7], "little"))

    def encode_text(self, prompt: str, batch_size: int, device):
        rows_f, rows_p = [], []
        for p in [prompt] * batch_size + [""] * batch_size:
            g = self._gen("xlmr", p)
            rows_f.append(torch.randn(self.ctx, self.d1, generator=g))
            rows_p.append(torch.randn(self.d2, generator=g))
        return torch.stack(rows_f).to(device), torch.stack(rows_p).to(device)

    def clip_text(self, prompts: Sequence[str], negative_prompt: str, device):
        feats, seqs, masks = [], [], []
        for p in list(prompts) + [negative_prompt] * len(prompts):
            g = self._gen("clip", p)
            feats.append(torch.randn(self.cd, generator=g))
            seqs.append(torch.randn(self.ctx, self.cw, generator=g))
            n = min(self.ctx, 2 + len(p.split()))           # <start> words <end>, padded like the CLIP tokenizer
            m = torch.zeros(self.ctx, dtype=torch.bool)
            m[:n] = True
            masks.append(m)
        return torch.stack(feats).to(device), torch.stack(seqs).to(device), torch.stack(masks).to(device)

    def encode_image(self, image, device):
        key = hashlib.sha256(image.detach().float().cpu().numpy().tobytes()).hexdigest() if torch.is_tensor(image) else repr(image)
        return torch.randn(1, self.cd, generator=self._gen("clipimg", key)).to(device)

    def zero_image_emb(self, device):
        return torch.randn(1, self.cd, generator=self._gen("clipimg", "<zeros>")).to(device) * 0.1


class ReferenceConditioner:
    """The reference's encoder calls (kandinsky2_1_model.py:117-181, 294-297) on the reference's own PyTorch modules:
    text_encoder (kandinsky2.model.text_encoders.TextEncoder), tokenizer1 (XLM-R AutoTokenizer), tokenizer2
    (CustomizedTokenizer), clip_model / preprocess (clip.load(...)).  They run once per prompt, outside the hot path."""

    def __init__(self, text_encoder, tokenizer1, tokenizer2, clip_model, preprocess, text_ctx=77, clip_image_size=224):
        self.text_encoder, self.tokenizer1, self.tokenizer2 = text_encoder, tokenizer1, tokenizer2
        self.clip_model, self.preprocess, self.text_ctx, self.clip_image_size = clip_model, preprocess, text_ctx, clip_image_size

    @torch.no_grad()
    def encode_text(self, prompt, batch_size, device):
        enc = self.tokenizer1([prompt] * batch_size + [""] * batch_size, max_length=77, padding="max_length", truncation=True,
                              return_attention_mask=True, add_special_tokens=True, return_tensors="pt")
        full_emb, pooled_emb = self.text_encoder(tokens=enc["input_ids"].to(device), mask=enc["attention_mask"].to(device))
        return full_emb.float(), pooled_emb.float()

    @torch.no_grad()
    def clip_text(self, prompts, negative_prompt, device):
        tok, mask = self.tokenizer2.padded_tokens_and_mask(list(prompts), self.text_ctx)
        cf_token, cf_mask = self.tokenizer2.padded_tokens_and_mask([negative_prompt], self.text_ctx)
        if cf_token.shape != tok.shape:
            cf_token, cf_mask = cf_token.expand(tok.shape[0], -1), cf_mask.expand(tok.shape[0], -1)
        tok, mask = torch.cat([tok, cf_token], 0).to(device), torch.cat([mask, cf_mask], 0).to(device)
        cm = self.clip_model
        x = cm.token_embedding(tok).type(cm.dtype) + cm.positional_embedding.type(cm.dtype)
        x = cm.transformer(x.permute(1, 0, 2)).permute(1, 0, 2)
        x = cm.ln_final(x).type(cm.dtype)
        txt_feat = x[torch.arange(x.shape[0]), tok.argmax(dim=-1)] @ cm.text_projection
        return txt_feat.float(), x.float(), mask

    @torch.no_grad()
    def encode_image(self, image, device):
        if not torch.is_tensor(image):
            image = self.preprocess(image).unsqueeze(0)
        return self.clip_model.encode_image(image.to(device)).float()

    def zero_image_emb(self, device):
        return self.encode_image(torch.zeros(1, 3, self.clip_image_size, self.clip_image_size), device)


class Kandinsky2_1HIP:
    """`Kandinsky2_1` (kandinsky2/kandinsky2_1_model.py:20-548) on the HIP engines.  model_path / prior_path are checkpoint
    paths or state dicts; MoVQ weights come from config["image_enc_params"]["ckpt_path"] (path or state dict), clip_mean / clip_std
    from config["prior"]["clip_mean_std_path"] (path or a (mean, std) pair).  backend_dtype=torch.float32 selects the parity path."""

    def __init__(self, config, model_path, prior_path, device="cuda", task_type="text2img", *, conditioner=None,
                 backend_dtype: torch.dtype = torch.bfloat16, use_graph: bool = True, whole_loop_graph: Optional[bool] = None,
                 movq_dtype: Optional[torch.dtype] = None, chains: Optional[int] = None):
        if task_type not in ("text2img", "inpainting"):
            raise ValueError("Only text2img and inpainting is available")
        if torch.device(device).type != "cuda":
            raise RuntimeError("Kandinsky2_1HIP runs on the GPU only (no CPU fallback)")
        self.config = copy.deepcopy(config)
        self.device = device
        self.task_type = task_type
        self.backend_dtype = backend_dtype
        # p_sampler: the whole denoising loop as ONE hipGraph replay (k22_unet_sample_loop) - on whenever graphs are
        self.whole_loop_graph = use_graph if whole_loop_graph is None else bool(whole_loop_graph)
        self.use_fp16 = False                       # public tensors are fp32; engine precision is backend_dtype
        self.model_dtype = torch.float32
        self.clip_image_size = config.get("clip_image_size", 224)
        self.config["model_config"]["up"] = False
        self.config["model_config"]["inpainting"] = task_type == "inpainting"
        mcfg = self.config["model_config"]
        hp = self.config["prior"]["params"]["model"]["hparams"]
        # The reference builds its XLM-R and CLIP encoders here (kandinsky2_1_model.py:57-66).  A drop-in that silently
        # conditioned real checkpoints on hash-seeded noise would ignore the prompt, so the stand-in is an explicit opt-in:
        # conditioner="seeded" (benchmarks / parity tests), else a HIPConditioner / ReferenceConditioner object is required.
        if conditioner is None:
            raise ValueError("Kandinsky2_1HIP: pass conditioner=HIPConditioner(...) / ReferenceConditioner(...) (the XLM-R and CLIP "
                             "encoders of Kandinsky2_1.__init__), or conditioner='seeded' for the offline benchmark stand-in")
        if isinstance(conditioner, str):
            if conditioner != "seeded":
                raise ValueError("conditioner must be an object or the string 'seeded'")
            conditioner = SeededConditioner(mcfg.get("text_encoder_in_dim1", 1024), mcfg.get("text_encoder_in_dim2", 768), hp["clip_dim"],
                                            hp["clip_xf_width"], hp["text_ctx"])
        self.conditioner = conditioner

        ms = self.config["prior"]["clip_mean_std_path"]
        clip_mean, clip_std = _load(ms) if isinstance(ms, (str, os.PathLike)) else ms
        # the split-precision arithmetic ("f16x3") exists in the UNet engine only: beside it the prior (and MoVQ below) run their fp32 parity path
        aux_dtype, movq_auto = aux_engine_dtypes(backend_dtype, movq_dtype)
        self.prior = PriorDiffusionModelHIP(hp, self.config["prior"]["params"]["diffusion"], clip_mean.reshape(-1), clip_std.reshape(-1),
                                            backend_dtype=aux_dtype)
        self.prior.load_state_dict(_load(prior_path), strict=False)
        self.prior = self.prior.to(device)

        ie = self.config["image_enc_params"]
        if ie is None or ie.get("name") != "MOVQ":
            raise NotImplementedError("only the MOVQ image encoder of Kandinsky 2.1")
        self.use_image_enc, self.scale = True, ie["scale"]
        movq_sd = _load(ie["ckpt_path"])
        # MoVQ runs once per image and its output IS the picture.  movq_dtype=None (default) follows the engines' precision the way the reference
        # does (kandinsky2_1_model.py:92-94, 287-288: under use_fp16 the image encoder is .half() and decodes half latents): fp32 engines (the
        # parity path) decode in fp32 - uint8 image within ONE grey level of the reference's fp32 decode; 16-bit engines (the product path) decode
        # in fp16 - within 3 grey levels, 88 % of the bytes identical, at 12 ms instead of 53 (bf16 would move pixels by up to 24 levels, which
        # is why a bf16 UNet still gets an fp16 MoVQ; tests/test_movq_gpu.py, profiles/r03_movq_precision.txt).
        self.movq_dtype = movq_auto
        self.image_encoder = _MoVQ(ie["params"], movq_sd, self.movq_dtype, device)

        # chains: Text2ImUNetHIP(chains=...) - 2 = the CFG pair as two half-batch engines side by side (None: env K22_CHAINS, else 1)
        self.model = Text2ImUNetHIP(make_arch(mcfg, inpainting=mcfg["inpainting"]), backend_dtype=backend_dtype, use_graph=use_graph,
                                    cache_text_emb=True, chains=chains)
        self.model.load_state_dict(_load(model_path))
        self.model = self.model.to(device).eval()

    # ---- kandinsky2_1_model.py:105-112 --------------------------------------------------------------------------------
    def get_new_h_w(self, h, w):
        new_h = h // 64 + (1 if h % 64 else 0)
        new_w = w // 64 + (1 if w % 64 else 0)
        return new_h * 8, new_w * 8

    def encode_text(self, prompt, batch_size):
        return self.conditioner.encode_text(prompt, batch_size, self.device)

    # ---- kandinsky2_1_model.py:135-175 --------------------------------------------------------------------------------
    @torch.no_grad()
    def generate_clip_emb(self, prompt, batch_size=1, prior_cf_scale=4, prior_steps="25", negative_prior_prompt="",
                          noise: Optional[torch.Tensor] = None, noise_seq: Optional[torch.Tensor] = None):
        prompts_batch = [prompt for _ in range(batch_size)]
        scales = torch.tensor([prior_cf_scale] * batch_size, device=self.device, dtype=torch.float32)
        txt_feat, txt_feat_seq, mask = self.conditioner.clip_text(prompts_batch, negative_prior_prompt, self.device)
        return self.prior(txt_feat, txt_feat_seq, mask, scales, timestep_respacing=prior_steps, noise=noise, noise_seq=noise_seq).to(self.model_dtype)

    @torch.no_grad()
    def encode_images(self, image, is_pil=False):
        return self.conditioner.encode_image(image, self.device).to(self.model_dtype)

    @torch.no_grad()
    def create_zero_img_emb(self, batch_size):
        return self.conditioner.zero_image_emb(self.device).to(self.model_dtype).repeat(batch_size, 1)

    # ---- kandinsky2_1_model.py:183-292 --------------------------------------------------------------------------------
    @torch.no_grad()
    def generate_img(self, prompt, img_prompt, batch_size=1, diffusion=None, guidance_scale=7, init_step=None, noise=None,
                     init_img=None, img_mask=None, h=512, w=512, sampler="ddim_sampler", num_steps=50,
                     noise_seq: Optional[torch.Tensor] = None, output_type: str = "pil", *, text_embs=None, decode: bool = True):
        """text_embs = (full_emb, pooled_emb) computed by the caller (generate_text2img_many computes them beside the prior, on its stream);
        decode=False returns the final latents [batch_size, 4, h/8, w/8] and leaves the MoVQ decode to the caller."""
        new_h, new_w = self.get_new_h_w(h, w)
        full_batch_size = batch_size * 2
        if full_batch_size > 8:
            # The engine plans CFG batches of at most 8.  A larger batch cannot simply be split here: the reference's dynamic
            # threshold scales the WHOLE batch by the percentile of its element 0 (gaussian_diffusion.py:288-292), so chunks
            # would not reproduce one reference call.  Independent calls of <= 4 images (one per GPU stream / rank,
            # kandinsky2_amd.parallel.shard_range) are the supported way to a large batch.
            raise ValueError("batch_size must be <= 4 per call (CFG batch <= 8); shard larger batches over calls / ranks")
        model_kwargs = {}
        model_kwargs["full_emb"], model_kwargs["pooled_emb"] = self.encode_text(prompt, batch_size) if text_embs is None else text_embs
        model_kwargs["image_emb"] = img_prompt.to(self.device).float()
        self._last_image_emb = model_kwargs["image_emb"]
        if self.task_type == "inpainting":
            init_img = init_img.to(self.device).float()
            img_mask = img_mask.to(self.device).float()
            model_kwargs["inpaint_image"] = init_img * img_mask
            model_kwargs["inpaint_mask"] = img_mask
        else:
            init_img = img_mask = None              # the reference's text2img denoised_fun is the plain clamp (:241-243)
        if noise is not None:
            noise = noise.float()
        self.model.del_cache()
        if sampler == "p_sampler":
            samples = diffusion.p_sample_loop(
                self.model, (full_batch_size, 4, new_h, new_w), device=self.device, noise=noise, model_kwargs=model_kwargs,
                init_step=init_step, guidance_scale=guidance_scale, init_img=init_img, img_mask=img_mask, noise_seq=noise_seq,
                whole_loop_graph=self.whole_loop_graph)[:batch_size]
        elif sampler in ("ddim_sampler", "plms_sampler"):
            cls = DDIMSamplerHIP if sampler == "ddim_sampler" else PLMSSamplerHIP
            samples, _ = cls(self.model, diffusion, guidance_scale).sample(
                num_steps, batch_size * 2, (4, new_h, new_w), conditioning=model_kwargs, x_T=noise, init_step=init_step, device=self.device)
            samples = samples[:batch_size]
        else:
            raise ValueError("Only ddim_sampler and plms_sampler is available")
        self.model.del_cache()
        self.last_latent = samples
        if not decode:
            return samples
        return self._decode_images(samples, h, w, output_type)

    def _decode_images(self, samples, h, w, output_type):
        _, u8 = self.image_encoder.decode(samples / self.scale, return_uint8=True)
        return process_images(u8[:, :h, :w].contiguous(), output_type)

    # ---- a batch of prompts as a three-stage pipeline (BASELINE north star: "batch-of-prompts generation") -----------------------------
    @torch.no_grad()
    def generate_text2img_many(self, prompts, num_steps=100, batch_size=1, guidance_scale=7, h=512, w=512, sampler="ddim_sampler",
                               prior_cf_scale=4, prior_steps="25", negative_prior_prompt="", negative_decoder_prompt="", *,
                               noises=None, noise_seqs=None, prior_noises=None, prior_noise_seqs=None, output_type="pil", prior_group: int = 1):
        """generate_text2img for a LIST of prompts -> list of results, each equal to what generate_text2img(prompt, ...) returns for the same
        noise.  The three engines of a generation (conditioning + prior | denoise loop | MoVQ decode) are independent between prompts, so they
        run as a pipeline on three streams of the device: while the UNet denoises prompt i, the prior (a weight stream that leaves the matrix
        cores idle) already samples the embedding of prompt i + 1 and the decoder finishes prompt i - 1.  The reference generates one prompt
        after the other (kandinsky2_1_model.py:299-351); images per second of a prompt batch are bounded by the slowest stage instead of the
        sum.  noises / noise_seqs / prior_noises / prior_noise_seqs: optional per-prompt lists (parity tests).
        prior_group > 1: the diffusion prior samples the embeddings of up to prior_group prompts in ONE call (its batch = 2 * batch_size *
        group <= 8): the prior is a weight stream - 2 GB per forward whatever the batch - so a group of four costs about what one prompt
        costs.  Every row of the prior is computed independently of the other rows, but a larger batch runs other GEMM tile shapes (other
        fp32 summation orders): results then equal the per-prompt calls to rounding, not bit for bit (the default, 1, does)."""
        prompts = list(prompts)
        n = len(prompts)
        pick = lambda lst, i: None if lst is None else lst[i]          # noqa: E731
        cur = torch.cuda.current_stream()
        if getattr(self, "_pipe_streams", None) is None:
            self._pipe_streams = (torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device))
        s_prior, s_dec = self._pipe_streams
        s_prior.wait_stream(cur)
        s_dec.wait_stream(cur)
        _, diffusion = self._diffusion(sampler, num_steps)

        group = max(1, min(int(prior_group), 4 // max(1, batch_size)))
        if negative_decoder_prompt != "":
            group = 1                     # (the negative decoder prompt takes a prior call of its own: kept per prompt)
        grouped = {}                      # prompt index -> image embedding rows computed by its group's prior call

        def prior_for_group(i0):
            """one prior call for prompts i0 .. i0 + group - 1: rows [cond of every prompt | uncond of every prompt]"""
            idx = list(range(i0, min(i0 + group, n)))
            bs = batch_size
            plist = [prompts[i] for i in idx for _ in range(bs)]
            scales = torch.tensor([prior_cf_scale] * len(plist), device=self.device, dtype=torch.float32)
            txt_feat, txt_feat_seq, mask = self.conditioner.clip_text(plist, negative_prior_prompt, self.device)
            nz = nzs = None
            if prior_noises is not None:   # per-prompt [2 bs, D] -> [cond rows of all | uncond rows of all]
                nz = torch.cat([prior_noises[i][:bs] for i in idx] + [prior_noises[i][bs:] for i in idx], 0)
            if prior_noise_seqs is not None:
                nzs = torch.cat([prior_noise_seqs[i][:, :bs] for i in idx] + [prior_noise_seqs[i][:, bs:] for i in idx], 1)
            emb = self.prior(txt_feat, txt_feat_seq, mask, scales, timestep_respacing=prior_steps, noise=nz, noise_seq=nzs).to(self.model_dtype)
            for k, i in enumerate(idx):
                grouped[i] = emb[k * bs:(k + 1) * bs]

        def stage_a(i):
            with torch.cuda.stream(s_prior):
                if group > 1:
                    if i not in grouped:
                        prior_for_group(i)
                    emb = torch.cat([grouped.pop(i), self.create_zero_img_emb(batch_size=batch_size)], dim=0).to(self.device)
                else:
                    emb = self._image_embs(prompts[i], batch_size, prior_cf_scale, prior_steps, negative_prior_prompt, negative_decoder_prompt,
                                           noise=pick(prior_noises, i), noise_seq=pick(prior_noise_seqs, i))
                txt = self.encode_text(prompts[i], batch_size)
                ev = torch.cuda.Event()
                ev.record(s_prior)
            return emb, txt, ev

        results, pending = [], stage_a(0) if n else None
        for i in range(n):
            emb, txt, ev = pending
            cur.wait_event(ev)
            for t in (emb,) + tuple(txt):
                if torch.is_tensor(t):
                    t.record_stream(cur)
            pending = stage_a(i + 1) if i + 1 < n else None          # enqueued BEFORE this prompt's denoise loop: it overlaps it
            lat = self.generate_img(prompt=prompts[i], img_prompt=emb, batch_size=batch_size, guidance_scale=guidance_scale, h=h, w=w,
                                    sampler=sampler, num_steps=num_steps, diffusion=diffusion, noise=pick(noises, i), noise_seq=pick(noise_seqs, i),
                                    output_type=output_type, text_embs=txt, decode=False)
            ev_u = torch.cuda.Event()
            ev_u.record(cur)
            with torch.cuda.stream(s_dec):
                s_dec.wait_event(ev_u)
                lat.record_stream(s_dec)
                results.append(self._decode_images(lat, h, w, "tensor"))
        cur.wait_stream(s_dec)
        cur.wait_stream(s_prior)
        for r in results:
            r.record_stream(cur)
        return [process_images(r, output_type) if output_type != "tensor" else r for r in results]

    def _image_embs(self, prompt, batch_size, prior_cf_scale, prior_steps, negative_prior_prompt, negative_decoder_prompt, **prior_noise):
        image_emb = self.generate_clip_emb(prompt, batch_size=batch_size, prior_cf_scale=prior_cf_scale, prior_steps=prior_steps,
                                           negative_prior_prompt=negative_prior_prompt, **prior_noise)
        if negative_decoder_prompt == "":
            zero_image_emb = self.create_zero_img_emb(batch_size=batch_size)
        else:
            zero_image_emb = self.generate_clip_emb(negative_decoder_prompt, batch_size=batch_size, prior_cf_scale=prior_cf_scale,
                                                    prior_steps=prior_steps, negative_prior_prompt=negative_prior_prompt)
        return torch.cat([image_emb, zero_image_emb], dim=0).to(self.device)

    def _diffusion(self, sampler, num_steps):
        cfg = copy.deepcopy(self.config["diffusion_config"])
        if sampler == "p_sampler":
            cfg["timestep_respacing"] = str(num_steps)
        return cfg, create_gaussian_diffusion(**cfg)

    # ---- kandinsky2_1_model.py:299-351 --------------------------------------------------------------------------------
    @torch.no_grad()
    def generate_text2img(self, prompt, num_steps=100, batch_size=1, guidance_scale=7, h=512, w=512, sampler="ddim_sampler",
                          prior_cf_scale=4, prior_steps="25", negative_prior_prompt="", negative_decoder_prompt="", *,
                          noise=None, noise_seq=None, prior_noise=None, prior_noise_seq=None, output_type="pil"):
        image_emb = self._image_embs(prompt, batch_size, prior_cf_scale, prior_steps, negative_prior_prompt, negative_decoder_prompt,
                                     noise=prior_noise, noise_seq=prior_noise_seq)
        _, diffusion = self._diffusion(sampler, num_steps)
        return self.generate_img(prompt=prompt, img_prompt=image_emb, batch_size=batch_size, guidance_scale=guidance_scale, h=h, w=w,
                                 sampler=sampler, num_steps=num_steps, diffusion=diffusion, noise=noise, noise_seq=noise_seq,
                                 output_type=output_type)

    # ---- kandinsky2_1_model.py:353-426 --------------------------------------------------------------------------------
    @torch.no_grad()
    def mix_images(self, images_texts, weights, num_steps=100, batch_size=1, guidance_scale=7, h=512, w=512, sampler="ddim_sampler",
                   prior_cf_scale=4, prior_steps="25", negative_prior_prompt="", negative_decoder_prompt="", *, output_type="pil"):
        assert len(images_texts) == len(weights) and len(images_texts) > 0
        image_emb = None
        for it, wt in zip(images_texts, weights):
            e = (self.generate_clip_emb(it, batch_size=1, prior_cf_scale=prior_cf_scale, prior_steps=prior_steps,
                                        negative_prior_prompt=negative_prior_prompt) if isinstance(it, str)
                 else self.encode_images(it, is_pil=True)) * wt
            image_emb = e if image_emb is None else image_emb + e
        image_emb = image_emb.repeat(batch_size, 1)
        if negative_decoder_prompt == "":
            zero_image_emb = self.create_zero_img_emb(batch_size=batch_size)
        else:
            zero_image_emb = self.generate_clip_emb(negative_decoder_prompt, batch_size=batch_size, prior_cf_scale=prior_cf_scale,
                                                    prior_steps=prior_steps, negative_prior_prompt=negative_prior_prompt)
        image_emb = torch.cat([image_emb, zero_image_emb], dim=0).to(self.device)
        _, diffusion = self._diffusion(sampler, num_steps)
        return self.generate_img(prompt="", img_prompt=image_emb, batch_size=batch_size, guidance_scale=guidance_scale, h=h, w=w,
                                 sampler=sampler, num_steps=num_steps, diffusion=diffusion, output_type=output_type)

    # ---- kandinsky2_1_model.py:428-483 --------------------------------------------------------------------------------
    @torch.no_grad()
    def generate_img2img(self, prompt, pil_img, strength=0.7, num_steps=100, batch_size=1, guidance_scale=7, h=512, w=512,
                         sampler="ddim_sampler", prior_cf_scale=4, prior_steps="25", *, q_noise=None, noise_seq=None, output_type="pil"):
        image_emb = self._image_embs(prompt, batch_size, prior_cf_scale, prior_steps, "", "")
        cfg, diffusion = self._diffusion(sampler, num_steps)
        image = prepare_image(pil_img, h=h, w=w).to(self.device)
        image = prestep.img2img_init_latent(self.image_encoder.encoder, image, self.scale, diffusion.timestep_map, diffusion.num_timesteps,
                                            strength, noise=q_noise, schedule_name=cfg["noise_schedule"], schedule_steps=cfg["steps"])
        start_step = int(diffusion.num_timesteps * (1 - strength))
        image = image.repeat(2, 1, 1, 1)
        return self.generate_img(prompt=prompt, img_prompt=image_emb, batch_size=batch_size, guidance_scale=guidance_scale, h=h, w=w,
                                 sampler=sampler, num_steps=num_steps, diffusion=diffusion, noise=image, init_step=start_step,
                                 noise_seq=noise_seq, output_type=output_type)

    # ---- kandinsky2_1_model.py:485-548 --------------------------------------------------------------------------------
    @torch.no_grad()
    def generate_inpainting(self, prompt, pil_img, img_mask, num_steps=100, batch_size=1, guidance_scale=7, h=512, w=512,
                            sampler="ddim_sampler", prior_cf_scale=4, prior_steps="25", negative_prior_prompt="", negative_decoder_prompt="",
                            *, noise=None, noise_seq=None, output_type="pil"):
        image_emb = self._image_embs(prompt, batch_size, prior_cf_scale, prior_steps, negative_prior_prompt, "")
        _, diffusion = self._diffusion(sampler, num_steps)
        image = prepare_image(pil_img, w, h).to(self.device)
        image = self.image_encoder.encode(image) * self.scale
        image_shape = tuple(image.shape[-2:])
        m = img_mask if torch.is_tensor(img_mask) else torch.from_numpy(np.asarray(img_mask))
        m = torch.nn.functional.interpolate(m.float().unsqueeze(0).unsqueeze(0), image_shape, mode="nearest")
        m = prestep.prepare_mask(m.to(self.device))
        image = image.repeat(2 * batch_size, 1, 1, 1) if batch_size > 1 else image.repeat(2, 1, 1, 1)
        m = m.repeat(image.shape[0], 1, 1, 1)
        return self.generate_img(prompt=prompt, img_prompt=image_emb, batch_size=batch_size, guidance_scale=guidance_scale, h=h, w=w,
                                 sampler=sampler, num_steps=num_steps, diffusion=diffusion, init_img=image, img_mask=m, noise=noise,
                                 noise_seq=noise_seq, output_type=output_type)


class _MoVQ:
    """self.image_encoder of the reference (`MOVQ`, kandinsky2/vqgan/autoencoder.py:163-185): .decode / .encode on the HIP engines;
    the encoder engine is built on first use (text2img never needs it)."""

    def __init__(self, params, state_dict, backend_dtype, device):
        self._params, self._sd, self._dt, self._dev = params, state_dict, backend_dtype, device
        self.decoder = MoVQDecoderHIP(params["ddconfig"], params.get("n_embed", 16384), params.get("embed_dim", 4), backend_dtype=backend_dtype)
        self.decoder.load_state_dict(state_dict, strict=True)
        self.decoder = self.decoder.to(device)
        self._encoder = None

    @property
    def encoder(self):
        if self._encoder is None:
            e = MoVQEncoderHIP(self._params["ddconfig"], self._params.get("n_embed", 16384), self._params.get("embed_dim", 4), backend_dtype=self._dt)
            e.load_state_dict(self._sd, strict=True)
            self._encoder = e.to(self._dev)
        return self._encoder

    def decode(self, quant, return_uint8=False):
        return self.decoder.decode(quant, return_uint8=return_uint8)

    def encode(self, x):
        return self.encoder.encode(x)

    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self


def aux_engine_dtypes(backend_dtype, movq_dtype=None):
    """(prior / towers dtype, MoVQ dtype) that go with a UNet engine of `backend_dtype` - ONE rule for the 2.1 and the 2.2 drivers
    (pipeline22 had its own copy, which sent an fp16 MoVQ beside the "f16x3" engine).  The split-precision arithmetics are
    strings and exist in the UNet engine only: beside "f16x3" everything runs its fp32 parity path, beside "f16x2" fp16 (below).  MoVQ: movq_dtype=None follows the
    engines the way the reference does under use_fp16 (kandinsky2_1_model.py:92-94, 287-288): fp32 beside fp32-class engines, fp16
    beside 16-bit engines (also beside bf16: a bf16 decode moves pixels by up to 24 grey levels)."""
    # round 6: beside the ASYMMETRIC split ("f16x2": weights at ~22 bits, activation operands at fp16) the prior, the towers and the MoVQ decode run
    # in fp16, the reference's own use_fp16 practice - its activations are fp16-class anyway, and fp32 aux engines cost more than its whole
    # denoise loop (e2e 1004 -> 67x ms per image; image-level distance measured: tests/test_pipeline_gpu.py::test_generate_text2img_f16x2_*).
    # "f16x3" (the 1e-6-class engine) keeps everything beside it in fp32.
    if backend_dtype == _lib.F16X2:
        aux = torch.float16
    else:
        aux = torch.float32 if isinstance(backend_dtype, str) else backend_dtype
    movq = (torch.float32 if aux == torch.float32 else torch.float16) if movq_dtype is None else movq_dtype
    return aux, movq


def _conditioner_from_cache_dir(cache_dir, device, backend_dtype, tokenizer2=None):
    """The encoders Kandinsky2_1.__init__ builds (kandinsky2_1_model.py:57-66; files as kandinsky2/__init__.py:124-160 stores them):
    cache_dir/text_encoder (XLM-R tokenizer + pytorch_model.bin of MultilingualCLIP) and cache_dir/ViT-L-14.pt, on the HIP encoder
    engine.  tokenizer2 = the CLIP byte-pair tokenizer (what the reference builds as kandinsky2.model.prior.CustomizedTokenizer()):
    None = the package's own ClipBPETokenizer on the merges file found in cache_dir; an object with padded_tokens_and_mask(...) overrides
    it.  The product path imports nothing from the reference package.  Raises - never substitutes seeded noise - when a file is missing."""
    backend_dtype = aux_engine_dtypes(backend_dtype)[0]   # the split-precision arithmetics exist in the UNet engine only
    from .encoders import CLIPModelHIP, HIPConditioner, TextEncoderHIP
    te_dir, clip_pt = os.path.join(cache_dir, "text_encoder"), os.path.join(cache_dir, "ViT-L-14.pt")
    missing = [p for p in (os.path.join(te_dir, "pytorch_model.bin"), clip_pt) if not os.path.exists(p)]
    if missing:
        raise FileNotFoundError(f"Kandinsky 2.1 conditioning encoders not found: {missing}; pass conditioner=... explicitly "
                                "(conditioner='seeded' is the offline benchmark stand-in)")
    from transformers import AutoTokenizer
    tokenizer1 = AutoTokenizer.from_pretrained(te_dir)
    if tokenizer2 is None:
        # the package's own CLIP byte-pair tokenizer (tokenizer.py: the published algorithm of OpenAI clip's SimpleTokenizer + the
        # reference's padded_tokens_and_mask) on the merges file the `clip` package ships; tokenizer2= stays an override
        from .tokenizer import BPE_FILE_NAMES, ClipBPETokenizer, find_bpe_file
        bpe = find_bpe_file(cache_dir)
        if bpe is None:
            raise FileNotFoundError(f"CLIP byte-pair merges not found: put one of {BPE_FILE_NAMES} (it ships inside OpenAI's `clip` package) into "
                                    f"{cache_dir}, or pass tokenizer2= (any object with padded_tokens_and_mask(texts, text_ctx)) / a ready conditioner=")
        tokenizer2 = ClipBPETokenizer(bpe)
    clip_sd = torch.jit.load(clip_pt, map_location="cpu").state_dict()   # the OpenAI checkpoint is a TorchScript archive (clip.load)
    clip_model = CLIPModelHIP(backend_dtype=backend_dtype)
    clip_model.load_state_dict({k: v.float() for k, v in clip_sd.items() if k in clip_model.state_dict()}, strict=True)
    text_encoder = TextEncoderHIP(te_dir, "multiclip", state_dict=torch.load(os.path.join(te_dir, "pytorch_model.bin"), map_location="cpu"),
                                  backend_dtype=backend_dtype)
    return HIPConditioner(text_encoder.to(device), tokenizer1, tokenizer2, clip_model.to(device).eval())


def get_kandinsky2(device, task_type="text2img", cache_dir="/tmp/kandinsky2", use_auth_token=None, model_version="2.1",
                   use_flash_attention=False, *, conditioner=None, backend_dtype: torch.dtype = torch.bfloat16, use_graph: bool = True,
                   movq_dtype: Optional[torch.dtype] = None, tokenizer2=None):
    """`get_kandinsky2` (kandinsky2/__init__.py:164-192) for the HIP engines.  The reference downloads the checkpoints into
    cache_dir (kandinsky2/__init__.py:100-160); this box-local variant reads the same file names from cache_dir and raises if
    they are not there (there is no download path).  use_flash_attention is accepted and ignored: attention always runs in the
    engine's own fused kernel."""
    if model_version == "2.1":
        config = copy.deepcopy(CONFIG_2_1)
        model_name = {"text2img": "decoder_fp16.ckpt", "inpainting": "inpainting_fp16.ckpt"}.get(task_type)
        if model_name is None:
            raise ValueError("Only text2img and inpainting is available")
        need = {n: os.path.join(cache_dir, n) for n in (model_name, "prior_fp16.ckpt", "movq_final.ckpt", "ViT-L-14_stats.th")}
        missing = [p for p in need.values() if not os.path.exists(p)]
        if missing:
            raise FileNotFoundError(f"Kandinsky 2.1 checkpoints not found (no download path in this build): {missing}")
        config["prior"]["clip_mean_std_path"] = need["ViT-L-14_stats.th"]
        config["image_enc_params"]["ckpt_path"] = need["movq_final.ckpt"]
        if conditioner is None:
            conditioner = _conditioner_from_cache_dir(cache_dir, device, backend_dtype, tokenizer2)
        return Kandinsky2_1HIP(config, need[model_name], need["prior_fp16.ckpt"], device, task_type=task_type, conditioner=conditioner,
                               backend_dtype=backend_dtype, use_graph=use_graph, movq_dtype=movq_dtype)
    if model_version == "2.2":
        from .pipeline22 import Kandinsky2_2HIP
        return Kandinsky2_2HIP(device=device, task_type=task_type, cache_dir=cache_dir, conditioner=conditioner, backend_dtype=backend_dtype,
                               use_graph=use_graph, movq_dtype=movq_dtype)
    raise ValueError("Only 2.1 and 2.2 are available on the HIP engines")
