"""Kandinsky 2.2 driver on the HIP engines: `Kandinsky2_2` (kandinsky2/kandinsky2_2_model.py:15-173) with the decoder side -
the injected UNet, the DDPM loop of `KandinskyV22Pipeline` / `KandinskyV22ControlnetPipeline`, `movq.decode` - native.

PARITY UNPINNED (see unet22.py / oracle/unet22_ref.py): the 2.2 arithmetic is diffusers', absent here.

The 2.2 PRIOR (`KandinskyV22PriorPipeline`: CLIP-bigG text tower + diffusers' PriorTransformer + UnCLIPScheduler) is a
conditioning encoder in the sense of SURVEY 8f-3 - it runs once per prompt and its network is not in the reference tree - and is
reached through the conditioner:  conditioner.prior22(prompt, negative_prompt | None, batch_size, steps, guidance, device) ->
(image_embeds [bs,1280], negative_image_embeds [bs,1280]);  conditioner.encode_image22(image, device) -> [1,1280] (mix_images).
"""
from __future__ import annotations

import hashlib
import os
from typing import Optional

import torch

from . import _lib, prestep
from .movq import MoVQDecoderHIP, MoVQEncoderHIP
from .pipeline import prepare_image, process_images
from .unet22 import SCHEDULER_CONFIG_2_2, DDPMSchedulerHIP, UNet2DConditionHIP, make_arch22


class SeededPrior22:
    """Deterministic stand-in for KandinskyV22PriorPipeline (+ its CLIP image encoder): N(0,1) 1280-d embeddings seeded by the
    prompts.  negative_prompt=None gives the pipeline's zero-image embedding (get_zero_embed), the same for every prompt."""

    def __init__(self, dim=1280, seed=0):
        self.dim, self.seed = dim, seed

    def _emb(self, tag, text, batch_size, device):
        h = hashlib.sha256(f"{self.seed}|{tag}|{text}".encode()).digest()
        g = torch.Generator().manual_seed(int.from_bytes(h[:7], "little"))
        return torch.randn(1, self.dim, generator=g).repeat(batch_size, 1).to(device)

    def prior22(self, prompt, negative_prompt, batch_size, steps, guidance, device):
        """-> (image_embeds, negative_image_embeds) of prior(prompt=, negative_prompt=, num_images_per_prompt=batch_size, ...)"""
        neg = self._emb("zero", "", batch_size, device) * 0.1 if negative_prompt is None else self._emb("neg", negative_prompt, batch_size, device)
        return self._emb("pos", prompt, batch_size, device), neg

    def encode_image22(self, image, device):
        """image_encoder(preprocessed image).image_embeds of prior.interpolate: [1, dim]"""
        key = hashlib.sha256(image.detach().float().cpu().numpy().tobytes()).hexdigest() if torch.is_tensor(image) else repr(image)
        return self._emb("img", key, 1, device)


def _downscale(height, width, f=8):
    """downscale_height_and_width of the diffusers Kandinsky pipelines: latent size of a (height, width) image"""
    return (height // f ** 2 + (1 if height % f ** 2 else 0)) * f, (width // f ** 2 + (1 if width % f ** 2 else 0)) * f


class KandinskyV22DecoderHIP:
    """The decoder call of the reference (kandinsky2_2_model.py:77-80; ControlNet: notebooks/kandinsky2_2_controlnet.ipynb:9235):
        decoder(image_embeds=, negative_image_embeds=, num_inference_steps=, height=, width=, guidance_scale= [, hint=]).images
    One step = UNet on the CFG batch + k22_sampler_step (guidance, learned-range variance, clip +-2, ancestral noise), fused.
    The CFG batch is laid out [cond | uncond] inside (the engine's sampler step expects that order; batch elements are independent,
    so the order is an internal convention: diffusers uses [uncond | cond])."""

    def __init__(self, unet: UNet2DConditionHIP, movq: MoVQDecoderHIP, scheduler: Optional[DDPMSchedulerHIP] = None, movq_scale_factor: int = 8):
        self.unet, self.movq, self.scheduler = unet, movq, scheduler or DDPMSchedulerHIP.from_config(SCHEDULER_CONFIG_2_2)
        self.movq_scale_factor = movq_scale_factor

    def _check(self, image_embeds):
        if image_embeds.device.type != "cuda":
            raise RuntimeError(f"{type(self).__name__} runs on the GPU only (no CPU fallback)")
        if 2 * image_embeds.shape[0] > 8:
            raise ValueError("at most 4 images per call (CFG batch <= 8); shard larger batches over calls / ranks")

    def _denoise(self, x, emb, timesteps, guidance_scale, noise_seq, generator, hint2=None, extra=None, after_step=None):
        """x [bs,4,h,w]; emb [2bs,D] = [cond | uncond]; extra [2bs,5,h,w]: channels appended to every UNet input (inpainting);
        after_step(k, latents [bs,4,h,w]) -> latents"""
        bs = x.shape[0]
        x = torch.cat([x, x], 0).contiguous()
        self.unet.del_cache()
        ack = {"image_embeds": emb} if hint2 is None else {"image_embeds": emb, "hint": hint2}
        for k, t in enumerate(timesteps):
            half = x[:bs]
            inp = torch.cat([half, half], 0)
            if extra is not None:
                inp = torch.cat([inp, extra], 1)
            out = self.unet(inp, t, encoder_hidden_states=None, added_cond_kwargs=ack, return_dict=False)[0]
            nz = None
            if noise_seq is not None:
                nz = torch.cat([noise_seq[k], noise_seq[k]], 0).to(x.device)
            x = self.scheduler.step(out, t, x, noise=nz, generator=generator, guidance_scale=guidance_scale).prev_sample
            if after_step is not None:
                h = after_step(k, x[:bs])
                x = torch.cat([h, h], 0)
        self.unet.del_cache()
        self.last_latent = x[:bs]
        return self.last_latent

    def _images(self, latents, height, width, output_type):
        if output_type == "latent":
            return latents
        _, u8 = self.movq.decode(latents, return_uint8=True)                # movq.decode(latents, force_not_quantize=True)["sample"]
        return process_images(u8[:, :height, :width].contiguous(), output_type)

    @torch.no_grad()
    def __call__(self, image_embeds, negative_image_embeds, height=512, width=512, num_inference_steps=100, guidance_scale=4.0,
                 hint=None, latents=None, noise_seq=None, generator=None, output_type="pil"):
        self._check(image_embeds)
        dev, bs = image_embeds.device, image_embeds.shape[0]
        h, w = _downscale(height, width, self.movq_scale_factor)
        emb = torch.cat([image_embeds, negative_image_embeds], 0).float().contiguous()
        hint2 = None if hint is None else torch.cat([hint, hint], 0).float().contiguous()
        x = latents.float() if latents is not None else torch.randn(bs, 4, h, w, generator=generator, device=dev)
        self.scheduler.set_timesteps(num_inference_steps, device=dev)
        x = self._denoise(x, emb, self.scheduler.timesteps.tolist(), guidance_scale, noise_seq, generator, hint2=hint2)
        return self._images(x, height, width, output_type)


class KandinskyV22Img2ImgDecoderHIP(KandinskyV22DecoderHIP):
    """KandinskyV22Img2ImgPipeline as kandinsky2_2_model.py:105-109 calls it:
        decoder(image_embeds=, negative_image_embeds=, num_inference_steps=, height=, width=, guidance_scale=, strength=, image=).images
    movq.encode(image) -> add_noise at the first retained timestep -> the text2img loop over timesteps[t_start:]
    (get_timesteps: t_start = steps - min(int(steps * strength), steps)).  PARITY UNPINNED (diffusers, recalled)."""

    def __init__(self, unet, movq, movq_encoder: MoVQEncoderHIP, scheduler=None, movq_scale_factor: int = 8):
        super().__init__(unet, movq, scheduler, movq_scale_factor)
        self.movq_encoder = movq_encoder

    def get_timesteps(self, num_inference_steps, strength, device="cuda"):
        """KandinskyV22Img2ImgPipeline.get_timesteps: the last min(int(steps * strength), steps) timesteps of the schedule"""
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        t_start = max(num_inference_steps - min(int(num_inference_steps * strength), num_inference_steps), 0)
        ts = self.scheduler.timesteps.tolist()[t_start:]
        if not ts:
            raise ValueError("strength too small: no denoising step left")
        return ts

    def _latents_of(self, image, height, width, bs):
        """image: PIL, [n,3,H,W] in [-1,1], or latents [n,4,h,w] (passed through, as the diffusers pipeline does)"""
        if not torch.is_tensor(image):
            image = prepare_image(image, w=width, h=height)
        image = image.to("cuda").float()
        lat = image if image.shape[1] == 4 else self.movq_encoder.encode(image)
        if lat.shape[0] not in (1, bs):
            raise ValueError(f"{lat.shape[0]} init images for a batch of {bs}")
        return lat.repeat(bs // lat.shape[0], 1, 1, 1).contiguous()

    @torch.no_grad()
    def __call__(self, image_embeds, negative_image_embeds, image, height=512, width=512, num_inference_steps=100, guidance_scale=4.0,
                 strength=0.3, noise=None, noise_seq=None, generator=None, output_type="pil"):
        self._check(image_embeds)
        dev, bs = image_embeds.device, image_embeds.shape[0]
        emb = torch.cat([image_embeds, negative_image_embeds], 0).float().contiguous()
        lat0 = self._latents_of(image, height, width, bs)
        ts = self.get_timesteps(num_inference_steps, strength, dev)
        nz0 = noise.to(dev).float() if noise is not None else torch.randn(lat0.shape, generator=generator, device=dev)
        x = self.scheduler.add_noise(lat0, nz0, ts[0])
        x = self._denoise(x, emb, ts, guidance_scale, noise_seq, generator)
        return self._images(x, height, width, output_type)


class KandinskyV22InpaintDecoderHIP(KandinskyV22Img2ImgDecoderHIP):
    """KandinskyV22InpaintPipeline as kandinsky2_2_model.py:169-173 calls it (9-channel -decoder-inpaint UNet):
        decoder(image_embeds=, negative_image_embeds=, num_inference_steps=, height=, width=, guidance_scale=, image=, mask_image=).images
    UNet input = [latents | image_latents * mask | mask]; after every scheduler step the known region is re-imposed at the NEXT
    timestep's noise level, x = mask * add_noise(image_latents, initial_noise, t_next) + (1 - mask) * x, and un-noised after the last.
    mask_image: [H, W] / [1,1,H,W] array or tensor, 1 = keep, 0 = repaint - the convention of the reference's 2.1 `img_mask` and of the
    diffusers commit the reference's notebooks install (before diffusers 0.19 inverted it); pass repaint_white=True for the later
    convention.  The latent-resolution mask goes through the same 1-pixel erosion as 2.1 (prepare_mask).  PARITY UNPINNED."""

    @torch.no_grad()
    def __call__(self, image_embeds, negative_image_embeds, image, mask_image, height=512, width=512, num_inference_steps=100,
                 guidance_scale=4.0, latents=None, noise_seq=None, generator=None, repaint_white=False, output_type="pil"):
        import torch.nn.functional as F
        self._check(image_embeds)
        if self.unet.config.in_channels != 9:
            raise ValueError("the inpainting decoder needs the 9-channel UNet (make_arch22(inpainting=True))")
        dev, bs = image_embeds.device, image_embeds.shape[0]
        emb = torch.cat([image_embeds, negative_image_embeds], 0).float().contiguous()
        lat0 = self._latents_of(image, height, width, 1)                                   # [1,4,h,w]: one image per call, as the pipeline
        h, w = lat0.shape[-2:]
        m = torch.as_tensor(mask_image).float().to(dev)
        m = m.reshape((1, 1) + tuple(m.shape[-2:]))
        m = (m >= 0.5).float()
        if repaint_white:
            m = 1.0 - m
        m = prestep.prepare_mask(F.interpolate(m, (h, w), mode="nearest"))                  # [1,1,h,w], 1 = keep
        masked = (lat0 * m).contiguous()
        extra = torch.cat([masked, m], 1).repeat(2 * bs, 1, 1, 1).contiguous()
        x = latents.float() if latents is not None else torch.randn(bs, 4, h, w, generator=generator, device=dev)
        noise0 = x.clone()
        self.scheduler.set_timesteps(num_inference_steps, device=dev)
        ts = self.scheduler.timesteps.tolist()
        ac, L = self.scheduler.alphas_cumprod, _lib.lib()

        def reimpose(k, cur):
            out = torch.empty_like(cur)
            a = float(ac[ts[k + 1]]) if k + 1 < len(ts) else 1.0
            # image 0's latents / mask serve the whole batch; the noise is each sample's own initial noise
            for j in range(bs):
                _lib.check(L.k22_blend_noised(cur[j].contiguous().data_ptr(), lat0.data_ptr(), noise0[j].contiguous().data_ptr(), m.data_ptr(),
                                              a ** 0.5, (1.0 - a) ** 0.5, out[j].data_ptr(), 1, 4, h * w, 0, _lib.current_stream()))
            return out

        x = self._denoise(x, emb, ts, guidance_scale, noise_seq, generator, extra=extra, after_step=reimpose)
        return self._images(x, height, width, output_type)


def _load_weights(folder):
    """diffusion_pytorch_model.safetensors / .bin of one diffusers sub-folder"""
    import json
    st, bn = os.path.join(folder, "diffusion_pytorch_model.safetensors"), os.path.join(folder, "diffusion_pytorch_model.bin")
    if os.path.exists(st):
        from safetensors.torch import load_file
        sd = load_file(st)
    elif os.path.exists(bn):
        sd = torch.load(bn, map_location="cpu")
    else:
        raise FileNotFoundError(f"no diffusion_pytorch_model.safetensors / .bin under {folder}")
    cfg = os.path.join(folder, "config.json")
    return sd, (json.load(open(cfg)) if os.path.exists(cfg) else None)


def load_decoder22_from_cache_dir(cache_dir, task_type="text2img", controlnet=False):
    """What `from_pretrained('kandinsky-community/kandinsky-2-2-<repo>', subfolder='unet' | 'movq' | 'scheduler')` reads
    (kandinsky2_2_model.py:26-41), from a local copy `cache_dir/<repo>/{unet,movq,scheduler}/`: the UNet and MoVQ state dicts plus
    unet/config.json and scheduler/scheduler_config.json, which DRIVE the architecture and the scheduler (make_arch22,
    DDPMSchedulerHIP.from_config) - nothing about the checkpoint is assumed.  Raises when a folder is missing (no download path)."""
    import json
    repo = "kandinsky-2-2-controlnet-depth" if controlnet else ("kandinsky-2-2-decoder-inpaint" if task_type == "inpainting" else "kandinsky-2-2-decoder")
    root = os.path.join(cache_dir, repo)
    if not os.path.isdir(root):
        raise FileNotFoundError(f"{root} not found (expected the decoder repository's unet/, movq/, scheduler/ folders)")
    unet_sd, unet_cfg = _load_weights(os.path.join(root, "unet"))
    movq_sd, _ = _load_weights(os.path.join(root, "movq"))
    sc = os.path.join(root, "scheduler", "scheduler_config.json")
    return {"unet": unet_sd, "unet_config": unet_cfg, "movq": movq_sd, "scheduler_config": json.load(open(sc)) if os.path.exists(sc) else None}


class Kandinsky2_2HIP:
    """`Kandinsky2_2` (kandinsky2_2_model.py:15-173): get_new_h_w, generate_text2img, generate_img2img, mix_images,
    generate_inpainting with the reference's argument names and its (task_type -> pipeline, UNet) table."""

    def __init__(self, device="cuda", task_type="text2img", *, unet_state_dict=None, movq_state_dict=None, conditioner=None,
                 cache_dir=None, backend_dtype: torch.dtype = torch.bfloat16, use_graph: bool = True, unet_config=None, controlnet=None,
                 scheduler_config=None, movq_dtype: Optional[torch.dtype] = None):
        if task_type not in ("text2img", "img2img", "inpainting"):
            raise ValueError("Only text2img, img2img, inpainting is available")
        if (unet_state_dict is None or movq_state_dict is None) and cache_dir is not None:
            # the files from_pretrained('kandinsky-community/kandinsky-2-2-decoder[-inpaint]', subfolder=...) caches (kandinsky2_2_model.py:26-41)
            loaded = load_decoder22_from_cache_dir(cache_dir, task_type, bool(controlnet))
            unet_state_dict = unet_state_dict if unet_state_dict is not None else loaded["unet"]
            movq_state_dict = movq_state_dict if movq_state_dict is not None else loaded["movq"]
            unet_config = unet_config if unet_config is not None else loaded["unet_config"]
            scheduler_config = scheduler_config if scheduler_config is not None else loaded["scheduler_config"]
        if unet_state_dict is None or movq_state_dict is None:
            raise FileNotFoundError("Kandinsky 2.2 weights: pass unet_state_dict= (diffusers UNet2DConditionModel keys) and movq_state_dict=, or "
                                    "cache_dir= holding the decoder repository's unet/, movq/, scheduler/ folders (no download path in this build)")
        if controlnet and task_type != "text2img":
            raise ValueError("the ControlNet-depth UNet is a text2img decoder")
        # The reference builds the CLIP-bigG image encoder and the prior pipeline here (kandinsky2_2_model.py:24-31).  Seeded stand-in
        # embeddings would ignore the prompt, so they are an explicit opt-in (conditioner="seeded"), never a silent default.
        if conditioner is None:
            raise ValueError("Kandinsky2_2HIP: pass conditioner= (an object with prior22 / encode_image22: the 2.2 prior pipeline and its "
                             "CLIP-bigG image encoder), or conditioner='seeded' for the offline benchmark stand-in")
        if isinstance(conditioner, str):
            if conditioner != "seeded":
                raise ValueError("conditioner must be an object or the string 'seeded'")
            conditioner = SeededPrior22()
        self.device, self.task_type = device, task_type
        self.conditioner = conditioner
        arch = make_arch22(unet_config, controlnet=controlnet if controlnet else None, inpainting=True if task_type == "inpainting" else None)
        if arch.inpainting != (task_type == "inpainting"):
            raise ValueError(f"task_type {task_type!r} does not match the UNet config (in_channels {arch.in_channels})")
        scheduler = lambda: DDPMSchedulerHIP.from_config(scheduler_config if scheduler_config is not None else SCHEDULER_CONFIG_2_2)  # noqa: E731
        self.unet = UNet2DConditionHIP(arch, backend_dtype=backend_dtype, use_graph=use_graph)
        self.unet.load_state_dict(unet_state_dict)
        self.unet = self.unet.to(device).eval()
        # None: fp32 engines decode in fp32, 16-bit engines in fp16 (pipeline.py: within one / three grey levels of the fp32 decode)
        from .pipeline import aux_engine_dtypes
        mdt = aux_engine_dtypes(backend_dtype, movq_dtype)[1]     # one rule with the 2.1 driver: fp32 beside fp32 / split-precision engines
        movq = MoVQDecoderHIP(backend_dtype=mdt)
        movq.load_state_dict(movq_state_dict, strict=True)          # decoder keys; a full MOVQ checkpoint's other keys are skipped
        movq = movq.to(device)
        if task_type == "text2img":
            self.decoder = KandinskyV22DecoderHIP(self.unet, movq, scheduler())
        else:
            enc = MoVQEncoderHIP(backend_dtype=mdt)
            enc.load_state_dict(movq_state_dict, strict=True)
            cls = KandinskyV22Img2ImgDecoderHIP if task_type == "img2img" else KandinskyV22InpaintDecoderHIP
            self.decoder = cls(self.unet, movq, enc.to(device), scheduler())

    def get_new_h_w(self, h, w):
        return (h // 64 + (1 if h % 64 else 0)) * 64, (w // 64 + (1 if w % 64 else 0)) * 64

    def _negative(self, second_prompt, negative_decoder_prompt, batch_size, prior_steps, prior_guidance_scale):
        """the second prior call of every generate_* (kandinsky2_2_model.py:71-76, 99-104, 131-136, 161-166): no negative_prompt,
        so its .negative_image_embeds is the zero-image embedding; taken when negative_decoder_prompt == "", else .image_embeds"""
        pos, zero = self.conditioner.prior22(second_prompt, None, batch_size, prior_steps, prior_guidance_scale, self.device)
        return zero if negative_decoder_prompt == "" else pos

    @torch.no_grad()
    def generate_text2img(self, prompt, batch_size=1, decoder_steps=50, prior_steps=25, decoder_guidance_scale=4, prior_guidance_scale=4,
                          h=512, w=512, negative_prior_prompt="", negative_decoder_prompt="", *, hint=None, latents=None, noise_seq=None,
                          output_type="pil"):
        if self.task_type != "text2img":
            raise ValueError(f"this model was built for {self.task_type}")
        h, w = self.get_new_h_w(h, w)
        img_emb, _ = self.conditioner.prior22(prompt, negative_prior_prompt, batch_size, prior_steps, prior_guidance_scale, self.device)
        negative_emb = self._negative(negative_decoder_prompt, negative_decoder_prompt, batch_size, prior_steps, prior_guidance_scale)
        return self.decoder(image_embeds=img_emb, negative_image_embeds=negative_emb, num_inference_steps=decoder_steps, height=h, width=w,
                            guidance_scale=decoder_guidance_scale, hint=hint, latents=latents, noise_seq=noise_seq, output_type=output_type)

    @torch.no_grad()
    def generate_img2img(self, prompt, image, strength=0.4, batch_size=1, decoder_steps=100, prior_steps=25, decoder_guidance_scale=4,
                         prior_guidance_scale=4, h=512, w=512, negative_prior_prompt="", negative_decoder_prompt="", *, noise=None,
                         noise_seq=None, output_type="pil"):
        if self.task_type != "img2img":
            raise ValueError(f"this model was built for {self.task_type}")
        h, w = self.get_new_h_w(h, w)
        img_emb, _ = self.conditioner.prior22(prompt, negative_prior_prompt, batch_size, prior_steps, prior_guidance_scale, self.device)
        negative_emb = self._negative(negative_prior_prompt, negative_decoder_prompt, batch_size, prior_steps, prior_guidance_scale)
        return self.decoder(image_embeds=img_emb, negative_image_embeds=negative_emb, num_inference_steps=decoder_steps, height=h, width=w,
                            guidance_scale=decoder_guidance_scale, strength=strength, image=image, noise=noise, noise_seq=noise_seq,
                            output_type=output_type)

    @torch.no_grad()
    def mix_images(self, images_texts, weights, batch_size=1, decoder_steps=50, prior_steps=25, decoder_guidance_scale=4,
                   prior_guidance_scale=4, h=512, w=512, negative_prior_prompt="", negative_decoder_prompt="", *, latents=None,
                   noise_seq=None, output_type="pil"):
        """prior.interpolate (kandinsky2_2_model.py:127-130): weighted sum of the prior embedding of every text and the CLIP image
        embedding of every image."""
        assert len(images_texts) == len(weights) and len(images_texts) > 0
        if self.task_type != "text2img":
            raise ValueError(f"this model was built for {self.task_type}")
        h, w = self.get_new_h_w(h, w)
        img_emb = None
        for it, wt in zip(images_texts, weights):
            e = (self.conditioner.prior22(it, negative_prior_prompt, 1, prior_steps, prior_guidance_scale, self.device)[0] if isinstance(it, str)
                 else self.conditioner.encode_image22(it, self.device)) * wt
            img_emb = e if img_emb is None else img_emb + e
        img_emb = img_emb.repeat(batch_size, 1)
        negative_emb = self._negative(negative_prior_prompt, negative_decoder_prompt, batch_size, prior_steps, prior_guidance_scale)
        return self.decoder(image_embeds=img_emb, negative_image_embeds=negative_emb, num_inference_steps=decoder_steps, height=h, width=w,
                            guidance_scale=decoder_guidance_scale, latents=latents, noise_seq=noise_seq, output_type=output_type)

    @torch.no_grad()
    def generate_inpainting(self, prompt, pil_img, img_mask, batch_size=1, decoder_steps=50, prior_steps=25, decoder_guidance_scale=4,
                            prior_guidance_scale=4, h=512, w=512, negative_prior_prompt="", negative_decoder_prompt="", *, latents=None,
                            noise_seq=None, repaint_white=False, output_type="pil"):
        if self.task_type != "inpainting":
            raise ValueError(f"this model was built for {self.task_type}")
        img_emb, _ = self.conditioner.prior22(prompt, negative_prior_prompt, batch_size, prior_steps, prior_guidance_scale, self.device)
        negative_emb = self._negative(negative_prior_prompt, negative_decoder_prompt, batch_size, prior_steps, prior_guidance_scale)
        return self.decoder(image_embeds=img_emb, negative_image_embeds=negative_emb, num_inference_steps=decoder_steps, height=h, width=w,
                            guidance_scale=decoder_guidance_scale, image=pil_img, mask_image=img_mask, latents=latents, noise_seq=noise_seq,
                            repaint_white=repaint_white, output_type=output_type)
