"""Kandinsky 2.2 driver on the HIP engines: `Kandinsky2_2` (kandinsky2/kandinsky2_2_model.py:15-173) with the decoder side -
the injected UNet, the DDPM loop of `KandinskyV22Pipeline` / `KandinskyV22ControlnetPipeline`, `movq.decode` - native.

PARITY UNPINNED (see unet22.py / oracle/unet22_ref.py): the 2.2 arithmetic is diffusers', absent here.

The 2.2 PRIOR (`KandinskyV22PriorPipeline`: CLIP-bigG text tower + diffusers' PriorTransformer + UnCLIPScheduler) is a
conditioning encoder in the sense of SURVEY 8f-3 - it runs once per prompt and its network is not in the reference tree - and is
reached through the conditioner:  conditioner.prior22(prompt, negative_prompt, batch_size, steps, guidance, device) ->
(image_embeds [bs,1280], negative_image_embeds [bs,1280]).
"""
from __future__ import annotations

import hashlib
from typing import Optional

import torch

from .movq import MoVQDecoderHIP
from .pipeline import process_images
from .unet22 import DDPMSchedulerHIP, UNet2DConditionHIP, make_arch22


class SeededPrior22:
    """Deterministic stand-in for KandinskyV22PriorPipeline: N(0,1) 1280-d embeddings seeded by the prompts."""

    def __init__(self, dim=1280, seed=0):
        self.dim, self.seed = dim, seed

    def prior22(self, prompt, negative_prompt, batch_size, steps, guidance, device):
        def emb(tag, text):
            h = hashlib.sha256(f"{self.seed}|{tag}|{text}".encode()).digest()
            g = torch.Generator().manual_seed(int.from_bytes(h[:7], "little"))
            return torch.randn(1, self.dim, generator=g).repeat(batch_size, 1).to(device)
        return emb("pos", prompt), emb("neg", negative_prompt)


class KandinskyV22DecoderHIP:
    """The decoder call of the reference (kandinsky2_2_model.py:77-80; ControlNet: notebooks/kandinsky2_2_controlnet.ipynb:9235):
        decoder(image_embeds=, negative_image_embeds=, num_inference_steps=, height=, width=, guidance_scale= [, hint=]).images
    One step = UNet on the CFG batch + k22_sampler_step (guidance, learned-range variance, clip +-2, ancestral noise), fused.
    The CFG batch is laid out [cond | uncond] inside (the engine's sampler step expects that order; batch elements are independent,
    so the order is an internal convention: diffusers uses [uncond | cond])."""

    def __init__(self, unet: UNet2DConditionHIP, movq: MoVQDecoderHIP, scheduler: Optional[DDPMSchedulerHIP] = None, movq_scale_factor: int = 8):
        self.unet, self.movq, self.scheduler = unet, movq, scheduler or DDPMSchedulerHIP()
        self.movq_scale_factor = movq_scale_factor

    @torch.no_grad()
    def __call__(self, image_embeds, negative_image_embeds, height=512, width=512, num_inference_steps=100, guidance_scale=4.0,
                 hint=None, latents=None, noise_seq=None, generator=None, output_type="pil"):
        dev = image_embeds.device
        if dev.type != "cuda":
            raise RuntimeError("KandinskyV22DecoderHIP runs on the GPU only (no CPU fallback)")
        bs = image_embeds.shape[0]
        if 2 * bs > 8:
            raise ValueError("at most 4 images per call (CFG batch <= 8); shard larger batches over calls / ranks")
        f = self.movq_scale_factor
        h, w = (height // f ** 2 + (1 if height % f ** 2 else 0)) * f, (width // f ** 2 + (1 if width % f ** 2 else 0)) * f   # downscale_height_and_width
        emb = torch.cat([image_embeds, negative_image_embeds], 0).float().contiguous()
        hint2 = None if hint is None else torch.cat([hint, hint], 0).float().contiguous()
        x = latents.float() if latents is not None else torch.randn(bs, 4, h, w, generator=generator, device=dev)
        x = torch.cat([x, x], 0).contiguous()
        self.scheduler.set_timesteps(num_inference_steps, device=dev)
        self.unet.del_cache()
        ack = {"image_embeds": emb} if hint2 is None else {"image_embeds": emb, "hint": hint2}
        for k, t in enumerate(self.scheduler.timesteps.tolist()):
            half = x[:bs]
            out = self.unet(torch.cat([half, half], 0), t, encoder_hidden_states=None, added_cond_kwargs=ack, return_dict=False)[0]
            nz = None
            if noise_seq is not None:
                nz = torch.cat([noise_seq[k], noise_seq[k]], 0).to(dev)
            x = self.scheduler.step(out, t, x, noise=nz, generator=generator, guidance_scale=guidance_scale).prev_sample
        self.unet.del_cache()
        self.last_latent = x[:bs]
        if output_type == "latent":
            return self.last_latent
        _, u8 = self.movq.decode(self.last_latent, return_uint8=True)       # movq.decode(latents, force_not_quantize=True)["sample"]
        return process_images(u8[:, :height, :width].contiguous(), output_type)


class Kandinsky2_2HIP:
    """`Kandinsky2_2` (kandinsky2_2_model.py:15-173): get_new_h_w, generate_text2img with the reference's argument names."""

    def __init__(self, device="cuda", task_type="text2img", *, unet_state_dict=None, movq_state_dict=None, conditioner=None,
                 cache_dir=None, backend_dtype: torch.dtype = torch.bfloat16, use_graph: bool = True, unet_config=None, controlnet=False):
        if task_type not in ("text2img",):
            raise ValueError("Only text2img is available on the HIP engines for 2.2 (img2img / inpainting pipelines: not built)")
        if unet_state_dict is None or movq_state_dict is None:
            raise FileNotFoundError("Kandinsky 2.2 weights: pass unet_state_dict= (diffusers UNet2DConditionModel keys) and movq_state_dict= "
                                    "(there is no download path in this build)")
        self.device, self.task_type = device, task_type
        self.conditioner = conditioner or SeededPrior22()
        self.unet = UNet2DConditionHIP(make_arch22(unet_config, controlnet=controlnet), backend_dtype=backend_dtype, use_graph=use_graph)
        self.unet.load_state_dict(unet_state_dict)
        self.unet = self.unet.to(device).eval()
        movq = MoVQDecoderHIP(backend_dtype=backend_dtype)
        movq.load_state_dict(movq_state_dict, strict=True)
        self.decoder = KandinskyV22DecoderHIP(self.unet, movq.to(device))

    def get_new_h_w(self, h, w):
        return (h // 64 + (1 if h % 64 else 0)) * 64, (w // 64 + (1 if w % 64 else 0)) * 64

    @torch.no_grad()
    def generate_text2img(self, prompt, batch_size=1, decoder_steps=50, prior_steps=25, decoder_guidance_scale=4, prior_guidance_scale=4,
                          h=512, w=512, negative_prior_prompt="", negative_decoder_prompt="", *, hint=None, latents=None, noise_seq=None,
                          output_type="pil"):
        h, w = self.get_new_h_w(h, w)
        img_emb, neg_of_prompt = self.conditioner.prior22(prompt, negative_prior_prompt, batch_size, prior_steps, prior_guidance_scale, self.device)
        if negative_decoder_prompt == "":
            negative_emb = neg_of_prompt                 # .negative_image_embeds of the prompt's prior call (kandinsky2_2_model.py:73-76)
        else:
            negative_emb, _ = self.conditioner.prior22(negative_decoder_prompt, "", batch_size, prior_steps, prior_guidance_scale, self.device)
        return self.decoder(image_embeds=img_emb, negative_image_embeds=negative_emb, num_inference_steps=decoder_steps, height=h, width=w,
                            guidance_scale=decoder_guidance_scale, hint=hint, latents=latents, noise_seq=noise_seq, output_type=output_type)
