"""img2img / inpainting pre-step on the device: the pieces of Kandinsky2_1.generate_img2img / generate_inpainting
that run once per call before the denoise loop (kandinsky2/kandinsky2_1_model.py:458-469, 519-534; kandinsky2/utils.py:11-54).

    latent = encoder.encode(image) * scale                      -> MoVQEncoderHIP (movq.py)
    image  = q_sample(latent, t, schedule_name, num_steps)      -> q_sample below (same fp64 tables as the reference)
    mask   = prepare_mask(F.interpolate(mask, latent_hw))       -> prepare_mask below (k22_prepare_mask; the reference is an
                                                                   O(h*w) Python loop over the latent grid)
No CPU fallback: the tensors must be on the GPU.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


def prepare_mask(mask: torch.Tensor) -> torch.Tensor:
    """mask [1, C, h, w] (the reference takes mask.float()[0]) -> [1, C, h, w] float32; kandinsky2/utils.py:11-31."""
    if mask.device.type != "cuda":
        raise RuntimeError("prepare_mask: the mask must be on the GPU (no CPU fallback)")
    m = mask.float()[0].contiguous()
    C, H, W = m.shape
    out = torch.empty_like(m)
    _lib.check(_lib.lib().k22_prepare_mask(m.data_ptr(), out.data_ptr(), C, H, W, _lib.current_stream()))
    return out.unsqueeze(0)


def q_sample(x_start: torch.Tensor, t, schedule_name: str = "linear", num_steps: int = 1000, noise: torch.Tensor = None,
             linear_start: float = 0.0001, linear_end: float = 0.02) -> torch.Tensor:
    """kandinsky2/utils.py:43-54: sqrt(alphas_cumprod[t]) * x_start + sqrt(1 - alphas_cumprod[t]) * noise with the fp64 numpy
    tables of kandinsky2/model/utils.py:get_named_beta_schedule — NOTE its "linear" schedule is the classic 1e-4 .. 2e-2 one
    (model/utils.py:32-40), not the 0.00085 .. 0.012 of the decoder's diffusion_config: the reference noises the init latent with
    that table and so does this function."""
    if schedule_name != "linear":
        raise NotImplementedError("only the 'linear' schedule (the 2.1 configuration)")
    if x_start.device.type != "cuda":
        raise RuntimeError("q_sample: tensors must be on the GPU (no CPU fallback)")
    scale = 1000 / num_steps
    betas = np.linspace(scale * linear_start, scale * linear_end, num_steps, dtype=np.float64)
    ac = np.cumprod(1.0 - betas, axis=0)
    if noise is None:
        noise = torch.randn_like(x_start)
    assert noise.shape == x_start.shape
    ti = torch.as_tensor(t).reshape(-1).long().cpu().numpy()
    a = torch.from_numpy(np.sqrt(ac)[ti]).to(device=x_start.device).float()
    b = torch.from_numpy(np.sqrt(1.0 - ac)[ti]).to(device=x_start.device).float()
    shape = (-1,) + (1,) * (x_start.dim() - 1)
    return a.reshape(shape) * x_start + b.reshape(shape) * noise


@torch.no_grad()
def img2img_init_latent(encoder, image: torch.Tensor, latent_scale: float, timestep_map, num_timesteps: int, strength: float,
                        noise: torch.Tensor = None, schedule_name: str = "linear", schedule_steps: int = 1000) -> torch.Tensor:
    """generate_img2img's pre-step (kandinsky2_1_model.py:458-469): encode, scale, noise to the step the loop starts from.
    timestep_map / num_timesteps are those of the (respaced) diffusion the loop will use."""
    latent = encoder.encode(image) * latent_scale
    start_step = int(num_timesteps * (1 - strength))
    t = torch.tensor(timestep_map[start_step - 1])
    return q_sample(latent, t, schedule_name=schedule_name, num_steps=schedule_steps, noise=noise)
