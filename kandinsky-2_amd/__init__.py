"""kandinsky-2_amd — MI355X-native Kandinsky-2 sampling engine (import as `kandinsky2_amd`).

Hot path only: the classifier-free-guided latent UNet denoise loop of Kandinsky 2.1/2.2
as hand-written gfx950 HIP kernels behind the C ABI in include/k22.h, plus the Python mirror of the
reference's module / sampler interface for that path.
"""
from .arch import MODEL_CONFIG_2_1, DIFFUSION_CONFIG_2_1, UNetArch, make_arch, param_shapes, tiny_model_config
from .unet import Text2ImUNetHIP, create_model
from .diffusion import SpacedDiffusionHIP, DDIMSamplerHIP, PLMSSamplerHIP, create_gaussian_diffusion, space_timesteps, percentile_index
from .weights import init_unet_state_dict, make_conditioning
from .prior import (PRIOR_HPARAMS_2_1, PRIOR_DIFFUSION_2_1, PriorDiffusionModelHIP, PriorSchedule, prior_param_shapes,
                    init_prior_state_dict, tiny_prior_hparams)
from . import prestep
from ._lib import F16X2, F16X3
from .tokenizer import ClipBPETokenizer
from . import pipeline22
from .unet22 import (DDPM_SCHEDULER_DEFAULTS, SCHEDULER_CONFIG_2_2, SCHEDULER_CONFIG_2_2_LEARNED_RANGE, UNET2D_DEFAULTS, UNET_CONFIG_2_2,
                     DDPMSchedulerHIP, UNet2DConditionHIP, init_unet22_state_dict, make_arch22, param_shapes22, resolve_unet22_config,
                     tiny_unet22_config)
from .pipeline import (CONFIG_2_1, Kandinsky2_1HIP, ReferenceConditioner, SeededConditioner, get_kandinsky2, prepare_image,
                       process_images)
from .encoders import (CLIP_BIGG_VISION, CLIP_VITL14, XLMR_LARGE, CLIPModelHIP, CLIPVisionModelWithProjectionHIP, HIPConditioner,
                       MultilingualCLIPHIP, TextEncoderHIP, clip_param_shapes, clip_vision_hf_param_shapes, init_clip_state_dict,
                       init_clip_vision_hf_state_dict, init_multiclip_state_dict, multiclip_param_shapes, tiny_clip_config,
                       tiny_clip_vision_hf_config, tiny_xlmr_config)
from .movq import (MOVQ_CONFIG_2_1, MoVQArch, MoVQDecoderHIP, MoVQEncoderHIP, movq_param_shapes, init_movq_state_dict,
                   movq_encoder_param_shapes, init_movq_encoder_state_dict)

__all__ = [
    "F16X3",
    "F16X2",
    "ClipBPETokenizer",
    "MODEL_CONFIG_2_1", "DIFFUSION_CONFIG_2_1", "UNetArch", "make_arch", "param_shapes", "tiny_model_config",
    "Text2ImUNetHIP", "create_model", "SpacedDiffusionHIP", "DDIMSamplerHIP", "PLMSSamplerHIP", "create_gaussian_diffusion", "space_timesteps",
    "percentile_index", "init_unet_state_dict", "make_conditioning",
    "PRIOR_HPARAMS_2_1", "PRIOR_DIFFUSION_2_1", "PriorDiffusionModelHIP", "PriorSchedule", "prior_param_shapes",
    "init_prior_state_dict", "tiny_prior_hparams",
    "MOVQ_CONFIG_2_1", "MoVQArch", "MoVQDecoderHIP", "MoVQEncoderHIP", "movq_param_shapes", "init_movq_state_dict",
    "movq_encoder_param_shapes", "init_movq_encoder_state_dict", "prestep",
    "UNET_CONFIG_2_2", "UNET2D_DEFAULTS", "DDPM_SCHEDULER_DEFAULTS", "SCHEDULER_CONFIG_2_2", "SCHEDULER_CONFIG_2_2_LEARNED_RANGE",
    "resolve_unet22_config", "DDPMSchedulerHIP", "UNet2DConditionHIP", "init_unet22_state_dict", "make_arch22", "param_shapes22", "tiny_unet22_config",
    "CLIP_VITL14", "CLIP_BIGG_VISION", "CLIPVisionModelWithProjectionHIP", "clip_vision_hf_param_shapes", "init_clip_vision_hf_state_dict",
    "tiny_clip_vision_hf_config", "XLMR_LARGE", "CLIPModelHIP", "HIPConditioner", "MultilingualCLIPHIP", "TextEncoderHIP", "clip_param_shapes",
    "init_clip_state_dict", "init_multiclip_state_dict", "multiclip_param_shapes", "tiny_clip_config", "tiny_xlmr_config",
    "CONFIG_2_1", "Kandinsky2_1HIP", "ReferenceConditioner", "SeededConditioner", "get_kandinsky2", "prepare_image", "process_images",
]
