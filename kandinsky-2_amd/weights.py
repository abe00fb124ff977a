"""Seeded random-init weights with the reference's state_dict keys and shapes.

There is no network for checkpoints, so tests and bench.py use weights drawn here.  The draw is
fully determined by (shapes, seed) and by torch's CPU generator, so the golden fixtures made in the
build container (oracle/make_golden.py, reference modules loaded with exactly this state_dict) can be
re-created bit-for-bit on the GPU box without the reference tree.

Unlike the reference's constructors, zero_module() parameters (unet.py:179-181, 258, 562) are drawn
non-zero; otherwise every residual branch, attention projection and the model output are exactly 0
and a parity test would pass vacuously.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch

from .arch import UNetArch, param_shapes


def init_unet_state_dict(arch: UNetArch, seed: int = 0, dtype=torch.float32) -> "OrderedDict[str, torch.Tensor]":
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in param_shapes(arch).items():
        leaf = name.rsplit(".", 1)[1]
        is_norm = (".in_layers.0." in name or ".out_layers.0." in name or ".norm." in name
                   or name.startswith("out.0.") or name.startswith("ln_model_n."))
        if is_norm:
            t = torch.randn(shape, generator=g) * 0.1
            if leaf == "weight":
                t = t + 1.0
        elif leaf == "bias":
            t = torch.randn(shape, generator=g) * 0.02
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = torch.randn(shape, generator=g) * (1.0 / math.sqrt(fan_in))
        sd[name] = t.to(dtype)
    return sd


def make_conditioning(arch: UNetArch, batch: int, seed: int = 2):
    """full_emb [B,77,d1], pooled_emb [B,d2], image_emb [B,di] ~ N(0,1)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    full = torch.randn(batch, arch.text_ctx, arch.text_dim1, generator=g)
    pooled = torch.randn(batch, arch.text_dim2, generator=g)
    image = torch.randn(batch, arch.image_dim, generator=g)
    return full, pooled, image
