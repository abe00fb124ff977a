"""CPU restatement (TEST INFRASTRUCTURE, see oracle/__init__.py) of the img2img / inpainting pre-step helpers of
kandinsky2/utils.py: prepare_mask (:11-31, here as the equivalent gather instead of the Python scatter loop) and q_sample
(:43-54).  Pinned against the reference functions themselves by oracle/make_golden.py (prestep_case)."""
import numpy as np
import torch


def prepare_mask(mask):
    m = mask.float()[0]
    old = m[0]
    H, W = old.shape
    bad = (old != 1)
    kill = torch.zeros_like(bad)
    # a pixel whose original value is not 1 zeroes: up, left, up-left, down, right, down-right
    for dy, dx in ((-1, 0), (0, -1), (-1, -1), (1, 0), (0, 1), (1, 1)):
        ys, xs = slice(max(0, -dy), H - max(0, dy)), slice(max(0, -dx), W - max(0, dx))      # sources whose target is in bounds
        yt, xt = slice(max(0, dy), H - max(0, -dy)), slice(max(0, dx), W - max(0, -dx))
        kill[yt, xt] |= bad[ys, xs]
    out = m.clone()
    out[:, kill] = 0
    return out.unsqueeze(0)


def q_sample(x_start, t, num_steps=1000, noise=None, linear_start=0.0001, linear_end=0.02):  # model/utils.py:32-40 (sic)
    scale = 1000 / num_steps
    betas = np.linspace(scale * linear_start, scale * linear_end, num_steps, dtype=np.float64)
    ac = np.cumprod(1.0 - betas, axis=0)
    ti = torch.as_tensor(t).reshape(-1).long().numpy()
    shape = (-1,) + (1,) * (x_start.dim() - 1)
    a = torch.from_numpy(np.sqrt(ac)[ti]).float().reshape(shape)
    b = torch.from_numpy(np.sqrt(1.0 - ac)[ti]).float().reshape(shape)
    return a * x_start + b * noise
