#!/usr/bin/env python
"""Prior transformer forward time at bs=1 (CFG batch 2), full 1.02 B-parameter configuration."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kandinsky2_amd as k22
hp = k22.PRIOR_HPARAMS_2_1
m = k22.PriorDiffusionModelHIP(hp, backend_dtype=torch.bfloat16)
m.load_state_dict(k22.init_prior_state_dict(hp, seed=0))
m = m.to("cuda")
N = 2
x, t = torch.randn(N, 768, device="cuda"), torch.full((N,), 500.0, device="cuda")
te, tq = torch.randn(N, 768, device="cuda"), torch.randn(N, 77, 768, device="cuda")
mask = torch.ones(N, 77, dtype=torch.bool, device="cuda")
out = m.transformer(x, t, te, tq, mask)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    out = m.transformer(x, t, te, tq, mask)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
wbytes = sum(v.numel() for k, v in m.state_dict().items() if k.startswith("model.transformer") and k.endswith("weight") and v.dim() == 2) * 2
print(f"prior forward bs=1 (2x81 tokens) bf16: {ms:.3f} ms  ({wbytes / ms / 1e6:.0f} GB/s of transformer weights, finite={bool(torch.isfinite(out).all())})")
print(m.tuning_report())
for it in range(2):   # the first call plans / tunes / captures the graph of the sampling batch shape
    t0 = time.perf_counter()
    s = m(te, tq, mask, torch.tensor([4.0], device="cuda"), timestep_respacing="25")
    torch.cuda.synchronize()
    print(f"prior 25-step sample ({'first call: plan + tile selection + graph capture' if it == 0 else 'steady state'}): {(time.perf_counter() - t0) * 1e3:.1f} ms")
