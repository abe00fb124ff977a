"""us per k22_sampler_step (x0 + exact 99.5th-percentile threshold + final) at the C2 latent shape.  (Exactness of the threshold on
adversarial key distributions is a test: tests/test_kernels_gpu.py::test_sampler_threshold_is_the_exact_order_statistic_on_adversarial_keys.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kandinsky2_amd as k22  # noqa: E402
from kandinsky2_amd import _lib  # noqa: E402

L = _lib.lib()
N, H, W = 2, 96, 96
d = k22.create_gaussian_diffusion(**dict(k22.DIFFUSION_CONFIG_2_1, timestep_respacing="50"))
table = torch.from_numpy(d.step_table()).cuda()
scratch = torch.empty(L.k22_sampler_scratch_bytes(N, H * W), dtype=torch.uint8, device="cuda")
lo, gamma = k22.percentile_index(4 * H * W)
g = torch.Generator().manual_seed(0)
x, mo, nz = (torch.randn(N, c, H, W, generator=g).cuda() for c in (4, 8, 4))
xo, x0o = torch.empty(N, 4, H, W, device="cuda"), torch.empty(N, 4, H, W, device="cuda")
st = torch.cuda.current_stream().cuda_stream


def step(s=25):
    _lib.check(L.k22_sampler_step(x.data_ptr(), mo.data_ptr(), nz.data_ptr(), None, None, table.data_ptr(), s, 4.0, 1, -2.0, 2.0, lo, gamma,
                                  scratch.data_ptr(), xo.data_ptr(), x0o.data_ptr(), N, H * W, st))


for _ in range(5):
    step()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(200):
    step()
b.record()
torch.cuda.synchronize()
print(f"k22_sampler_step @ 2x4x96x96: {a.elapsed_time(b) / 200 * 1e3:.1f} us per step (three kernels)")
