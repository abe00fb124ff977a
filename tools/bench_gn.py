"""Times GroupNorm32 (+SiLU, +FiLM, zero border) at the four UNet levels: the three-kernel path (gn_stats + gn_coeff + gn_apply through
k22_groupnorm - the engine drops gn_stats when the producer delivered the sums) against the one-pass kernel fed with group sums
(k22_groupnorm_from_group_sums).  us per call, HIP events over `reps` back-to-back calls."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from kandinsky2_amd import _lib  # noqa: E402


def timed(fn, reps=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
B = 2
print(f"{'C':>5} {'H':>3} act film pad | 3-kernel us | one-pass us |  MB moved")
for C, H in [(384, 96), (768, 48), (1152, 24), (1536, 12)]:
    for act, film, pad in [(1, True, 1), (1, False, 1), (0, False, 0)]:
        x = (torch.randn(B, H, H, C, device="cuda") * 1.3 + 0.2).to(torch.bfloat16)
        gamma, beta = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
        fl = torch.randn(B, 2 * C, device="cuda") * 0.3 if film else None
        out = torch.empty(B, H + 2 * pad, H + 2 * pad, C, device="cuda", dtype=torch.bfloat16)
        xf = x.double().view(B, H * H, 32, C // 32)
        g = torch.stack([xf.sum((1, 3)), (xf * xf).sum((1, 3))], -1)
        gs = torch.stack([(g[..., 0] * 2.0 ** 24).round().long(), (g[..., 1] * 2.0 ** 20).round().long()], -1).contiguous()
        scratch = torch.empty(L.k22_groupnorm_scratch_bytes(B, C), dtype=torch.uint8, device="cuda")
        t3 = timed(lambda: _lib.check(L.k22_groupnorm(x.data_ptr(), None, C, 0, B, H, H, gamma.data_ptr(), beta.data_ptr(), _lib.ptr(fl), 0 if fl is None else 2 * C,
                                                      1e-5, act, 0, pad, scratch.data_ptr(), out.data_ptr(), _lib.K22_BF16, st)))
        t1 = timed(lambda: _lib.check(L.k22_groupnorm_from_group_sums(x.data_ptr(), C, B, H, H, gs.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _lib.ptr(fl),
                                                                      0 if fl is None else 2 * C, 1e-5, act, 0, pad, out.data_ptr(), _lib.K22_BF16, st)))
        print(f"{C:5d} {H:3d} {act:3d} {int(film):4d} {pad:3d} | {t3:11.2f} | {t1:11.2f} | {(x.numel() + out.numel()) * 2 / 1e6:8.2f}")
