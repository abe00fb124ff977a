"""Timeline of the weight-streaming kernel: per-workgroup cycle stamps (prologue, every stage, fold + store) of one 3x3 convolution.

Needs a library built with the measurement variants:  make -C kandinsky-2_amd/csrc clean all EXTRA=-DK22_STREAM_DEBUG   (rebuild without it
afterwards).  K22_STREAM_DBG bits: 8 = stamps (default here), +1 no MFMA, +2 weight ring never refilled, +4 one LDS read per phase.

    python tools/stream_trace.py Cin Cout H splitk [bm]        e.g.  1536 1536 12 5
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kandinsky2_amd import _lib  # noqa: E402

os.environ.setdefault("K22_STREAM_DBG", "8")
L = _lib.lib()
fn = L.k22_debug_set_stream_trace
fn.argtypes = [C.c_void_p]
B, ci, co, h, sk = 2, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
bm = int(sys.argv[5]) if len(sys.argv) > 5 else 160
T = torch.bfloat16
x = torch.randn(B, h + 2, h + 2, ci, device="cuda").to(T)
ws = [(torch.randn(co, 9 * ci, device="cuda") * (9 * ci) ** -0.5).to(T) for _ in range(12)]
wf = [torch.empty_like(v) for v in ws]
st = torch.cuda.current_stream().cuda_stream
for v, f in zip(ws, wf):
    _lib.check(L.k22_stream_repack(v.data_ptr(), f.data_ptr(), co, 9, ci, _lib.K22_BF16, st))
bias = torch.randn(co, device="cuda")
out = torch.empty(B, h, h, co, device="cuda", dtype=T)
part = torch.empty(16 * B * h * h * co + 64, device="cuda")
trace = torch.zeros(1024 * 16, dtype=torch.int64, device="cuda")
fn(trace.data_ptr())
_lib.check(L.k22_set_option(b"conv_algo", 20))
for i in range(12):
    _lib.check(L.k22_debug_set_stream_frag(wf[i].data_ptr(), None))
    _lib.check(L.k22_conv3x3(x.data_ptr(), ws[i].data_ptr(), bias.data_ptr(), None, out.data_ptr(), part.data_ptr(), B, h, h, ci, co, co, 0, 0, sk, bm, 0,
                             _lib.K22_BF16, st))
torch.cuda.synchronize()
t = trace.cpu().reshape(-1, 16)
t = t[t[:, 0] != 0]
print(f"== {ci}->{co}@{h} bm {bm} split-K {sk} K22_STREAM_DBG={os.environ['K22_STREAM_DBG']}: workgroups", t.shape[0])
rt0 = t[:, 11].min()
d = lambda a, b: (t[:, a] - t[:, b]).float()
print("wall-clock start spread (10 ns ticks): min/median/max", (t[:, 11] - rt0).min().item(), (t[:, 11] - rt0).median().item(), (t[:, 11] - rt0).max().item())
print("wall-clock end - first start (ticks): median/max", (t[:, 15] - rt0).median().item(), (t[:, 15] - rt0).max().item())
print("cycles per 10 ns tick ~", (d(13, 0) / (t[:, 15] - t[:, 11]).float()).median().item())
names = ["prologue -> first slab in LDS"] + [f"stage {i}" for i in range(9)]
prev = 0
for k in range(1, 11):
    if (t[:, k] == 0).all():
        break
    dd = d(k, prev)
    print(f"{names[k - 1]:32s} median {dd.median().item():8.0f} min {dd.min().clamp(min=-1).item():8.0f} max {dd.max().item():8.0f} cycles")
    prev = k
dd = d(12, prev); print(f"{'(skip phase) to fold':32s} median {dd.median().item():8.0f} max {dd.max().item():8.0f}")
dd = d(13, 12); print(f"{'fold + partial store':32s} median {dd.median().item():8.0f} max {dd.max().item():8.0f}")
dd = d(13, 0); print(f"{'whole workgroup':32s} median {dd.median().item():8.0f} max {dd.max().item():8.0f}")
print("xcc histogram", torch.bincount(t[:, 14] & 15).tolist())
