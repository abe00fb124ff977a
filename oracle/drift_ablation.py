"""Which reduced-precision STORAGE POINTS buy the final-latent drift?  (TEST INFRASTRUCTURE, see oracle/__init__.py.)

    python oracle/drift_ablation.py [--device cpu|cuda] [--modes all:bf16,w:bf16,...] [--lat 96] [--steps 50] [--out file.json]

Runs the oracle's fp32 restatement of the reference UNet + p_sampler (oracle/unet_ref.py, oracle/diffusion_ref.py; pinned
bit-identical to the reference modules) at the benchmarked C2 shape with ROUNDING INJECTED at the places where the HIP engine's
reduced-precision mode stores a tensor, all arithmetic staying fp32 (which is what the engine does: fp32 accumulation, fp32
GroupNorm statistics, fp32 softmax, fp32 sampler).  The distance of each run's final latent from the committed reference
golden (tests/golden/c2_text2img.pt: reference create_model + p_sample_loop, fp32) says what that set of storage points costs.
A mode is a '+'-joined subset of

    w  block weights: 3x3 / 1x1 convolutions, qkv / proj_out / encoder_kv, emb_layers, to_model_dim_n   (engine arena, type T)
    h  the residual stream: stem output, every ResBlock / AttentionBlock output (after the residual add), the skip stack,
       the resampled x of up / down blocks
    g  GroupNorm(+SiLU, +FiLM) outputs = the operands of every 3x3 convolution and of qkv
    u  the intermediate convolution output of a ResBlock (in_layers conv -> out_layers GroupNorm)
    a  attention operands: ctx / encoder_kv, q, k, v, softmax probabilities P, the attention output

    s  the input of a ResBlock's 1x1 skip_connection, AS AN MFMA OPERAND (the tensor itself stays what `h` made it)

followed by ':bf16' or ':fp16' ('all' = w+h+g+u+a, the engine's storage map; 'none' = no rounding = the oracle itself).

Round 4 - SPLIT-PRECISION candidates.  The tensors stay fp32 in memory; what is emulated is the MFMA OPERAND:
    :x3      x -> hi + lo with hi = rne_fp16(x), lo = rne_fp16(x - hi)  (fp16 subnormals kept: 2^-25 absolute floor), i.e. what
             three v_mfma_f32_32x32x16_f16 (hi.hi + hi.lo + lo.hi, fp32 accumulate) see of an fp32 operand; the dropped lo.lo
             term is <= 2^-24 of a product and is not emulated
    :x3w     the same on x * 2^8 (the packed weights are pre-scaled by an exact power of two so that lo stays normal), / 2^8
    :bf16x3  hi + lo with bf16 halves (16 mantissa bits)
A mode may join several 'kinds:type' parts with '/', e.g. 'w:x3w/g+a+s:x3' = weights split, every activation operand split (three
MFMAs); 'w:x3w/g+a+s:fp16' = weights split, activations rounded to fp16 as operands (two MFMAs).
Round 4 - WINOGRAD F(2x2, 3x3) numerics (VERDICT r3 #5a, the go / no-go half that needs no GPU):
    wino:T   every stride-1 3x3 convolution of the ResBlocks is evaluated as the engine would evaluate it on 16-bit MFMAs: the filter
             transform U = G g G^T in fp32 from the fp32 weights, THEN rounded to T (pack time); the input transform V = B^T d B in fp32
             from the operand as stored (kind g), THEN rounded to T (what the LDS fill would write); 16 products sum_c U V in fp32 (MFMA
             accumulate); the output transform A^T M A in fp32.  The 3x3 weights are not pre-rounded by kind w in this mode.
Nothing here is product code; bench.py / tests only read the committed result (tests/golden/drift_ablation.json).
"""
import argparse
import json
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import kandinsky2_amd as k22  # noqa: E402
from oracle import diffusion_ref, unet_ref  # noqa: E402


X3W_SCALE = 256.0


def _split(x, dt):
    hi = x.to(dt).float()
    return hi + (x - hi).to(dt).float()


ROUNDERS = {
    "bf16": lambda x: x.to(torch.bfloat16).float(),
    "fp16": lambda x: x.to(torch.float16).float(),
    "": lambda x: x.to(torch.bfloat16).float(),
    "x3": lambda x: _split(x, torch.float16),
    "x3w": lambda x: _split(x * X3W_SCALE, torch.float16) / X3W_SCALE,
    "bf16x3": lambda x: _split(x, torch.bfloat16),
}


class Rounder:
    def __init__(self, mode):
        self.fn = {}
        for part in mode.split("/"):
            kinds, _, dt = part.partition(":")
            if kinds == "all":
                kinds = "w+h+g+u+a"
            if kinds == "none":
                continue
            for k in kinds.split("+"):
                self.fn[k] = ROUNDERS[dt]
        self.kinds = set(self.fn)

    def __call__(self, x, kind, sub=""):
        # round 5: a kind may be narrowed to a position ('g1' = in_layers conv input, 'g2' = out_layers conv input, 'gq' = qkv input,
        # 'go' = the out head's conv input) and / or to a resolution ('g@96', 'g1@24': the tensor's width) - the per-layer precision
        # plan of the K22_F16X2 engine is read off such runs (tests/golden/drift_ablation_x2.json)
        res = "@%d" % x.shape[-1] if x.dim() == 4 else ""
        for key in (kind, kind + sub, kind + res, kind + sub + res):
            f = self.fn.get(key)
            if f is not None:
                return f(x)
        return x


W_SUFFIXES = (".in_layers.2.weight", ".out_layers.3.weight", ".skip_connection.weight", ".qkv.weight", ".proj_out.weight",
              ".encoder_kv.weight", ".emb_layers.1.weight")


def block_resolutions(arch, lat):
    """block prefix -> the latent resolution its convolutions / attention run at (a down block's convolutions run AFTER the pooling, an
    up block's AFTER the upsampling: unet.py:157-164)"""
    res, out = lat, {}
    for b in arch.blocks:
        if b[0] == "res" and b[4] == 1:
            res //= 2
        elif b[0] == "res" and b[4] == 2:
            res *= 2
        out[b[1]] = res
    return out


def round_weights(sd, r, arch=None, lat=0):
    """kind 'w' rounds every block weight; 'w@<res>' (round 5) only - or, beside a plain 'w', differently - those of the blocks whose
    convolutions run at that latent resolution ('w:x3w/w@12:fp16' = weights split everywhere except plain fp16 at 12x12)."""
    out = {}
    bres = block_resolutions(arch, lat) if arch is not None else {}
    has_w = any(k == "w" or k.startswith("w@") for k in r.kinds)
    for k, v in sd.items():
        if "wino" in r.kinds and k.endswith((".in_layers.2.weight", ".out_layers.3.weight")):
            out[k] = v                       # transformed in fp32 first, rounded after (conv3)
        elif has_w and (k.endswith(W_SUFFIXES) or k == "to_model_dim_n.weight") and k.split(".")[0] in (
                "input_blocks", "middle_block", "output_blocks", "to_model_dim_n"):
            parts = k.split(".")
            pfx = next((".".join(parts[:n]) for n in (3, 2) if ".".join(parts[:n]) in bres), None)
            f = r.fn.get("w@%d" % bres[pfx]) if pfx is not None else None
            f = f or r.fn.get("w")
            out[k] = f(v) if f is not None else v
        else:
            out[k] = v
    return out


def gn(x, w, b, swish):
    y = F.group_norm(x, 32, w, b, eps=1e-5)
    return F.silu(y) if swish else y


_BT = torch.tensor([[1., 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
_G = torch.tensor([[1., 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
_AT = torch.tensor([[1., 1, 1, 0], [0, 1, -1, -1]])
_UCACHE = {}


def conv3(x, w, b, r):
    """3x3 'same' convolution; under kind 'wino' as Winograd F(2x2, 3x3) with the operand roundings of a 16-bit MFMA kernel."""
    if "wino" not in r.kinds:
        return F.conv2d(x, w, b, padding=1)
    B, C, H, W = x.shape
    O = w.shape[0]
    assert H % 2 == 0 and W % 2 == 0
    key = (w.data_ptr(), r.fn["wino"])
    if key not in _UCACHE:
        _UCACHE[key] = r(torch.einsum("ij,ocjk,lk->iloc", _G.to(w), w, _G.to(w)), "wino").reshape(16, O, C).contiguous()
    U = _UCACHE[key]
    th, tw = H // 2, W // 2
    d = F.unfold(F.pad(x, (1, 1, 1, 1)), kernel_size=4, stride=2).view(B, C, 4, 4, th * tw)
    V = r(torch.einsum("ij,bcjkt,lk->ilcbt", _BT.to(x), d, _BT.to(x)), "wino").reshape(16, C, B * th * tw)
    M = torch.bmm(U, V).view(4, 4, O, B, th, tw)
    Y = torch.einsum("pi,ilobyx,ql->boypxq", _AT.to(x), M, _AT.to(x)).reshape(B, O, H, W)
    return Y + b.view(1, -1, 1, 1)


def res_block(sd, pfx, x, emb, updown, r):
    h = gn(x, sd[pfx + ".in_layers.0.weight"], sd[pfx + ".in_layers.0.bias"], True)
    if updown == 1:
        h, x = F.avg_pool2d(h, 2), r(F.avg_pool2d(x, 2), "h")
    elif updown == 2:
        h, x = F.interpolate(h, scale_factor=2, mode="nearest"), F.interpolate(x, scale_factor=2, mode="nearest")
    h = r(h, "g", "1")
    h = r(conv3(h, sd[pfx + ".in_layers.2.weight"], sd[pfx + ".in_layers.2.bias"], r), "u")
    e = F.linear(F.silu(emb), sd[pfx + ".emb_layers.1.weight"], sd[pfx + ".emb_layers.1.bias"])[..., None, None]
    scale, shift = torch.chunk(e, 2, dim=1)
    h = gn(h, sd[pfx + ".out_layers.0.weight"], sd[pfx + ".out_layers.0.bias"], False) * (1 + scale) + shift
    h = conv3(r(F.silu(h), "g", "2"), sd[pfx + ".out_layers.3.weight"], sd[pfx + ".out_layers.3.bias"], r)
    if (pfx + ".skip_connection.weight") in sd:   # fused into the second convolution's accumulator by the engine: no rounding
        x = F.conv2d(r(x, "s"), sd[pfx + ".skip_connection.weight"], sd[pfx + ".skip_connection.bias"])
    return r(x + h, "h")


def attention_block(sd, pfx, x, encoder_out, r, head_ch=64):
    b, c, hh, ww = x.shape
    n = r(gn(x, sd[pfx + ".norm.weight"], sd[pfx + ".norm.bias"], False), "g", "q").view(b, c, -1)
    qkv = r(F.conv1d(n, sd[pfx + ".qkv.weight"], sd[pfx + ".qkv.bias"]), "a")
    ekv = r(F.conv1d(encoder_out, sd[pfx + ".encoder_kv.weight"], sd[pfx + ".encoder_kv.bias"]), "a")
    n_heads = c // head_ch
    bs, width, length = qkv.shape
    ch = width // (3 * n_heads)
    q, k, v = qkv.reshape(bs * n_heads, ch * 3, length).split(ch, dim=1)
    ek, ev = ekv.reshape(bs * n_heads, ch * 2, -1).split(ch, dim=1)
    k = torch.cat([ek, k], dim=-1)
    v = torch.cat([ev, v], dim=-1)
    scale = 1 / math.sqrt(math.sqrt(ch))
    w = torch.softmax(torch.einsum("bct,bcs->bts", q * scale, k * scale), dim=-1)
    a = r(torch.einsum("bts,bcs->bct", r(w, "a"), v), "a").reshape(bs, -1, length)
    h = F.conv1d(a, sd[pfx + ".proj_out.weight"], sd[pfx + ".proj_out.bias"])
    return r(x + h.reshape(b, c, hh, ww), "h")


@torch.no_grad()
def unet_forward(sd, arch, x, timesteps, cond, r):
    xf_proj, xf_out = cond
    emb = unet_ref.timestep_embedding(timesteps.cpu(), arch.model_channels).to(x.device)
    emb = F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    emb = emb + xf_proj
    h = x.float()
    hs = []
    blocks = arch.blocks

    def apply(b, h):
        if b[0] == "stem":
            return r(F.conv2d(h, sd[b[1] + ".weight"], sd[b[1] + ".bias"], padding=1), "h")
        if b[0] == "res":
            return res_block(sd, b[1], h, emb, b[4], r)
        return attention_block(sd, b[1], h, xf_out, r, arch.num_head_channels)

    n_in = sum(1 for b in blocks if b[1].startswith("input_blocks."))
    cur = None
    for b in blocks[:n_in]:
        seq = b[1].split(".")[1]
        if cur is not None and seq != cur:
            hs.append(h)
        cur = seq
        h = apply(b, h)
    hs.append(h)
    i = n_in
    while blocks[i][1].startswith("middle_block."):
        h = apply(blocks[i], h)
        i += 1
    cur = None
    for b in blocks[i:]:
        seq = b[1].split(".")[1]
        if seq != cur:
            h = torch.cat([h, hs.pop()], dim=1)
            cur = seq
        h = apply(b, h)
    h = r(gn(h, sd["out.0.weight"], sd["out.0.bias"], True), "g", "o")
    return F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)


def run_mode(mode, sd0, arch, fx, dev):
    r = Rounder(mode)
    B, lat, steps, bs = fx["B"], fx["lat"], fx["steps"], fx["bs"]
    sd = round_weights(sd0, r, arch, lat)
    full, pooled, image = [t.to(dev) for t in k22.make_conditioning(arch, B, seed=2)]
    g = torch.Generator().manual_seed(42)
    x_T = torch.randn(B, 4, lat, lat, generator=g).to(dev)
    noise_seq = torch.randn(steps, B, 4, lat, lat, generator=g).to(dev)
    xf_proj, xf_out = unet_ref.text_emb(sd, arch, full, pooled, image)
    cond = (xf_proj, r(xf_out, "a"))
    first = unet_forward(sd, arch, torch.cat([x_T[:bs], x_T[:bs]], 0), fx["first_ts"].float().to(dev), cond, r)
    d0 = (first.cpu() - fx["first_out"]).abs().max().item()
    rep = {"mode": mode, "first_forward_max_abs": d0, "first_forward_rel": d0 / fx["first_out"].abs().max().item(),
           "first_forward_rms": (first.cpu() - fx["first_out"]).pow(2).mean().sqrt().item()}
    if steps > 0 and not os.environ.get("K22_ABLATE_FIRST_ONLY"):
        d = diffusion_ref.RefDiffusion(steps)
        x = d.p_sample_loop(lambda xc, t: unet_forward(sd, arch, xc, t.to(dev), cond, r), x_T, noise_seq, fx["guidance"])
        dd = x.cpu() - fx["final"]
        rep.update(final_max_abs=dd.abs().max().item(), final_rms=dd.pow(2).mean().sqrt().item())
    return rep


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cpu")
    ap.add_argument("--modes", default="none,all:bf16,w:bf16,h+g+u+a:bf16,w+g+u+a:bf16,all:fp16")
    ap.add_argument("--golden", default=os.path.join(ROOT, "tests", "golden", "c2_text2img.pt"))
    ap.add_argument("--out", default="")
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    if a.threads:
        torch.set_num_threads(a.threads)
    fx = torch.load(a.golden, weights_only=False)
    arch = k22.make_arch(k22.MODEL_CONFIG_2_1)
    sd0 = {k: v.to(a.device) for k, v in k22.init_unet_state_dict(arch, seed=0).items()}
    if a.device != "cpu":
        torch.backends.cuda.matmul.allow_tf32 = False
        torch.backends.cudnn.allow_tf32 = False
    reps = []
    for mode in a.modes.split(","):
        t0 = time.time()
        rep = run_mode(mode, sd0, arch, fx, a.device)
        rep["seconds"] = round(time.time() - t0, 1)
        print(json.dumps(rep), flush=True)
        reps.append(rep)
        if a.out:
            with open(a.out, "w") as f:
                json.dump({"config": f"oracle UNet 1.23B + p_sampler, CFG batch {fx['B']}x4x{fx['lat']}x{fx['lat']}, {fx['steps']} steps, "
                           f"device {a.device}; distances from tests/golden/c2_text2img.pt (reference fp32)", "runs": reps}, f, indent=1)
