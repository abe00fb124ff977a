"""Round-4 debug probe for the split-precision nondeterminism under two chains: puts a guard region between the two
kids' workspaces (K22_KID_GAP_MB), fills it with a pattern before every forward and reports which bytes a forward changed.
    K22_CHAINS=2 K22_KID_GAP_MB=64 python tools/chains_gap_probe.py [f16x3|bf16|fp32] [reps] [graph 0/1]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kandinsky2_amd as k22

dtn = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
use_graph = (sys.argv[3] == "1") if len(sys.argv) > 3 else False
gap = int(os.environ.get("K22_KID_GAP_MB", "0")) << 20
fx = torch.load(os.path.join(ROOT, "tests", "golden", "c4_inpaint.pt"), weights_only=False)
arch = k22.make_arch(k22.MODEL_CONFIG_2_1, inpainting=True)
sd = k22.init_unet_state_dict(arch, seed=0)
B, lat, bs = fx["B"], fx["lat"], fx["bs"]
full, pooled, image = k22.make_conditioning(arch, B, seed=2)
g = torch.Generator().manual_seed(42); x_T = torch.randn(B, 4, lat, lat, generator=g)
g3 = torch.Generator().manual_seed(3); _x = torch.randn(B, 4, lat, lat, generator=g3); ii = torch.randn(B, 4, lat, lat, generator=g3).cuda()
mm = torch.zeros(B, 1, lat, lat); mm[..., : lat // 2] = 1.0; mm = mm.cuda()
kw = dict(full_emb=full.cuda(), pooled_emb=pooled.cuda(), image_emb=image.cuda(), inpaint_image=ii * mm, inpaint_mask=mm)
DT = {"f16x3": k22.F16X3, "fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[dtn]
m = k22.Text2ImUNetHIP(arch, backend_dtype=DT, use_graph=use_graph)
m.load_state_dict(sd); m = m.to("cuda"); m.prepare(free_params=True)
x = torch.cat([x_T[:bs], x_T[:bs]], 0).cuda()
ts = fx["first_ts"].float().cuda()
out = m(x, ts, **kw)            # plans + binds the workspace
torch.cuda.synchronize()
ws = m._ws
base = (ws.data_ptr() + 255) // 256 * 256 - ws.data_ptr()
out_bytes = B * arch.out_channels * lat * lat * 4
total = ws.numel() - 256                                   # = engine ws_bytes
kid_ws = (total - 256 - out_bytes - gap) // 2              # each kid's 256-aligned size
g0 = base + kid_ws
print(f"dtype {dtn} graph {use_graph} gap {gap >> 20} MB  ws {total} kid_ws {kid_ws}  golden distance of the planning forward {(out.cpu() - fx['first_out']).abs().max().item():.3e}")
bad = 0
for r in range(reps):
    if gap:
        ws[g0: g0 + gap].fill_(0xA5)
    torch.cuda.synchronize()
    o = m(x, ts, **kw)
    torch.cuda.synchronize()
    d = (o.cpu() - fx["first_out"]).abs()
    per_img = d.flatten(1).max(1).values
    line = f"rep {r}: max|d| {d.max().item():.3e}  per image {['%.1e' % v for v in per_img.tolist()]}"
    bad += d.max().item() > 1e-4
    for im in range(B):
        if per_img[im] > 1e-4:
            nz = (d[im] > 1e-4).nonzero()
            line += f"  [image {im}: {nz.shape[0]} of {d[im].numel()} elements > 1e-4, rows {nz[:, 1].min().item()}..{nz[:, 1].max().item()} cols {nz[:, 2].min().item()}..{nz[:, 2].max().item()}, median|d| {d[im].median().item():.1e}]"
    if gap:
        gz = ws[g0: g0 + gap]
        ch = (gz != 0xA5).nonzero().flatten()
        if ch.numel():
            lo, hi = ch.min().item(), ch.max().item()
            line += f"  GAP TOUCHED: {ch.numel()} bytes, offsets {lo}..{hi} from the end of kid 0 (gap - hi = {gap - hi})"
            vals = gz[lo: lo + 32].view(torch.float32) if lo % 4 == 0 else None
            line += f"  first words {vals.tolist()[:6] if vals is not None else ''}"
        else:
            line += "  gap clean"
    print(line, flush=True)
if os.environ.get("WS_DIFF"):
    # diff the second-enqueued kid's workspace between a good and a bad forward: the lowest differing offsets name the first wrong slot
    k1 = slice(g0 + gap, g0 + gap + kid_ws) if os.environ.get("K22_CHAINS_SWAP", "0") != "2" else slice(base, base + kid_ws)
    good = badws = None
    for r in range(40):
        o = m(x, ts, **kw)
        torch.cuda.synchronize()
        e = (o.cpu() - fx["first_out"]).abs().max().item()
        snap = ws[k1].clone()
        if e <= 1e-4 and good is None:
            good = snap
        if e > 1e-4 and badws is None:
            badws = snap
        if good is not None and badws is not None:
            break
    if good is None or badws is None:
        print("WS_DIFF: did not see both a good and a bad forward")
    else:
        df = (good != badws)
        blk = df.view(-1, 256).any(1).nonzero().flatten() * 256          # differing 256-byte blocks
        print(f"WS_DIFF: {int(df.sum())} differing bytes in {blk.numel()} 256-byte blocks; first blocks at kid offsets {blk[:12].tolist()}")
        # runs of differing blocks
        b = blk.cpu().numpy()
        import numpy as np
        if b.size:
            fine = np.nonzero(np.diff(b) > 256)[0]
            fs = np.concatenate([[b[0]], b[fine + 1]]); fe = np.concatenate([b[fine], [b[-1]]])
            print("  exact runs of differing 256-byte blocks below 4 MB:", [(int(a_), int(z_) + 256) for a_, z_ in zip(fs, fe) if a_ < (4 << 20)][:60])
            lo_, hi_ = 2507008, 2513152 + 24576 * 2
            for nm, o_, n_ in (("s_t", 1327104, 20), ("s_temb row0", 2507008, 8), ("s_e1 row0", 2513152, 8), ("s_e1 row1", 2513152 + 6144, 8), ("s_emb row0", 2537728, 8), ("s_emb row1", 2537728 + 6144, 8)):
                print(f"  {nm}: good {['%.6g' % v for v in good[o_: o_ + 4 * n_].view(torch.float32).tolist()]}")
                print(f"  {nm}: bad  {['%.6g' % v for v in badws[o_: o_ + 4 * n_].view(torch.float32).tolist()]}")
            cut = np.nonzero(np.diff(b) > 65536)[0]
            starts = np.concatenate([[b[0]], b[cut + 1]]); ends = np.concatenate([b[cut], [b[-1]]])
            for s_, e_ in list(zip(starts, ends))[:40]:
                gf = good[int(s_): int(s_) + 32].view(torch.float32).tolist(); bf = badws[int(s_): int(s_) + 32].view(torch.float32).tolist()
                print(f"  differing region {int(s_):12d} .. {int(e_) + 256:12d}  ({(int(e_) + 256 - int(s_)) / 1e6:8.2f} MB)  good {['%.4g' % v for v in gf[:4]]} bad {['%.4g' % v for v in bf[:4]]}")
print(f"RESULT {dtn} gap={gap >> 20}MB graph={use_graph}: bad forwards {bad} of {reps}")
