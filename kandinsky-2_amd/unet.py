"""Drop-in replacement of the reference's Text2ImUNet / InpaintText2ImUNet backed by the HIP engine.

Same constructor hyper-parameters (through create_model), same state_dict keys and shapes (reference
checkpoints load with load_state_dict), same call surface as the reference driver uses
(kandinsky2/kandinsky2_1_model.py:90-104, 208-225, 246-257):
    model(x, timesteps, full_emb=, pooled_emb=, image_emb= [, inpaint_image=, inpaint_mask=]) -> [B,8,h,w]
    model.del_cache(), model.convert_to_fp16(), model.eval(), model.to(device), model.dtype
Everything inside forward() runs in libk22hip.so; there is no PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch
import torch.nn as nn

from . import _lib
from .arch import UNetArch, make_arch, param_shapes
from .pack import pack_arena


def _register(root: nn.Module, dotted: str, param: nn.Parameter) -> None:
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if p not in m._modules:
            m.add_module(p, nn.Module())
        m = m._modules[p]
    m.register_parameter(parts[-1], param)


class Text2ImUNetHIP(nn.Module):
    """MI355X-native Text2ImUNet (kandinsky2/model/text2im_model2_1.py:13-103).

    backend_dtype: torch.bfloat16 (BASELINE's dtype, bf16 MFMA), torch.float16 (the reference's own use_fp16 mode: same MFMA rate,
    3 more mantissa bits - the mode that holds <= 3e-3 on the 50-step final latent), torch.float32 (parity path, exact-fp32 MFMA) or
    "f16x3" (kandinsky2_amd.F16X3, split precision: fp32 tensors, every MFMA operand as an fp16 (hi, lo) pair, three fp16 MFMAs per
    product - within 1e-3 of the reference p_sampler's final latent, BASELINE.json's gate, at several times the fp32 engine's speed).
    use_graph: replay each forward as one captured hipGraph.
    chains: 2 = an even batch runs as TWO half-batch engines (same arena, own workspaces and graphs) on two streams of the device - the two
    halves of the classifier-free-guidance pair never interact inside the UNet (GroupNorm and attention are per sample), so one chain's
    latency-bound stretches (GroupNorm coefficient / apply launches, the 20-us GEMMs of the AttentionBlocks, split-K finishes) overlap
    the other's MFMA-bound ones.  Default: env K22_CHAINS, else 1.  Each half runs the tile-table lines of ITS problem sizes, so the bits of a
    two-chain run differ from a one-chain run of the same inputs (other split-K factors); the distances from the reference do not.
    """

    def __init__(self, arch: UNetArch, backend_dtype: torch.dtype = torch.bfloat16, use_graph: bool = True,
                 cache_text_emb: bool = True, meta_params: bool = False, chains: Optional[int] = None):
        super().__init__()
        if backend_dtype not in (torch.bfloat16, torch.float16, torch.float32, _lib.F16X3, _lib.F16X2):
            raise ValueError('backend_dtype must be torch.bfloat16, torch.float16, torch.float32, "f16x3" (split precision) or "f16x2" '
                             '(asymmetric split: per-op precision plan)')
        self.arch = arch
        self.backend_dtype = backend_dtype
        self.use_graph = use_graph
        self.cache_text_emb = cache_text_emb
        self.dtype = torch.float32  # public tensors are fp32, like the reference's x / output
        self.model_channels = arch.model_channels
        for name, shape in param_shapes(arch).items():
            t = torch.empty(shape, device="meta") if meta_params else torch.zeros(shape)
            _register(self, name, nn.Parameter(t, requires_grad=False))
        self.chains = int(os.environ.get("K22_CHAINS", "1")) if chains is None else int(chains)
        if self.chains not in (1, 2):
            raise ValueError("chains must be 1 or 2")
        self._handle: Optional[C.c_void_p] = None
        self._handle2: Optional[C.c_void_p] = None      # second half-batch engine of the two-chain mode
        self._ws2 = None
        self._side = None                               # its stream
        self._arena = None
        self._weights_keepalive = None
        self._ws = None
        self._plan_key = None
        self._cond_key = None
        self.cache = None  # mirrors the reference attribute; holds the key of the cached conditioning

    # ---- reference-compatible no-ops -------------------------------------------------------------
    def convert_to_fp16(self):
        """The reference casts its conv / attention / head weights to fp16 here and computes the torso in fp16
        (unet.py:566-572, text2im_model2_1.py:49-55).  Same meaning here: the engine's storage and MFMA operand type becomes fp16
        (fp32 accumulation, GroupNorm statistics, softmax and sampler state stay fp32, as in the reference's fp16 mode).  An engine
        built with backend_dtype=torch.float32 is the parity path and stays fp32 until convert_to_fp16() is asked for explicitly."""
        if self.backend_dtype != torch.float16:
            self._need_fp32_params("convert_to_fp16", "backend_dtype=torch.float16")
            self.backend_dtype = torch.float16
            self._release()   # re-pack lazily in fp16
        return self

    def convert_to_fp32(self):
        """unet.py:574-580: back to the fp32 parity path"""
        if self.backend_dtype != torch.float32:
            self._need_fp32_params("convert_to_fp32", "backend_dtype=torch.float32")
            self.backend_dtype = torch.float32
            self._release()
        return self

    def _need_fp32_params(self, what: str, ctor_hint: str):
        """A dtype change re-packs the arena from the fp32 parameters: refuse when they are gone (released after packing, never
        materialised - meta_params - or replaced by an arena adopted from a broadcast), instead of dropping a working engine."""
        ps = list(self.parameters())
        if any(p.is_meta for p in ps) or (getattr(self, "_adopted", False) and not any(p.numel() for p in ps)) or getattr(self, "_arena_adopted", False):
            raise RuntimeError(f"{what}: this module has no fp32 parameters to re-pack from (prepare(free_params=True), meta_params=True or "
                               f"an adopted / broadcast arena); construct it with {ctor_hint} instead")

    def del_cache(self):
        self.cache = None
        self._cond_key = None

    # ---- engine management -------------------------------------------------------------------------
    def _release(self):
        if self._handle is not None:
            _lib.lib().k22_unet_destroy(self._handle)
            self._handle = None
        if getattr(self, "_handle2", None) is not None:
            _lib.lib().k22_unet_destroy(self._handle2)
            self._handle2 = None
        self._arena = None
        self._ws = None
        self._ws2 = None
        self._loop_bufs = None
        self._plan_key = None
        self._cond_key = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def load_state_dict(self, *args, **kwargs):
        r = super().load_state_dict(*args, **kwargs)
        self._release()  # weights changed: re-pack lazily
        return r

    def _apply(self, fn, *args, **kwargs):
        r = super()._apply(fn, *args, **kwargs)
        self._release()  # device move: re-pack lazily
        return r

    def arena_table(self):
        """name -> (byte offset, byte size) of the packed arena, derived from shapes only."""
        meta_sd = {k: torch.empty(v, device="meta") for k, v in param_shapes(self.arch).items()}
        return pack_arena(self.arch, meta_sd, self.backend_dtype, "meta")[1]

    def arena_bytes(self) -> int:
        t = self.arena_table()
        last_off, last_n = list(t.values())[-1]
        return (last_off + (last_n + 255) // 256 * 256) + 256

    def _engine_config(self):
        """K22UNetConfig of this architecture / dtype (host-side only: no device is touched)"""
        a = self.arch
        cfg = _lib.K22UNetConfig()
        cfg.dtype = _lib.dtype_code(self.backend_dtype)
        cfg.in_channels = a.in_channels
        cfg.model_channels = a.model_channels
        cfg.out_channels = a.out_channels
        cfg.num_res_blocks = a.num_res_blocks
        cfg.n_levels = len(a.channel_mult)
        for i, v in enumerate(a.channel_mult):
            cfg.channel_mult[i] = v
        cfg.n_attention_ds = len(a.attention_ds)
        for i, v in enumerate(a.attention_ds):
            cfg.attention_ds[i] = v
        cfg.num_head_channels = a.num_head_channels
        cfg.ctx_dim = a.model_dim
        cfg.ctx_len = a.ctx_len
        cfg.n_image_embs = a.num_image_embs
        cfg.text_dim1 = a.text_dim1
        cfg.text_dim2 = a.text_dim2
        cfg.image_dim = a.image_dim
        cfg.head_type = 1 if a.head == "2.2" else 0
        cfg.hint_channels = a.hint_channels
        return cfg

    def prepare(self, arena: Optional[torch.Tensor] = None, free_params: bool = False):
        """Packs the weights (or adopts a broadcast arena) and creates the native engine."""
        dev = arena.device if arena is not None else next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("Text2ImUNetHIP runs on the GPU only (no CPU fallback): move it with .to('cuda')")
        L = _lib.lib()
        self._release()
        self._arena_adopted = arena is not None
        if arena is None:
            arena, table = pack_arena(self.arch, self.state_dict(), self.backend_dtype, dev)
        else:
            table = self.arena_table()
            if arena.numel() < self.arena_bytes() or arena.dtype != torch.uint8:
                raise ValueError("arena does not match this architecture/dtype")
        self._arena = arena
        cfg = self._engine_config()
        base = arena.data_ptr()
        arr = (_lib.K22Weight * len(table))()
        names = []
        for i, (name, (off, _n)) in enumerate(table.items()):
            nb = name.encode()
            names.append(nb)
            arr[i].name = nb
            arr[i].ptr = base + off
        h = C.c_void_p()
        _lib.check(L.k22_unet_create(C.byref(cfg), arr, len(table), C.byref(h)))
        self._handle = h
        if self.chains == 2:
            if L.k22_build_flags() & 1:
                raise RuntimeError("chains=2 overlaps two engines on one device: this libk22hip.so was built with packed-fp32 instructions (NOPK=0)")
            h2 = C.c_void_p()
            _lib.check(L.k22_unet_create(C.byref(cfg), arr, len(table), C.byref(h2)))
            self._handle2 = h2
        if free_params:
            for p in self.parameters():
                p.data = torch.empty(0, device=dev)
        self._adopted = True
        return self

    def _ensure_plan(self, B: int, H: int, W: int):
        if self._handle is None:
            self.prepare()
        key = (B, H, W)
        if self._plan_key != key:
            self._plan_key = None   # a failed plan / bind leaves the native engine without a plan: never skip re-planning after it
            self._cond_key = None
            L = _lib.lib()
            nbytes = C.c_size_t()
            pb = B // 2 if self._chained(B) else B
            _lib.check(L.k22_unet_plan(self._handle, pb, H, W, C.byref(nbytes)))
            self._ws = torch.empty(nbytes.value + 256, dtype=torch.uint8, device=self._arena.device)
            p = self._ws.data_ptr()
            al = (p + 255) // 256 * 256
            _lib.check(L.k22_unet_bind(self._handle, al, nbytes.value))
            if self._chained(B):
                _lib.check(L.k22_unet_plan(self._handle2, pb, H, W, C.byref(nbytes)))
                self._ws2 = torch.empty(nbytes.value + 256, dtype=torch.uint8, device=self._arena.device)
                _lib.check(L.k22_unet_bind(self._handle2, (self._ws2.data_ptr() + 255) // 256 * 256, nbytes.value))
                if self._side is None:
                    self._side = torch.cuda.Stream(device=self._arena.device)
            self._plan_key = key
            self._cond_key = None

    def _chained(self, B: int) -> bool:
        return self.chains == 2 and self._handle2 is not None and B >= 2 and B % 2 == 0

    def _parts(self, B: int):
        """(handle, first row, rows, stream handle) of every chain of a batch of B; the side stream is ordered behind the caller's first"""
        cur = torch.cuda.current_stream()
        if not self._chained(B):
            return [(self._handle, 0, B, cur.cuda_stream)]
        self._side.wait_stream(cur)
        return [(self._handle, 0, B // 2, cur.cuda_stream), (self._handle2, B // 2, B // 2, self._side.cuda_stream)]

    def _join(self, B: int):
        if self._chained(B):
            torch.cuda.current_stream().wait_stream(self._side)

    def num_ops(self) -> int:
        return _lib.lib().k22_unet_num_ops(self._handle) if self._handle is not None else 0

    def profile(self, reps: int = 3):
        """Per-op-class device time of one forward (HIP events around every op, eager replay)."""
        kinds = ["conv3x3", "gemm", "groupnorm", "attention", "other"]
        tot = {k: dict(ms=0.0, flops=0.0, bytes=0.0, launches=0) for k in kinds}
        # two chains: the halves are timed one after the other, op by op - ISOLATED device times of both, summed (the step itself overlaps
        # them, so the class times then add up to more than the step)
        handles = [self._handle] + ([self._handle2] if self._plan_key is not None and self._chained(self._plan_key[0]) else [])
        for h in handles:
            ms, fl, by = (C.c_double * 5)(), (C.c_double * 5)(), (C.c_double * 5)()
            ln = (C.c_int * 5)()
            _lib.check(_lib.lib().k22_unet_profile(h, reps, ms, fl, by, ln, _lib.current_stream()))
            for i, k in enumerate(kinds):
                tot[k]["ms"] += ms[i]; tot[k]["flops"] += fl[i]; tot[k]["bytes"] += by[i]; tot[k]["launches"] += ln[i]
        return tot

    def tuning_report(self) -> str:
        """Tile configuration chosen (by measurement at the first forward) for every distinct conv / GEMM problem."""
        buf = C.create_string_buffer(1 << 16)
        _lib.check(_lib.lib().k22_unet_tuning_report(self._handle, buf, len(buf)))
        return buf.value.decode()

    def workspace_bytes(self) -> int:
        return 0 if self._ws is None else self._ws.numel()

    def set_condition(self, full_emb, pooled_emb, image_emb):
        """Text2ImUNet.get_text_emb (text2im_model2_1.py:57-80) + hoisted encoder_kv projections."""
        L = _lib.lib()
        if self._plan_key is None:
            raise RuntimeError("set_condition: no plan yet (call forward, which plans for its input shape)")
        B, a = self._plan_key[0], self.arch
        ntext = a.ctx_len - a.num_image_embs
        for name, t, want in (("full_emb", full_emb, (B, ntext, a.text_dim1)), ("pooled_emb", pooled_emb, (B, a.text_dim2)),
                              ("image_emb", image_emb, (B, a.image_dim))):
            if tuple(t.shape) != want:   # the engine copies B * ... bytes from these pointers: a wrong batch would read out of bounds
                raise ValueError(f"{name} must have shape {want} for a batch of {B}; got {tuple(t.shape)}")
            if t.device.type != "cuda":
                raise RuntimeError(f"{name} must be on the GPU (no CPU fallback)")
        f = full_emb.detach().float().contiguous()
        p = pooled_emb.detach().float().contiguous()
        i = image_emb.detach().float().contiguous()
        for h, r0, n, st in self._parts(B):
            _lib.check(L.k22_unet_set_condition(h, f[r0:r0 + n].data_ptr(), p[r0:r0 + n].data_ptr(), i[r0:r0 + n].data_ptr(), st))
        self._join(B)
        return self

    @torch.no_grad()
    def forward(self, x, timesteps, full_emb=None, pooled_emb=None, image_emb=None, inpaint_image=None, inpaint_mask=None):
        if x.device.type != "cuda":
            raise RuntimeError("Text2ImUNetHIP.forward: input must be on the GPU (no CPU fallback)")
        B, Cx, H, W = x.shape
        if Cx != 4:
            raise ValueError("expected a 4-channel latent")
        self._ensure_plan(B, H, W)
        # the reference caches the conditioning after the first call until del_cache() (text2im_model2_1.py:58-59, 82-83)
        if self._cond_key is None or not self.cache_text_emb:
            if full_emb is None or pooled_emb is None or image_emb is None:
                raise ValueError("full_emb, pooled_emb and image_emb are required")
            self.set_condition(full_emb, pooled_emb, image_emb)
            self._cond_key = True
            self.cache = {"cached": True}
        if timesteps.numel() != B:
            raise ValueError(f"timesteps must hold one value per batch element ({B}); got {tuple(timesteps.shape)}")
        xf = x.detach().float().contiguous()
        tf = timesteps.detach().float().reshape(B).contiguous().to(x.device)
        img = msk = None
        if self.arch.inpainting:
            # InpaintText2ImUNet.forward defaults (text2im_model2_1.py:146-150); operands are broadcast the way the
            # reference's torch.cat([x, inpaint_image * inpaint_mask, inpaint_mask], dim=1) would need them ([B,4,H,W], [B,1,H,W])
            try:
                img = torch.zeros_like(xf) if inpaint_image is None else inpaint_image.detach().float().to(x.device).expand(B, 4, H, W).contiguous()
                msk = torch.zeros_like(xf[:, :1]) if inpaint_mask is None else inpaint_mask.detach().float().to(x.device).expand(B, 1, H, W).contiguous()
            except RuntimeError as e:
                raise ValueError(f"inpaint_image / inpaint_mask do not match x {tuple(x.shape)}: {e}") from None
        elif inpaint_image is not None or inpaint_mask is not None:
            raise ValueError("inpaint_image / inpaint_mask given to a text2img UNet (create it with inpainting=True)")
        out = torch.empty(B, self.arch.out_channels, H, W, dtype=torch.float32, device=x.device)
        for h, r0, n, st in self._parts(B):      # one chain, or the two half-batch chains side by side (each replays its own graph on its stream)
            _lib.check(_lib.lib().k22_unet_forward(
                h, xf[r0:r0 + n].data_ptr(), tf[r0:r0 + n].data_ptr(), _lib.ptr(None if img is None else img[r0:r0 + n]),
                _lib.ptr(None if msk is None else msk[r0:r0 + n]), out[r0:r0 + n].data_ptr(), 1 if self.use_graph else 0, st))
        self._join(B)
        return out


    @torch.no_grad()
    def sample_loop(self, x, ts_rows, noise_seq, table, table_rows, guidance_scale, clamp, pct, *, full_emb=None, pooled_emb=None,
                    image_emb=None, inpaint_image=None, inpaint_mask=None, init_img=None, img_mask=None):
        """The whole guided p_sampler loop as ONE hipGraph replay (k22_unet_sample_loop); returns the final latent [N,4,h,w].
        x [N,4,h,w] = x_T; ts_rows [n_steps, N] and noise_seq [n_steps, N,4,h,w] in execution order; table [T,8] + table_rows (host ints):
        the schedule row of every step.  The operands are copied into buffers this module OWNS, so that a second generation of the
        same shape and step count replays the captured graph (its nodes hold those addresses) instead of capturing 20 000 nodes again."""
        B, Cx, H, W = x.shape
        if Cx != 4 or x.device.type != "cuda":
            raise ValueError("sample_loop: x must be [N,4,h,w] on the GPU")
        n_steps = len(table_rows)
        if tuple(ts_rows.shape) != (n_steps, B) or tuple(noise_seq.shape) != (n_steps, B, 4, H, W):
            raise ValueError("sample_loop: ts_rows must be [n_steps, N] and noise_seq [n_steps, N, 4, h, w]")
        self._ensure_plan(B, H, W)
        if self._cond_key is None or not self.cache_text_emb:
            if full_emb is None or pooled_emb is None or image_emb is None:
                raise ValueError("full_emb, pooled_emb and image_emb are required")
            self.set_condition(full_emb, pooled_emb, image_emb)
            self._cond_key = True
            self.cache = {"cached": True}
        if not self.arch.inpainting and (inpaint_image is not None or inpaint_mask is not None):
            raise ValueError("inpaint_image / inpaint_mask given to a text2img UNet (create it with inpainting=True)")
        dev = x.device
        if self._chained(B):
            # two chains: the loop is driven from the host - per step the two half-batch graphs side by side, a join, the sampler kernels (one
            # captured graph with two branches runs them back to back: measured in round 4).  Same step arithmetic as k22_unet_sample_loop.
            L = _lib.lib()
            cur, nxt = x.detach().float().clone(), torch.empty(B, 4, H, W, dtype=torch.float32, device=dev)
            scratch = torch.empty(L.k22_sampler_scratch_bytes(B, H * W), dtype=torch.uint8, device=dev)
            tbl = table.float().contiguous()
            ii = None if init_img is None else init_img.float().expand(B, 4, H, W).contiguous()
            mm = None if init_img is None else img_mask.float().expand(B, 1, H, W).contiguous()
            for k in range(n_steps):
                half = cur[: B // 2]
                out = self.forward(torch.cat([half, half], 0), ts_rows[k], inpaint_image=inpaint_image, inpaint_mask=inpaint_mask)
                _lib.check(L.k22_sampler_step(cur.data_ptr(), out.data_ptr(), noise_seq[k].contiguous().data_ptr(), _lib.ptr(ii), _lib.ptr(mm), tbl.data_ptr(),
                                              int(table_rows[k]), float(guidance_scale), 1, float(clamp[0]), float(clamp[1]), int(pct[0]), float(pct[1]),
                                              scratch.data_ptr(), nxt.data_ptr(), None, B, H * W, _lib.current_stream()))
                cur, nxt = nxt, cur
            return cur
        key = (B, H, W, n_steps, tuple(table.shape), init_img is not None, str(dev))
        bufs = getattr(self, "_loop_bufs", None)
        if bufs is None or bufs["key"] != key:
            f32 = dict(dtype=torch.float32, device=dev)
            bufs = {"key": key, "x": torch.empty(B, 4, H, W, **f32), "tmp": torch.empty(B, 4, H, W, **f32), "ts": torch.empty(n_steps, B, **f32),
                    "noise": torch.empty(n_steps, B, 4, H, W, **f32), "table": torch.empty(tuple(table.shape), **f32),
                    "scratch": torch.empty(_lib.lib().k22_sampler_scratch_bytes(B, H * W), dtype=torch.uint8, device=dev),
                    "init": torch.empty(B, 4, H, W, **f32) if init_img is not None else None,
                    "mask": torch.empty(B, 1, H, W, **f32) if init_img is not None else None,
                    "img": torch.zeros(B, 4, H, W, **f32) if self.arch.inpainting else None,
                    "msk": torch.zeros(B, 1, H, W, **f32) if self.arch.inpainting else None}
            self._loop_bufs = bufs
        bufs["x"].copy_(x); bufs["ts"].copy_(ts_rows); bufs["noise"].copy_(noise_seq); bufs["table"].copy_(table)
        if init_img is not None:
            bufs["init"].copy_(init_img.float().expand(B, 4, H, W)); bufs["mask"].copy_(img_mask.float().expand(B, 1, H, W))
        if self.arch.inpainting:
            bufs["img"].zero_() if inpaint_image is None else bufs["img"].copy_(inpaint_image.float().expand(B, 4, H, W))
            bufs["msk"].zero_() if inpaint_mask is None else bufs["msk"].copy_(inpaint_mask.float().expand(B, 1, H, W))
        rows = (C.c_int * n_steps)(*[int(r) for r in table_rows])
        _lib.check(_lib.lib().k22_unet_sample_loop(
            self._handle, bufs["x"].data_ptr(), bufs["tmp"].data_ptr(), bufs["ts"].data_ptr(), bufs["noise"].data_ptr(), _lib.ptr(bufs["init"]),
            _lib.ptr(bufs["mask"]), _lib.ptr(bufs["img"]), _lib.ptr(bufs["msk"]), bufs["table"].data_ptr(), rows, n_steps, float(guidance_scale),
            float(clamp[0]), float(clamp[1]), int(pct[0]), float(pct[1]), bufs["scratch"].data_ptr(), 1 if self.use_graph else 0,
            _lib.current_stream()))
        return bufs["x"].clone()


def create_model(backend_dtype: torch.dtype = torch.bfloat16, use_graph: bool = True, inpainting: bool = False,
                 up: bool = False, **model_config) -> Text2ImUNetHIP:
    """Same keyword schema as the reference's create_model (kandinsky2/model/model_creation.py:9-83);
    version must be "2.1"."""
    if model_config.get("version", "2.1") != "2.1":
        raise NotImplementedError("only the 2.1 UNet is implemented")
    arch = make_arch(model_config, inpainting=inpainting)
    return Text2ImUNetHIP(arch, backend_dtype=backend_dtype, use_graph=use_graph,
                          cache_text_emb=model_config.get("cache_text_emb", True))
