"""ctypes binding of libk22hip.so (the C ABI declared in include/k22.h).

The product path has no CPU fallback: if the HIP library is missing or an entry point fails, these
helpers raise.  Build the library with `python __graft_entry__.py` / `make -C kandinsky-2_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("K22_LIB_PATH") or os.path.join(_HERE, "libk22hip.so")   # K22_LIB_PATH: developer knob (A/B of two builds)
# conv / GEMM tile configurations measured once on an MI355X for the shapes of BASELINE.json's configs (csrc/tuning.h):
# loaded when the library is opened so that those shapes run the same configurations - the same bits - on every box.
# K22_TILE_TABLE=<file> substitutes another table, K22_TILE_TABLE=0 starts with an empty one.
TILE_TABLE_PATH = os.path.join(_HERE, "tiles_gfx950.txt")

K22_BF16, K22_F32, K22_F16, K22_F16X3, K22_F16X2 = 0, 1, 2, 3, 4
# backend_dtype of the split-precision UNet engine (no torch dtype names it): fp32 tensors, MFMA operands as fp16 (hi, lo) pairs, three
# fp16 MFMAs per product - the arithmetic that meets the 1e-3 final-latent gate at 16-bit MFMA rate (include/k22.h: K22_F16X3)
F16X3 = "f16x3"
# backend_dtype of the ASYMMETRIC split engine (round 5; include/k22.h: K22_F16X2): the same tensors, arena and operand formats, with the
# precision of every MFMA op a property of the plan - weights always as (hi, lo) pairs, the activation operand at fp16 precision where
# the operand-rounding ablation says it is cheap (two MFMAs per product; attention one), all three MFMAs where it is not
F16X2 = "f16x2"
SPLIT_DTYPES = (F16X3, F16X2)
X3_WSCALE = 256.0   # csrc/common.h: K22_X3_WSCALE


def dtype_code(backend_dtype) -> int:
    """torch dtype of an engine's storage / MFMA operand type -> K22DType: bfloat16 (product path), float16 (the reference's own
    use_fp16 mode: same speed, 3 more mantissa bits), float32 (parity path, exact-fp32 MFMA)."""
    import torch
    try:
        return {torch.bfloat16: K22_BF16, torch.float32: K22_F32, torch.float16: K22_F16, F16X3: K22_F16X3, F16X2: K22_F16X2}[backend_dtype]
    except KeyError:
        raise ValueError('backend_dtype must be torch.bfloat16, torch.float16, torch.float32 "f16x3" or "f16x2"') from None
ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2
OUT_ROWMAJOR, OUT_ROWMAJOR_F32, OUT_NCHW_F32 = 0, 1, 2


class K22UNetConfig(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("in_channels", C.c_int), ("model_channels", C.c_int), ("out_channels", C.c_int),
        ("num_res_blocks", C.c_int), ("n_levels", C.c_int), ("channel_mult", C.c_int * 8),
        ("n_attention_ds", C.c_int), ("attention_ds", C.c_int * 8), ("num_head_channels", C.c_int),
        ("ctx_dim", C.c_int), ("ctx_len", C.c_int), ("n_image_embs", C.c_int), ("text_dim1", C.c_int),
        ("text_dim2", C.c_int), ("image_dim", C.c_int), ("head_type", C.c_int), ("hint_channels", C.c_int),
    ]


class K22MoVQConfig(C.Structure):
    _fields_ = [("dtype", C.c_int), ("ch", C.c_int), ("n_levels", C.c_int), ("ch_mult", C.c_int * 8),
                ("num_res_blocks", C.c_int), ("attn_levels", C.c_int), ("z_channels", C.c_int), ("out_ch", C.c_int)]


class K22PriorConfig(C.Structure):
    _fields_ = [("dtype", C.c_int), ("text_ctx", C.c_int), ("xf_width", C.c_int), ("xf_layers", C.c_int), ("xf_heads", C.c_int),
                ("xf_final_ln", C.c_int), ("clip_dim", C.c_int), ("clip_xf_width", C.c_int)]


class K22EncoderConfig(C.Structure):
    _fields_ = [("dtype", C.c_int), ("kind", C.c_int), ("width", C.c_int), ("layers", C.c_int), ("heads", C.c_int), ("n_ctx", C.c_int),
                ("vocab", C.c_int), ("out_dim", C.c_int), ("image_size", C.c_int), ("patch", C.c_int), ("max_pos", C.c_int),
                ("pad_id", C.c_int), ("ln_eps", C.c_float), ("mlp_dim", C.c_int), ("hidden_act", C.c_int)]


class K22Weight(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ptr", C.c_void_p)]


_P, _I, _L, _F, _D, _Z = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes); every symbol here is declared in include/k22.h
SIGNATURES = {
    "k22_version": (_I, []),
    "k22_build_flags": (_I, []),
    "k22_last_error": (C.c_char_p, []),
    "k22_set_option": (_I, [C.c_char_p, _I]),
    "k22_comm_broadcast_weights": (_I, [_P, _Z, _I, _P, _P]),
    "k22_tile_table_load": (_I, [C.c_char_p]),
    "k22_tile_table_save": (_I, [C.c_char_p]),
    "k22_tile_table_size": (_I, []),
    "k22_tile_table_measured": (_I, []),
    "k22_tile_table_clear": (None, []),
    "k22_unet_create": (_I, [C.POINTER(K22UNetConfig), C.POINTER(K22Weight), _I, C.POINTER(_P)]),
    "k22_unet_destroy": (None, [_P]),
    "k22_unet_plan": (_I, [_P, _I, _I, _I, C.POINTER(_Z)]),
    "k22_unet_bind": (_I, [_P, _P, _Z]),
    "k22_unet_set_condition": (_I, [_P, _P, _P, _P, _P]),
    "k22_unet_set_hint": (_I, [_P, _P, _P]),
    "k22_unet_forward": (_I, [_P, _P, _P, _P, _P, _P, _I, _P]),
    "k22_unet_sample_loop": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(_I), _I, _F, _F, _F, _I, _D, _P, _I, _P]),
    "k22_unet_num_ops": (_I, [_P]),
    "k22_unet_set_autotune": (_I, [_P, _I]),
    "k22_unet_tuning_report": (_I, [_P, C.c_char_p, _Z]),
    "k22_unet_profile": (_I, [_P, _I, C.POINTER(_D), C.POINTER(_D), C.POINTER(_D), C.POINTER(_I), _P]),
    "k22_afrag_bytes": (_Z, [_I, _I]),
    "k22_afrag_pack": (_I, [_P, _L, _P, _I, _I, _I, _P]),
    "k22_skinny_gemm": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "k22_finish_ln": (_I, [_P, _I, _P, _P, _L, _P, _P, _P, _I, _I, _F, _I, _P]),
    "k22_small_attention": (_I, [_P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P, _I, _I, _P]),
    "k22_prior_create": (_I, [C.POINTER(K22PriorConfig), C.POINTER(K22Weight), _I, C.POINTER(_P)]),
    "k22_prior_destroy": (None, [_P]),
    "k22_prior_plan": (_I, [_P, _I, C.POINTER(_Z)]),
    "k22_prior_bind": (_I, [_P, _P, _Z]),
    "k22_prior_tuning_report": (_I, [_P, C.c_char_p, _Z]),
    "k22_prior_forward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "k22_prior_sampler_step": (_I, [_P, _P, _P, _P, _P, _F, _P, _I, _I, _P]),
    "k22_encoder_create": (_I, [C.POINTER(K22EncoderConfig), C.POINTER(K22Weight), _I, C.POINTER(_P)]),
    "k22_encoder_destroy": (None, [_P]),
    "k22_encoder_plan": (_I, [_P, _I, C.POINTER(_Z)]),
    "k22_encoder_bind": (_I, [_P, _P, _Z]),
    "k22_encoder_forward": (_I, [_P, _P, _P, _P, _P, _P, _P]),
    "k22_movq_create": (_I, [C.POINTER(K22MoVQConfig), C.POINTER(K22Weight), _I, C.POINTER(_P)]),
    "k22_movq_destroy": (None, [_P]),
    "k22_movq_plan": (_I, [_P, _I, _I, _I, C.POINTER(_Z)]),
    "k22_movq_bind": (_I, [_P, _P, _Z]),
    "k22_movq_decode": (_I, [_P, _P, _P, _P, _P]),
    "k22_movq_plan_encoder": (_I, [_P, _I, _I, _I, C.POINTER(_Z)]),
    "k22_movq_encode": (_I, [_P, _P, _P, _P]),
    "k22_movq_num_ops": (_I, [_P]),
    "k22_ddim_step": (_I, [_P, _P, _P, _P, _F, _I, _P, _P, _I, _I, _P]),
    "k22_blend_noised": (_I, [_P, _P, _P, _P, _F, _F, _P, _I, _I, _I, _I, _P]),
    "k22_prepare_mask": (_I, [_P, _P, _I, _I, _I, _P]),
    "k22_plms_step": (_I, [_P, _P, _P, _P, _P, _I, _P, _F, _I, _P, _P, _P, _I, _I, _P]),
    "k22_sampler_scratch_bytes": (_Z, [_I, _I]),
    "k22_sampler_step": (_I, [_P, _P, _P, _P, _P, _P, _I, _F, _I, _F, _F, _I, _D, _P, _P, _P, _I, _I, _P]),
    "k22_gemm": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _L, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "k22_x3_pack": (_I, [_P, _P, _L, _F, _P]),
    "k22_conv3x3": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "k22_conv3x3_skip": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "k22_gemm_gnstats": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, C.POINTER(_I), _I, _P]),
    "k22_conv3x3_gnstats": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, C.POINTER(_I), _I, _P]),
    "k22_conv3x3_gn": (_I, [_P, _P, _I, _I, _P, _P, _P, _L, _F, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "k22_debug_conv_trace": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "k22_groupnorm": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _L, _F, _I, _I, _I, _P, _P, _I, _P]),
    "k22_groupnorm_scratch_bytes": (_Z, [_I, _I]),
    "k22_attention": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "k22_qkv_project": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "k22_qkv_project_stream": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "k22_linear_smallm": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "k22_debug_counter": (_L, [C.c_char_p]),
    "k22_stream_frag_bytes": (C.c_size_t, [_I, _I, _I, _I]),
    "k22_stream_repack": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "k22_debug_set_stream_frag": (_I, [_P, _P]),
    "k22_debug_set_stream_scratch": (_I, [_P, C.c_size_t]),
}

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is the only compute path (no CPU fallback). "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'`.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
        tbl = os.environ.get("K22_TILE_TABLE", TILE_TABLE_PATH)
        if tbl != "0" and os.path.exists(tbl):
            if l.k22_tile_table_load(tbl.encode()) < 0:
                raise RuntimeError(f"k22: cannot read the tile table {tbl}")
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().k22_last_error()
        raise RuntimeError(f"k22 error {rc}: {msg.decode() if msg else ''}")


def ptr(t) -> int:
    """data pointer of a (contiguous) torch tensor, or NULL for None."""
    if t is None:
        return None
    assert t.is_contiguous(), "k22: tensors crossing the C ABI must be contiguous"
    return t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
