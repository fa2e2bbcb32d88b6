"""ctypes binding of libdclip_hip.so (the C ABI in include/dclip_hip.h).

The library is built in-tree by `make` / `__graft_entry__.build()`; there is NO fallback: if
it is missing or a symbol is absent, importing the product path fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DCLIP_LIB_PATH: another BUILD of this library (same-box A/B runs of bench.py, tools/ab/*.so); never a fallback
LIB_PATH = os.environ.get("DCLIP_LIB_PATH") or os.path.join(_HERE, "libdclip_hip.so")

P = C.c_void_p          # device pointers and the stream
I = C.c_int
F = C.c_float
Z = C.c_size_t
L = C.c_int64

# name -> (restype, argtypes); mirrors include/dclip_hip.h declaration by declaration
SIGNATURES = {
    "dclip_abi_version": (I, []),
    "dclip_last_error": (C.c_char_p, []),
    "dclip_gemm_f32_workspace": (Z, [I, I, I, I, I]),
    "dclip_gemm_f32": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, I, F, I, P, Z, P]),
    "dclip_colsum_f32_workspace": (Z, [I, I]),
    "dclip_colsum_f32": (I, [P, P, I, I, I, I, P, Z, P]),
    "dclip_layernorm_fwd": (I, [P, P, P, P, P, P, I, I, F, P]),
    "dclip_layernorm_bwd_workspace": (Z, [I, I]),
    "dclip_layernorm_bwd": (I, [P, P, P, P, P, P, P, P, P, I, I, I, P, Z, P]),
    "dclip_layernorm_bwd_ex": (I, [P, P, P, P, P, P, P, P, P, P, P, I, I, I, P, Z, P]),
    "dclip_attention_fwd": (I, [P, P, P, I, I, I, I, P]),
    "dclip_attention_bwd": (I, [P, P, P, P, P, P, I, I, I, I, P]),
    "dclip_attention_bwd_workspace": (Z, [I, I, I, I]),
    "dclip_attention_bwd_ws": (I, [P, P, P, P, P, P, Z, I, I, I, I, P]),
    "dclip_attention_cls_fwd": (I, [P, P, P, I, I, I, P]),
    "dclip_attention_cls_bwd": (I, [P, P, P, P, P, P, I, I, I, P]),
    "dclip_attention_row_fwd": (I, [P, P, P, P, I, I, I, P]),
    "dclip_attention_row_fwd_bf16": (I, [P, P, P, I, I, I, P]),
    "dclip_im2col_bf16": (I, [P, P, I, I, I, I, I, I, P]),
    "dclip_im2col": (I, [P, P, I, I, I, I, I, P]),
    "dclip_vision_assemble_fwd": (I, [P, P, P, P, I, I, I, P]),
    "dclip_vision_assemble_bwd": (I, [P, P, I, I, I, P]),
    "dclip_text_embed_fwd": (I, [P, P, P, P, I, I, I, I, P]),
    "dclip_text_embed_bwd": (I, [P, P, P, I, I, I, I, P]),
    "dclip_first_eos": (I, [P, P, I, I, L, P]),
    "dclip_gather_rows": (I, [P, P, P, I, I, I, P]),
    "dclip_scatter_rows": (I, [P, P, P, I, I, I, P]),
    "dclip_normalize_rows_fwd": (I, [P, P, P, I, I, F, P]),
    "dclip_normalize_rows_bwd": (I, [P, P, P, P, I, I, F, I, P]),
    "dclip_contrastive_workspace": (Z, [I, I, I]),
    "dclip_contrastive_lse": (I, [P, P, P, P, I, I, I, I, F, P, Z, P]),
    "dclip_contrastive_grad": (I, [P, P, P, P, P, I, I, I, I, F, F, P, Z, P]),
    "dclip_cosine_loss_fwd": (I, [P, P, P, P, I, I, P]),
    "dclip_cosine_loss_bwd": (I, [P, P, P, P, I, I, F, I, P]),
    "dclip_sub_reduce": (I, [P, P, P, I, F, I, P]),
    "dclip_cross_attention_fwd": (I, [P, P, P, P, I, I, I, I, P]),
    "dclip_cross_attention_bwd": (I, [P, P, P, P, P, P, P, P, I, I, I, I, P]),
    "dclip_aggregation_fwd": (I, [P, P, P, I, I, I, F, F, I, P]),
    "dclip_aggregation_bwd": (I, [P, P, P, P, I, I, I, F, F, P]),
    "dclip_pack_tokens": (I, [P, P, P, P, I, I, I, I, P]),
    "dclip_sanitize_groups": (I, [P, P, I, I, I, I, P]),
    "dclip_mask_rows": (I, [P, P, I, I, I, P]),
    "dclip_rank_count_workspace": (Z, [I, I]),
    "dclip_rowdot_gather": (I, [P, P, P, P, I, I, I, P]),
    "dclip_rank_count": (I, [P, P, P, P, P, I, I, I, P, Z, P]),
    "dclip_crop_resize_workspace": (Z, [I, I, I, I]),
    "dclip_crop_resize_u8": (I, [P, P, P, P, I, I, I, I, I, I, I, P, Z, P]),
    "dclip_clip_preprocess_workspace": (Z, [I, I, I, I]),
    "dclip_clip_preprocess_u8": (I, [P, P, P, I, I, I, I, P, P, P, Z, P]),
    "dclip_gemm_bf16": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "dclip_attention_fwd_bf16": (I, [P, P, I, I, I, I, P]),
    "dclip_attention_fwd_bf16_lse": (I, [P, P, P, I, I, I, I, P]),
    "dclip_attention_bwd_bf16": (I, [P, P, P, P, P, I, I, I, I, P]),
    "dclip_cast_f32_bf16": (I, [P, P, I, I, I, I, P]),
    "dclip_layernorm_fwd_bf16": (I, [P, P, P, P, I, I, F, P]),
    "dclip_gemm_bf16_ex": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "dclip_layernorm_fwd_bf16_stats": (I, [P, P, P, P, P, P, I, I, F, P]),
    "dclip_transpose_to_bf16": (I, [P, I, P, P, I, I, I, I, I, P]),
    "dclip_attention_fwd_io16": (I, [P, P, P, I, I, I, I, P]),
    "dclip_attention_bwd_io16": (I, [P, P, P, P, P, I, I, I, I, P]),
    "dclip_mt_weights_record_bytes": (I, []),
    "dclip_mt_weights_bf16": (I, [P, I, I, P]),
    "dclip_rowsum_bf16": (I, [P, P, I, I, I, P]),
    "dclip_gemm_bf16_splitk_plan": (I, [I, I, I]),
    "dclip_gemm_bf16_splitk_workspace": (Z, [I, I, I]),
    "dclip_gemm_bf16_splitk": (I, [P, P, P, I, I, I, I, I, I, I, P, Z, P]),
    "dclip_gemm_bf16_wgrad_tokmajor_plan": (I, [I, I, I]),
    "dclip_gemm_bf16_wgrad_tokmajor": (I, [P, P, P, I, I, I, I, I, I, I, P, Z, P]),
    "dclip_colsum_bf16": (I, [P, P, I, I, I, I, P, Z, P]),
    "dclip_sumsq_blocks": (I, [Z]),
    "dclip_sumsq_f32": (I, [P, Z, P, P]),
    "dclip_clip_coef": (I, [P, I, F, P, P, P]),
    "dclip_adamw_f32": (I, [P, P, P, P, Z, F, F, F, F, F, I, P, P]),
    "dclip_mt_record_bytes": (I, []),
    "dclip_mt_chunk_elems": (I, []),
    "dclip_mt_sumsq_f32": (I, [P, I, I, P, P]),
    "dclip_mt_adamw_f32": (I, [P, I, I, F, F, F, F, F, P, P]),
    "dclip_mt_adam_f32": (I, [P, I, I, F, F, F, F, F, P, P]),
    "dclip_axpby": (I, [P, P, F, F, Z, P]),
    "dclip_fill": (I, [P, F, Z, P]),
}

_lib = None


class DclipError(RuntimeError):
    pass


def load():
    """Load the shared library and bind every declared entry point (raises if any is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DclipError(
            f"{LIB_PATH} not found: build it with `make` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "dclip_amd has no CPU or PyTorch fallback for its kernels.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.dclip_abi_version() != 1:
        raise DclipError(f"ABI version mismatch: library reports {lib.dclip_abi_version()}")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().dclip_last_error().decode(errors="replace")
        raise DclipError(f"{what or 'dclip call'} failed (rc={rc}): {msg}")
