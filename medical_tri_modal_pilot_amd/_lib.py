"""ctypes binding of libmtmp_hip.so (C ABI declared in include/mtmp.h).

The product path has NO fallback: if the shared library is missing or a symbol is
absent this module raises at import of the first op, and every non-zero status
from the library becomes a RuntimeError carrying mtmp_last_error().
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_longlong, c_uint, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MTMP_LIB", os.path.join(_HERE, "libmtmp_hip.so"))   # MTMP_LIB: diagnostic builds only

F32, BF16 = 0, 1

# name -> (restype, argtypes); kept in one table so tests can check it against include/mtmp.h
SIGNATURES = {
    "mtmp_abi_version": (c_int, []),
    "mtmp_last_error": (c_char_p, []),
    "mtmp_attn_fwd": (c_int, [c_int] + [c_void_p] * 9 + [c_int] * 5 + [c_float, c_void_p]),
    "mtmp_attn_cls_fwd": (c_int, [c_int] + [c_void_p] * 3 + [c_int, c_void_p, c_int] + [c_void_p] * 5 + [c_int] * 4 + [c_float, c_void_p]),
    "mtmp_attn_cls_bwd": (c_int, [c_int] + [c_void_p] * 3 + [c_int] + [c_void_p] * 8 + [c_int] * 5 + [c_float, c_void_p]),
    "mtmp_attn_fwd_grouped": (c_int, [c_int, c_int] + [c_void_p] * 13 + [c_int, c_int, c_float, c_void_p]),
    "mtmp_attn_bwd_grouped": (c_int, [c_int, c_int] + [c_void_p] * 17 + [c_int, c_int, c_float, c_void_p]),
    "mtmp_key_norms_floats": (c_longlong, [c_longlong, c_int]),
    "mtmp_key_norms": (c_int, [c_int, c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p]),
    "mtmp_attn_bwd": (c_int, [c_int] + [c_void_p] * 11 + [c_int] * 7 + [c_float, c_void_p]),
    "mtmp_ln_gemm": (c_int, [c_int] + [c_void_p] * 8 + [c_int] * 4 + [c_float, c_int, c_float, c_uint, c_void_p, c_void_p]),
    "mtmp_ln_gemm_qkv": (c_int, [c_int] + [c_void_p] * 9 + [c_int, c_int, c_float, c_void_p]),
    "mtmp_gemm_nt": (c_int, [c_int] + [c_void_p] * 5 + [c_int] * 7 + [c_float, c_uint, c_void_p, c_void_p, c_float, c_void_p, c_int,
                             c_void_p]),
    "mtmp_gemm_nt_live": (c_int, [c_int] + [c_void_p] * 5 + [c_int] * 7 + [c_float, c_uint, c_void_p, c_void_p, c_float, c_void_p, c_int,
                             c_void_p, c_void_p]),
    "mtmp_sign_bits_bytes": (c_longlong, [c_int, c_int]),
    "mtmp_ln_gemm_signs": (c_int, [c_int] + [c_void_p] * 8 + [c_int] * 4 + [c_float, c_float, c_uint, c_void_p, c_void_p, c_void_p]),
    "mtmp_gemm_nt_signs": (c_int, [c_int] + [c_void_p] * 3 + [c_int] * 4 + [c_void_p, c_float, c_void_p]),
    "mtmp_gemm_nt_signs_drop": (c_int, [c_int] + [c_void_p] * 3 + [c_int] * 4 + [c_void_p, c_float, c_float, c_uint, c_void_p,
                                                                                   c_void_p, c_void_p]),
    "mtmp_ln_gemm_qkv_grouped": (c_int, [c_int, c_int] + [c_void_p] * 11 + [c_float, c_void_p, c_void_p]),
    "mtmp_ln_gemm_signs_grouped": (c_int, [c_int, c_int] + [c_void_p] * 10 + [c_int, c_void_p, c_float, c_float, c_void_p, c_void_p,
                                                                                   c_void_p, c_void_p]),
    "mtmp_gemm_nt_grouped": (c_int, [c_int, c_int] + [c_void_p] * 6 + [c_int, c_int] + [c_void_p] * 3 + [c_int, c_float, c_void_p,
                                                                                                       c_void_p, c_void_p, c_void_p]),
    "mtmp_gemm_nt_signs_drop_grouped": (c_int, [c_int, c_int] + [c_void_p] * 4 + [c_int, c_void_p, c_void_p, c_float, c_float,
                                                                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mtmp_gemm_lnbwd_grouped": (c_int, [c_int, c_int] + [c_void_p] * 11 + [c_int, c_void_p, c_float, c_void_p, c_void_p]),
    "mtmp_gemm_tn_group_plan": (c_int, [c_int, c_void_p, c_int, c_int, c_void_p]),
    "mtmp_gemm_tn_grouped": (c_int, [c_int, c_int] + [c_void_p] * 4 + [c_int, c_int] + [c_void_p] * 4 + [c_void_p]),
    "mtmp_gemm_tn_slab_rows": (c_int, [c_int] * 4),
    "mtmp_gemm_lnbwd_slab_rows": (c_int, [c_int]),
    "mtmp_reduce_batch": (c_int, [c_void_p] * 6 + [c_int, c_void_p]),
    "mtmp_reduce_scatter": (c_int, [c_void_p] * 5 + [c_int, c_void_p]),
    "mtmp_stream_input_slab_rows": (c_int, [c_int]),
    "mtmp_stream_input_bwd_partials": (c_int, [c_int] + [c_void_p] * 7 + [c_int] * 3 + [c_float, c_uint, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mtmp_stream_input_bwd_grouped": (c_int, [c_int, c_int] + [c_void_p] * 18),
    "mtmp_stream_input_fwd_add": (c_int, [c_int] + [c_void_p] * 8 + [c_int] * 3 + [c_float, c_float, c_uint, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "mtmp_token_sums": (c_int, [c_int, c_int] + [c_void_p] * 5),
    "mtmp_tie_bwd_slab_rows": (c_int, [c_int]),
    "mtmp_tie_time_embed_bwd_partials": (c_int, [c_int, c_void_p, c_int, c_void_p, c_int, c_int] + [c_void_p] * 6),
    "mtmp_publish_scalar": (c_int, [c_void_p, c_void_p, c_void_p]),
    "mtmp_copy_batch": (c_int, [c_void_p] * 4 + [c_int, c_void_p]),
    "mtmp_stream_lengths": (c_int, [c_void_p] * 4 + [c_int] * 3 + [c_void_p]),
    "mtmp_row_starts": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mtmp_image_slots": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "mtmp_transpose_batch": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "mtmp_layernorm_rows": (c_int, [c_int] + [c_void_p] * 4 + [c_longlong, c_int, c_float, c_int, c_int, c_int, c_void_p]),
    "mtmp_layernorm_rows_live": (c_int, [c_int] + [c_void_p] * 4 + [c_longlong, c_int, c_float, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mtmp_swin_window_attn": (c_int, [c_int] + [c_void_p] * 3 + [c_int] * 6 + [c_float, c_void_p]),
    "mtmp_swin_window_attn_live": (c_int, [c_int] + [c_void_p] * 3 + [c_int] * 6 + [c_float, c_void_p, c_void_p]),
    "mtmp_layernorm_rows_bwd_slab_rows": (c_int, [c_longlong, c_int]),
    "mtmp_layernorm_rows_bwd": (c_int, [c_int] + [c_void_p] * 5 + [c_longlong, c_int, c_float, c_void_p]),
    "mtmp_gelu_fwd": (c_int, [c_int, c_void_p, c_void_p, c_longlong, c_void_p]),
    "mtmp_gelu_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_longlong, c_void_p]),
    "mtmp_swin_window_attn_bwd": (c_int, [c_int] + [c_void_p] * 5 + [c_int] * 6 + [c_float, c_void_p]),
    "mtmp_gemm_tn_ws_floats": (c_longlong, [c_int, c_int, c_int]),
    "mtmp_gemm_tn": (c_int, [c_int] + [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "mtmp_gemm_tn_live": (c_int, [c_int] + [c_void_p] * 5 + [c_int] * 5 + [c_void_p, c_void_p]),
    "mtmp_ln_bwd_ws_floats": (c_int, [c_int]),
    "mtmp_ln_bwd": (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                            c_void_p, c_void_p, c_int, c_float, c_void_p]),
    "mtmp_gemm_lnbwd_ws_floats": (c_int, [c_int]),
    "mtmp_gemm_lnbwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "mtmp_tie_embed_fwd": (c_int, [c_int] + [c_void_p] * 4 + [c_int, c_void_p]),
    "mtmp_tie_bwd_ws_floats": (c_int, [c_int]),
    "mtmp_tie_embed_bwd": (c_int, [c_int] + [c_void_p] * 5 + [c_int, c_void_p]),
    "mtmp_time_embed_fwd": (c_int, [c_int] + [c_void_p] * 4 + [c_int, c_void_p]),
    "mtmp_time_embed_bwd": (c_int, [c_int] + [c_void_p] * 5 + [c_int, c_void_p]),
    "mtmp_tie_embed_packed_fwd": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mtmp_tie_embed_packed_bwd": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mtmp_stream_input_ws_floats": (c_int, [c_int]),
    "mtmp_stream_input_fwd": (c_int, [c_int] + [c_void_p] * 8 + [c_int] * 3 + [c_float, c_float, c_uint, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mtmp_stream_input_bwd": (c_int, [c_int] + [c_void_p] * 8 + [c_int] * 3 + [c_float, c_uint, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mtmp_head_ws_floats": (c_int, [c_int]),
    "mtmp_head_fwd": (c_int, [c_void_p] * 6 + [c_int, c_float, c_float, c_float, c_int, c_void_p]),
    "mtmp_head_bwd": (c_int, [c_void_p] * 12 + [c_int, c_float, c_int, c_void_p]),
    "mtmp_head_fwd_t": (c_int, [c_int] + [c_void_p] * 6 + [c_int, c_float, c_float, c_float, c_int, c_void_p, c_void_p]),
    "mtmp_head_bwd_scatter": (c_int, [c_int] + [c_void_p] * 9 + [c_int, c_float, c_int, c_void_p]),
    "mtmp_bce_logits_mean": (c_int, [c_void_p] * 4 + [c_int, c_void_p]),
    "mtmp_timestamp": (c_int, [c_void_p, c_void_p]),
    "mtmp_swin_ln_linear": (c_int, [c_int] + [c_void_p] * 6 + [c_longlong, c_int, c_int, c_float, c_void_p]),
    "mtmp_swin_ln_linear_live": (c_int, [c_int] + [c_void_p] * 6 + [c_longlong, c_int, c_int, c_float, c_void_p, c_void_p]),
    "mtmp_swin_mlp": (c_int, [c_int] + [c_void_p] * 8 + [c_int, c_void_p, c_longlong, c_int, c_float, c_void_p]),
    "mtmp_swin_mlp_live": (c_int, [c_int] + [c_void_p] * 8 + [c_int, c_void_p, c_longlong, c_int, c_float, c_void_p, c_void_p]),
    "mtmp_swin_attn_block": (c_int, [c_int] + [c_void_p] * 3 + [c_float] + [c_void_p] * 7 + [c_int] * 6 + [c_float, c_void_p, c_void_p]),
    "mtmp_swin_stem_fwd": (c_int, [c_int] + [c_void_p] * 6 + [c_int] * 3 + [c_void_p]),
    "mtmp_swin_stem_fwd_live": (c_int, [c_int] + [c_void_p] * 6 + [c_int] * 3 + [c_void_p, c_void_p, c_void_p]),
    "mtmp_adamw_step": (c_int, [c_void_p] * 5 + [c_longlong] + [c_float] * 5 + [c_int, c_float, c_void_p]),
    "mtmp_bottleneck_exchange_fwd": (c_int, [c_int] + [c_void_p] * 3 + [c_int] * 4 + [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mtmp_bottleneck_exchange_bwd": (c_int, [c_int] + [c_void_p] * 3 + [c_int] * 4 + [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mtmp_dropout_bwd": (c_int, [c_int, c_void_p, c_void_p, c_longlong, c_uint, c_void_p, c_float, c_void_p]),
}

_lib = None


def lib():
    """The loaded library (loads on first use; raises if it was not built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or `make -C medical_tri_modal_pilot_amd/csrc`). There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().mtmp_last_error()
        raise RuntimeError(f"libmtmp_hip {what} failed (status {rc}): {msg.decode() if msg else '?'}")


def call(name: str, *args):
    check(getattr(lib(), name)(*args), name)
