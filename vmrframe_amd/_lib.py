"""ctypes binding of libvmr_hip.so (the C ABI declared in include/vmr_hip.h).

The product path has NO fallback: if the shared library is missing or a call
fails, a RuntimeError is raised.  torch is imported first so the library binds
to the HIP runtime already loaded by PyTorch-ROCm (same SONAME), which makes
torch's streams and device pointers directly usable by the kernels.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL: loads libamdhip64)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvmr_hip.so")

F32, BF16, F16 = 0, 1, 2          # == VMR_F32 / VMR_BF16 / VMR_F16
LN_BWD_MAX_BLOCKS = 8192   # == VMR_LN_BWD_MAX_BLOCKS
MATCH_LOSS_SCRATCH = 512   # == VMR_MATCH_LOSS_SCRATCH


def cq_apply_parts_floats(B: int, Lc: int, Lq: int, D: int) -> int:
    """== VMR_CQ_APPLY_PARTS_FLOATS(B, Lc, Lq, D)"""
    return B * (D // (64 if max(Lc, Lq) > 128 else 128)) * 2 * ((Lc + 15) // 16 * 16) * ((Lq + 15) // 16 * 16)


def ln_bwd_ws_floats(rows: int, D: int) -> int:
    """== VMR_LN_BWD_WS_FLOATS(rows, D)"""
    return ((rows + 7) // 8) * 2 * (512 if D <= 512 else (1024 if D <= 1024 else 2048))

EPI_BIAS, EPI_RELU, EPI_DROPOUT, EPI_RESIDUAL, EPI_AUX, EPI_OUT_F32, EPI_ACCUM, EPI_ROWSCALE, EPI_SLAB, EPI_RES_PRE = \
    1, 2, 4, 8, 16, 32, 64, 128, 256, 512
EPI_AUX_BITS = 1024


class ColReduceItem(C.Structure):
    """== vmr_colreduce_item_t"""
    _fields_ = [("part", C.c_void_p), ("out0", C.c_void_p), ("out1", C.c_void_p), ("nblocks", C.c_int32),
                ("n0", C.c_int32), ("n1", C.c_int32), ("slots", C.c_int32)]


class TransposeItem(C.Structure):
    """== vmr_transpose_item_t"""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32)]


class GemmDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
                ("bias", C.c_void_p), ("residual", C.c_void_p), ("aux", C.c_void_p),
                ("rowscale", C.c_void_p),
                ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64), ("ldr", C.c_int64),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("transA", C.c_int32), ("transB", C.c_int32),
                ("dtype", C.c_int32), ("flags", C.c_int32), ("alpha", C.c_float),
                ("Z1", C.c_int32), ("Z2", C.c_int32),
                ("sA1", C.c_int64), ("sA2", C.c_int64), ("sB1", C.c_int64), ("sB2", C.c_int64),
                ("sC1", C.c_int64), ("sC2", C.c_int64),
                ("splitk", C.c_int32), ("drop_p", C.c_float), ("drop_seed", C.c_uint32),
                ("drop_row0", C.c_uint32), ("drop_step", C.c_void_p),
                ("bias2", C.c_void_p), ("bias_scale", C.c_float), ("res_div", C.c_int32),
                ("a_colsum", C.c_void_p)]


_lib = None

# name -> argtypes (restype is always int unless noted); mirrors include/vmr_hip.h
_P, _I, _L, _F, _U = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint32
SIGNATURES = {
    "vmr_gemm": [C.POINTER(GemmDesc), _P],
    "vmr_gemm2": [_P, _P, _P],
    "vmr_gemm2_reduce": [_P, _P, _P, _P, _I, _L, _I, _L, _P],
    "vmr_layernorm_fwd": [_P, _P, _P, _F, _P, _I, _P, _P, _P, _L, _I, _I, _F, _U, _P, _P],
    "vmr_layernorm_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _I, _F, _U, _P, _P],
    "vmr_ln_dwconv_fwd": [_P, _P, _P, _F, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vmr_dwconv_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vmr_layernorm_bwd_deferred": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _I, _F, _U, _P, _P, _P],
    "vmr_dwconv_bwd2_deferred": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P],
    "vmr_colreduce_batched": [_P, _I, _P],
    "vmr_transpose_batched": [_P, _I, _P],
    "vmr_ln_dwconv_fwd2": [_P, _P, _P, _F, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vmr_dwconv_bwd2": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vmr_lstm_seq_supported": [_I, _I, _I, _I],
    "vmr_lstm_seq_hist_bytes": [_I, _I, _I, _P],
    "vmr_lstm_seq_sentinel": [],
    "vmr_lstm_seq_bwd_sentinel": [],
    "vmr_lstm_seq_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vmr_lstm_seq_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vmr_convblock_bwd_supported": [_I, _I],
    "vmr_convblock_bwd_blocks": [_I, _I, _I, _I, _I],
    "vmr_convblock_bwd": [_P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P],
    "vmr_softmax_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _F, _U, _P, _P],
    "vmr_softmax_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _F, _U, _P, _P],
    "vmr_attention_fwd_supported": [_I, _I, _I],
    "vmr_attention_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _F, _U, _P, _P],
    "vmr_weighted_pool_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vmr_weighted_pool_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vmr_infer_basic": [_P, _P, _P, _P, _P, _I, _I, _P],
    "vmr_iou_metrics": [_P, _P, _P, _P, _I, _P],
    "vmr_attention_bwd_supported": [_I, _I, _I, _I],
    "vmr_attention_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _I, _I, _F, _U, _P, _P],
    "vmr_narrow_linear_fwd": [_P, _P, _P, _P, _L, _I, _I, _L, _I, _P],
    "vmr_narrow_linear_bwd": [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _L, _I, _P],
    "vmr_narrow_linear_bwd_add": [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _L, _I, _P],
    "vmr_gumbel_softmax_fwd": [_P, _P, _F, _U, _P, _P, _P, _L, _I, _I, _I, _P],
    "vmr_gumbel_softmax_bwd": [_P, _P, _P, _F, _P, _L, _I, _I, _I, _P],
    "vmr_match_loss_fwd": [_P, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    "vmr_match_loss_bwd": [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    "vmr_scale_shift_fwd": [_P, _P, _P, _P, _L, _I, _I, _P],
    "vmr_scale_shift_bwd": [_P, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    "vmr_char_cnn_ws_floats": [_I, _I, _P, _I],
    "vmr_char_cnn_fwd": [_P, _P, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _F, _U, _P, _P],
    "vmr_char_cnn_bwd": [_P, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _U, _P, _P],
    "vmr_resample_pad": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _L, _I, _P],
    "vmr_cq_score_supported": [_I, _I, _I, _I],
    "vmr_cq_score_split_supported": [_I, _I, _I, _I],
    "vmr_cq_score_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "vmr_cq_score_ws_floats": [_I],
    "vmr_debug_set_cq_split": [_I],
    "vmr_cq_score_fwd_ws": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "vmr_cq_apply_supported": [_I, _I, _I, _I],
    "vmr_cq_apply_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vmr_cq_apply_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vmr_cq_softmax_bwd_parts": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vmr_cq_score_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vmr_cq_softmax_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vmr_cq_softmax_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vmr_soft_ce_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _P],
    "vmr_soft_ce_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "vmr_cast": [_P, _I, _P, _I, _L, _I, _L, _L, _F, _U, _P, _P],
    "vmr_relu_bwd_bias": [_I, _P, _P, _P, _P, _L, _I, _L, _F, _I, _F, _U, _P, _P, _F, _P],
    "vmr_dropout_mask": [_P, _L, _F, _U, _P],
    "vmr_word_embedding_fwd": [_P, _P, _P, _P, _P, _L, _I, _L, _L, _I, _I, _I, _F, _U, _P, _P],
    "vmr_word_embedding_bwd": [_P, _P, _P, _L, _I, _L, _I, _F, _U, _P, _P],
    "vmr_embedding_fwd": [_P, _P, _P, _L, _I, _L, _P],
    "vmr_embedding_bwd": [_P, _P, _P, _L, _I, _L, _L, _P],
    "vmr_eltwise": [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    "vmr_splitk_reduce": [_P, _P, _I, _L, _I, _L, _P],
    "vmr_splitk_reduce_cast": [_P, _I, _L, _I, _P, _P, _P, _I, _P],
    "vmr_map2d_cells": [_P, _I, _I],
    "vmr_map2d_pool_fwd": [_P, _P, _P, _L, _P, _P, _I, _P, _P, _I, _I, _I, _I, _P],
    "vmr_map2d_pool_bwd": [_P, _P, _P, _P, _P, _I, _P, _P, _P, _L, _I, _I, _I, _I, _P],
    "vmr_map2d_scatter": [_P, _P, _P, _P, _I, _I, _I, _L, _I, _P],
    "vmr_sumsq": [_P, _P, _L, _P],
    "vmr_lstm_cell_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vmr_lstm_cell_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vmr_lstm_reverse_rows": [_P, _P, _P, _I, _I, _I, _I, _P],
    "vmr_lstm_step_supported": [_I, _I],
    "vmr_ban_sample_host": [_P, _P, _I, _I, _F, _I, _I, _I, _I, _P],
    "vmr_ban_sample": [_P, _P, _I, _I, _F, _I, _I, _I, _I, _P, _P, _P],
    "vmr_cos_rows_supported": [_I],
    "vmr_cos_rows_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vmr_cos_rows_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vmr_lstm_step_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vmr_lstm_step_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vmr_add_pos_fwd": [_P, _P, _P, _L, _I, _I, _I, _P],
    "vmr_label_fuse_fwd": [_P, _P, _P, _P, _P, _L, _I, _I, _I, _P],
    "vmr_label_fuse_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P],
    "vmr_add_pos_bwd": [_P, _P, _L, _I, _I, _I, _P],
    "vmr_debug_poison_lds": [_U, _P, _P],
    "vmr_debug_set_gemm_p8": [_I],
    "vmr_debug_set_gemm_dma": [_I],
    "vmr_gemm_aux_bits_supported": [_P],
    "vmr_adamw": [_P, _P, _P, _P, _P, _P, _I, _P, _F, _F, _F, _F, _F, _F, _I, _P, _F, _F, _P, _L, _P],
    "vmr_loss_scale_update": [_P, _P, _P, _I, _F, _P],
}


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). vmrframe_amd has no non-HIP fallback.")
        h = C.CDLL(LIB_PATH)
        h.vmr_version.restype = C.c_int
        h.vmr_last_error.restype = C.c_char_p
        for name, args in SIGNATURES.items():
            fn = getattr(h, name)  # AttributeError if the .so lacks a declared symbol
            fn.argtypes = args
            fn.restype = C.c_int
        _lib = h
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {lib().vmr_last_error().decode()}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float16:
        return F16
    raise TypeError(f"unsupported dtype {t.dtype}")


def is_16bit(dtype) -> bool:
    """the two 16-bit compute dtypes the kernels take (same layouts; v_mfma_f32_16x16x32_{bf16,f16})"""
    return dtype in (torch.bfloat16, torch.float16)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("vmrframe_amd ops run only on a HIP device (no CPU fallback); "
                               "got a tensor on %s" % t.device)
