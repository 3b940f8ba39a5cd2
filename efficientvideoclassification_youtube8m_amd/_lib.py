"""ctypes binding of libevc_hip.so (the C ABI declared in include/evc.h).

The library is built in-tree by ``csrc/build.sh`` (see ``__graft_entry__.build``).
Loading fails loudly: there is no CPU or PyTorch fallback for any kernel.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EVC_LIB", os.path.join(_HERE, "libevc_hip.so"))   # EVC_LIB: debug builds only

vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> argtypes (every function returns int except the two noted below)
SIGNATURES = {
    "evc_check_device": [i32],
    "evc_l2norm_chunk_fwd": [vp, vp, vp, i32, i32, i32, i32, vp, i32, i32, vp, i32, vp, vp, i32, vp, i32, vp, i32, vp],
    "evc_sort_rows_by_len": [vp, i32, i32, vp, vp, vp, vp],
    "evc_frame_counts": [vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp],
    "evc_gemm_nt": [vp, i64, vp, i64, vp, i64, i32, i32, i32, vp, i32, i32, vp],
    "evc_gemm_nt_sqnorm": [vp, i64, vp, i64, vp, i64, i32, i32, i32, vp, f32, vp, vp, i64, vp],
    "evc_gemm_tn": [vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, i32, vp],
    "evc_gemm_tn2": [vp, i64, vp, i64, i32, vp, i64, i32, i32, vp, i64, i32, i32, i32, i32, vp],
    "evc_gemm_tn2_rows": [vp, i64, vp, i64, i32, vp, i64, i32, i32, vp, i64, i32, i32, i32, vp, i32, i32, vp],
    "evc_gemm_tn2_slabs": [vp, i64, vp, i64, i32, vp, i64, i32, i32, vp, i64, i64, i32, i32, i32, i32, vp],
    "evc_sum_slabs": [vp, i64, i32, i32, i32, i64, vp, i64, i32, vp],
    "evc_colsum_bf16": [vp, i64, i32, i32, i32, vp, vp],
    "evc_colsum_bf16_det": [vp, i64, i32, i32, i32, vp, vp, i32, vp],
    "evc_lstm_layer_fwd": [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp],
    "evc_lstm_layer_fwd_hp": [vp, vp, i64, vp, i64, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, i64, vp, vp, vp],
    "evc_gemm_nt_split": [vp, i64, vp, i64, vp, i64, i32, i32, i32, vp, vp],
    "evc_cast_f32_to_f16_wide": [vp, i64, i32, i32, i32, i32, i32, vp, vp],
    "evc_cast_f32_to_bf16_wide": [vp, i64, i32, i32, vp, i64, i32, vp],
    "evc_lstm_layer_fwd_f16": [vp, i64, vp, vp, vp, i32, i32, i32, i32, vp, i32, vp, vp, vp, i64, vp, vp, vp, vp, vp],
    "evc_lstm_layer_fwd_f16_fp8lo": [vp, i64, i32, i64, i32, vp, vp, i32, i32, vp, vp, i32, i32, i32, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, i32, vp],
    "evc_l2norm_chunk_int": [vp, vp, i32, i32, i32, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp, i32, vp],
    "evc_lstm_level2_fwd_high": [vp, i64, i32, i64, i32, vp, vp, i32, i32, vp, vp, vp, i32, vp, i64, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i64,
                                 vp, vp, vp, vp, vp, vp, vp],
    "evc_lstm_layer_fwd_f16_dith": [vp, i64, i32, i64, i32, vp, i64, vp, i64, i32, vp, vp, i32, i32, i32, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp],
    "evc_cast_f32_to_f16_dither": [vp, i64, i32, i64, C.c_uint32, vp, i64, i64, vp],
    "evc_lstm_stack2_fwd_f16_fp8lo": [vp, i32, vp, vp, vp, vp, vp, i32, i32, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp],
    "evc_cast_f32_to_fp8_lo": [vp, i64, i32, i32, i32, i32, i32, vp, i64, vp],
    "evc_cast_f32_to_fp8_lohi": [vp, i64, i32, i32, i32, i32, i32, vp, i64, vp],
    "evc_lstm_layer_bwd": [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "evc_transpose_to_bf16": [vp, i32, i64, i32, i32, vp, i64, i32, i32, vp],
    "evc_cast_f32_to_bf16": [vp, i64, i32, i32, vp, i64, vp],
    "evc_cast_f32_to_f16": [vp, i64, i32, i32, vp, i64, vp],
    "evc_cast_f32_to_bf16_split": [vp, i64, i32, i32, vp, vp, i64, vp],
    "evc_rowsum_bf16": [vp, i64, i32, i32, vp, vp],
    "evc_lstm_stack2_fwd_f16": [vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp],
    "evc_cast_f32_to_f16_segs": [vp, i64, i32, i32, i32, vp, vp],
    "evc_cast_f32_to_f16_wlo": [vp, i64, i32, i32, i32, vp, vp],
    "evc_lstm_level2_fwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp],
    "evc_lstm_stack2_fwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp],
    "evc_moe_grad_update": [vp, i64, vp, i64, i32, i32, i32, vp, vp, vp, vp, vp, i64, f32, vp, vp, f32, f32, f32, f32, f32, vp],
    "evc_moe_grad_update_wide": [vp, i64, vp, i64, i32, i32, i32, vp, vp, vp, vp, vp, i64, vp, vp, vp, i32, i32, f32, vp, vp, f32, f32, f32, f32, f32, vp],
    "evc_gram_slabs": [vp, i64, i32, i32, i32, vp, vp],
    "evc_moe_grad_norms": [vp, i32, vp, i32, i32, vp, i64, vp, i64, vp, i32, i32, f32, vp, vp, vp, vp],
    "evc_moe_grad_update_apply": [vp, i64, vp, i64, i32, i32, i32, vp, vp, vp, vp, vp, i64, vp, vp, vp, i32, i32, f32, vp, vp, f32, f32, f32, f32, f32, vp, vp],
    "evc_gemm_nt_f16_fp8": [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, i32, vp, vp],
    "evc_cast_f32_to_f16_fp8x": [vp, i64, i32, i32, i32, i32, vp, vp],
    "evc_absmax_partials": [vp, i64, i32, i32, vp, vp],
    "evc_cast_f32_to_f16_fp8x_dyn": [vp, i64, i32, i32, i32, i32, vp, vp, vp],
    "evc_gemm_nt_f16_fp8_dyn": [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, i32, vp, i32, vp, vp],
    "evc_moe_grad_update_phase": [vp, i64, vp, i64, i32, i32, i32, vp, vp, vp, vp, vp, i64, f32, vp, vp, f32, f32, f32, f32, f32, i32, vp],
    "evc_moe_tail_fwd": [vp, vp, i32, i32, i32, vp, vp, vp],
    "evc_moe_tail_bwd": [vp, vp, vp, i32, i32, i32, vp, i64, vp, i64, vp],
    "evc_ce_loss": [vp, vp, i32, i32, f32, vp, vp, i32, vp],
    "evc_ce_loss_ordered": [vp, vp, i32, i32, f32, vp, vp, i32, vp, vp],
    "evc_rep_loss_ordered": [vp, vp, i32, i32, f32, vp, vp, i32, vp, vp],
    "evc_kl_pred_loss": [vp, vp, vp, vp, i32, i32, f32, vp, vp, i32, vp],
    "evc_rep_loss": [vp, vp, i32, i32, f32, vp, vp, i32, vp],
    "evc_grad_sqnorm": [vp, vp, f32, i64, vp, vp],
    "evc_clip_adam_step": [vp, vp, vp, vp, i64, f32, vp, f32, f32, f32, f32, f32, vp, vp],
    "evc_clip_adam_small": [i32, vp, vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, vp],
    "evc_sqnorm2_partials": [vp, i64, vp, i64, vp, vp],
    "evc_lstm_adam_fused": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, f32, f32, f32, f32, f32, vp, vp, i64, vp, i64, i32, i32,
                            vp, i64, i32, i32, i32, i32, i32, vp],
    "evc_adam2d_fused": [vp, vp, vp, vp, i32, i32, vp, vp, f32, f32, f32, f32, f32, vp, vp, i64, vp, i64, vp, i64, i32, i32, i32, vp],
    "evc_meanpool_fwd": [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp],
    "evc_sigmoid_fwd": [vp, i64, vp],
    "evc_sigmoid_bwd": [vp, vp, i64, vp, vp],
    "evc_sample_frames_gather": [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp],
    "evc_sample_sequence_gather": [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp],
    "evc_relu6_fwd": [vp, i64, vp, vp, vp],
    "evc_relu6_bwd": [vp, vp, i64, vp, vp, vp],
    "evc_framepool_mean_fwd": [vp, i32, i32, i32, vp, vp, vp],
    "evc_framepool_mean_bwd": [vp, i32, i32, i32, vp, vp],
    "evc_bn_stats": [vp, i32, i32, vp, vp, vp, vp],
    "evc_bn_apply": [vp, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp],
    "evc_bn_relu6_bwd": [vp, vp, i32, i32, vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp],
    "evc_bn_stats_partial": [vp, i32, i32, vp, vp],
    "evc_bn_stats_finalize": [vp, i32, i32, vp, vp, vp],
    "evc_ema_update": [vp, vp, f32, i32, vp],
    "evc_bn_bwd_partial": [vp, vp, i32, i32, vp, vp, vp, vp, i32, vp, i32, vp, vp],
    "evc_bn_bwd_finalize": [vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp],
    "evc_bn_relu6_framepool_fwd": [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp],
    "evc_framepool_max_fwd": [vp, i32, i32, i32, vp, vp, vp, vp],
    "evc_framepool_max_bwd": [vp, vp, i32, i32, i32, vp, vp],
    "evc_fill_f32": [vp, i64, f32, vp],
    "evc_debug_occupy": [i32, i32, i32, C.c_double, vp],
    "evc_stream_create_cu_mask": [vp, i32, vp],
    "evc_stream_destroy": [vp],
    "evc_lstm_stack2_bwd": [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "evc_netvlad_softmax_fwd": [vp, i32, i32, vp, vp, vp, vp, vp, vp],
    "evc_netvlad_softmax_bwd": [vp, vp, i32, i32, vp, vp],
    "evc_netvlad_aggregate_fwd": [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp],
    "evc_netvlad_aggregate_bwd": [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "evc_netvlad_dcenters": [vp, vp, i32, i32, i32, vp, vp],
    "evc_netvlad_normalize_fwd": [vp, i32, i32, i32, vp, vp, vp, vp, vp],
    "evc_netvlad_normalize_bwd": [vp, vp, vp, vp, i32, i32, i32, vp, vp],
    "evc_dbof_workspace": [i32, i32, vp, vp, vp],
    "evc_dbof_gather": [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp],
    "evc_bn_partials_reduce": [vp, i32, i32, vp, vp],
    "evc_bn_finalize_ema": [vp, i32, i32, vp, vp, vp, vp, f32, vp],
    "evc_dbof_input_bn_apply": [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp],
    "evc_dbof_cluster_pool_fwd": [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp],
    "evc_dbof_input_bn_apply_f16fp8": [vp, i32, i32, i32, vp, vp, vp, vp, vp, i32, i32, vp, vp],
    "evc_dbof_cluster_pool_fwd_f16fp8": [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp],
    "evc_dbof_pool_finish": [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp],
    "evc_dbof_dact": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp],
    "evc_gemm_tn_slabs": [vp, i64, vp, i64, vp, i32, i32, i32, i32, vp],
    "evc_dbof_wgrad_finish": [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp],
}
EXPORTS = tuple(SIGNATURES) + ("evc_version", "evc_last_error")

_lib = None


class EvcError(RuntimeError):
    pass


def load():
    """Load libevc_hip.so; raise (never fall back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EvcError(
            "libevc_hip.so not found at %s - build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or csrc/build.sh. "
            "There is no CPU fallback for the HIP kernels." % LIB_PATH)
    # PyTorch-ROCm bundles its own HIP runtime; it must be the first one loaded into the process
    # (loading this library first pulls in /opt/rocm's copy and torch then sees "No HIP GPUs").
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.evc_version.argtypes = []
    lib.evc_version.restype = C.c_int
    lib.evc_last_error.argtypes = []
    lib.evc_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def call(name, *args):
    """Invoke an entry point; raise EvcError with evc_last_error() on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise EvcError("%s failed (%d): %s" % (name, rc, lib.evc_last_error().decode()))
