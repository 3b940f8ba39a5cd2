"""Low-precision operand formats of the "high" precision forward, restated in numpy (TEST INFRASTRUCTURE ONLY).

The product contracts `f16(x) . f16(W)^T` on the 16-bit MFMA and adds the operands' low-order corrections as OCP e4m3
operands on the MX-scaled MFMA (efficientvideoclassification_youtube8m_amd/csrc/gemm_core_v3.h LOOP_FP8_TAIL).
What a kernel must reproduce is the OPERAND rounding - the accumulation is f32 either way - so the checker is:

  e4m3_round(a)          OCP e4m3fn (4 exponent bits, bias 7, 3 mantissa bits, no infinity, max 448, subnormal step 2^-9):
                         round to nearest even of a, clamped to +-448 first (the kernels clamp: codes above 448 are NaN in
                         this format) - the reference for v_cvt_pk_fp8_f32 and for torch.float8_e4m3fn
  f16_round(a)           IEEE binary16 round to nearest even
  corrected_product(..)  f16(x) . f16(W)^T + e4m3(x sx) . e4m3(W_lo sw_lo)^T / (sx sw_lo) + e4m3(x_lo sx_lo) . e4m3(W sw)^T / (sx_lo sw)
                         in float64: what evc_gemm_nt_f16_fp8 / evc_lstm_layer_fwd_f16_fp8lo / evc_dbof_cluster_pool_fwd_f16fp8
                         compute per contraction (the LSTM levels use the W_lo term for every operand and the x_lo term for
                         the input frames only)

The reference computes these contractions in float32 (tf.matmul / BasicLSTMCell: cs/frame_level_models.py:221-257,
cs/video_level_models.py:423-448): the corrected product is a cheaper way to the same number within ~2^-16 relative, not a
different algorithm."""
import numpy as np


def f16_round(a):
    return np.asarray(a, np.float64).astype(np.float32).astype(np.float16).astype(np.float64)


def e4m3_round(a):
    a = np.clip(np.asarray(a, np.float64), -448.0, 448.0)
    mag = np.abs(a)
    # exponent of the binade (normal numbers: 2^-6 .. 2^8), subnormals share the 2^-6 binade's step
    e = np.floor(np.log2(np.where(mag > 0, mag, 1.0)))
    e = np.clip(e, -6, 8)
    step = 2.0 ** (e - 3)                       # 3 mantissa bits
    q = np.round(mag / step)                    # numpy rounds half to even
    out = q * step
    return np.sign(a) * np.minimum(out, 448.0)


def corrected_product(x, w, x_hi_exp, x_lo_exp, w_lo_exp, w_hi_exp, correct_x=True):
    """x [M][K], w [N][K] float: the f16 product plus the e4m3 corrections (scales 2^exp, with x_hi + w_lo == x_lo + w_hi)."""
    assert x_hi_exp + w_lo_exp == x_lo_exp + w_hi_exp
    x, w = np.asarray(x, np.float64), np.asarray(w, np.float64)
    x16, w16 = f16_round(x), f16_round(w)
    z = x16 @ w16.T
    z = z + (e4m3_round(x * 2.0 ** x_hi_exp) @ e4m3_round((w - w16) * 2.0 ** w_lo_exp).T) * 2.0 ** -(x_hi_exp + w_lo_exp)
    if correct_x:
        z = z + (e4m3_round((x - x16) * 2.0 ** x_lo_exp) @ e4m3_round(w * 2.0 ** w_hi_exp).T) * 2.0 ** -(x_lo_exp + w_hi_exp)
    return z
