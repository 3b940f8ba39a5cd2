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
different algorithm.

Round 5 adds the OCP FP6 format e2m3 (the same MFMA opcode takes it at twice the e4m3 rate; scripts/probes/fp6_mfma_probe.hip):

  e2m3_round(a)          sign, 2 exponent bits (bias 1), 3 mantissa bits, no infinity / NaN: +-{0, 0.125 .. 0.875 (subnormal), 1 .. 1.875,
                         2 .. 3.75, 4 .. 7.5}; round to nearest even, saturating at +-7.5
  e2m3_encode / decode   the 6-bit codes (pinned on the full 64-entry table in tests/test_oracle_lowprec.py)
  pack_fp6 / unpack_fp6  32 codes -> 24 bytes, element i at bits 6i .. 6i+5 (little-endian): what one MFMA lane holds
  row_scale_exp(m)       the per-row power of two s (an e8m0 exponent, 127 + s on the device) that puts a row maximum m into [4, 8):
                         e2m3 has no range to spare, so every operand row carries its own scale (free on the MX-scaled MFMA: the
                         scale operands are per lane = per row)
  corrected_product6(..) f16(x) . f16(W)^T + the two low-order corrections on e2m3 operands with per-row scales

Round 5 (second session): time-dithered f16 images of a recurrent layer's weights (evc_cast_f32_to_f16_dither; DESIGN.md 7, "dither"):

  f16_dither_images(w, T, seed)   T IEEE f16 images of w; image t holds, per element, one of w's two f16 NEIGHBOURS (dn <= w <= up), the upper one
                         when the element's 32-bit phase at step t - a hash of its flat index, rotated by the golden ratio per step - falls
                         below frac = (w - dn) / (up - dn): in any run of n consecutive steps an element is rounded up n frac times +- 2, so the
                         rounding errors a recurrence integrates over its steps cancel instead of adding up.  Integer arithmetic: bit-exact."""
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


def fp8_range_drop(absmax, hi_exp):
    """Bits d >= 0 by which an e4m3 image scaled by 2^hi_exp is shifted down so that its largest element stays representable:
    the fewest with absmax * 2^(hi_exp - d) <= 448 (csrc/evc_common.h::fp8_range_drop, on the maximum of the 64 partial maxima)."""
    absmax = float(absmax)
    if not (absmax > 0.0) or not np.isfinite(absmax):
        return 0
    m, k = np.frexp(np.float32(absmax))
    e = (9 if m <= 0.875 else 8) - int(k)
    return int(min(max(hi_exp - e, 0), 40))


def corrected_product_dyn(x, w, x_hi_exp, x_lo_exp, w_lo_exp, w_hi_exp):
    """corrected_product with the DYNAMIC range of the activation operand (evc_cast_f32_to_f16_fp8x_dyn + evc_gemm_nt_f16_fp8_dyn):
    both e4m3 images of x are scaled by 2^(exp - d), d = fp8_range_drop(max|x|, x_hi_exp); d = 0 is corrected_product itself."""
    d = fp8_range_drop(np.abs(np.asarray(x, np.float32)).max(), x_hi_exp)
    return corrected_product(x, w, x_hi_exp - d, x_lo_exp - d, w_lo_exp, w_hi_exp)


# ---- OCP FP6 e2m3 ------------------------------------------------------------------------------------------------------------
E2M3_MAX = 7.5


def e2m3_decode(codes):
    """6-bit codes (s eeee mmm -> s ee mmm) -> float64."""
    c = np.asarray(codes, np.int64)
    s, e, m = (c >> 5) & 1, (c >> 3) & 3, c & 7
    mag = np.where(e == 0, m * 0.125, (1.0 + m / 8.0) * 2.0 ** (e - 1.0))
    return np.where(s == 1, -mag, mag)


def e2m3_encode(a):
    """Round to nearest even on the e2m3 grid, saturating at +-7.5; -0 encodes as +0 (magnitude code 0 keeps the sign bit clear)."""
    a = np.asarray(a, np.float64)
    mag = np.minimum(np.abs(a), E2M3_MAX)
    # uniform grids per range: step 1/8 below 2, 1/4 in [2, 4), 1/2 in [4, 7.5]; np.rint rounds half to even, and the first code of a
    # range is even, so ties at a range boundary go the right way
    code = np.where(mag < 2.0, np.rint(mag * 8.0), np.where(mag < 4.0, 8.0 + np.rint(mag * 4.0), 16.0 + np.rint(mag * 2.0)))
    code = np.minimum(code, 31.0).astype(np.int64)
    return np.where((a < 0) & (code > 0), code | 32, code)


def e2m3_round(a):
    return e2m3_decode(e2m3_encode(a))


def pack_fp6(codes):
    """codes [..., 32 n] (6-bit ints) -> uint8 [..., 24 n]: element i of each 32-block at bits 6i .. 6i+5 of the block's 192 bits."""
    c = np.asarray(codes, np.uint64)
    assert c.shape[-1] % 32 == 0
    blk = c.reshape(c.shape[:-1] + (c.shape[-1] // 4, 4))                 # 4 codes = 24 bits = 3 bytes
    w = blk[..., 0] | (blk[..., 1] << np.uint64(6)) | (blk[..., 2] << np.uint64(12)) | (blk[..., 3] << np.uint64(18))
    out = np.stack([(w >> np.uint64(8 * i)) & np.uint64(255) for i in range(3)], -1).astype(np.uint8)
    return out.reshape(c.shape[:-1] + (c.shape[-1] // 4 * 3,))


def unpack_fp6(b):
    b = np.asarray(b, np.uint64)
    assert b.shape[-1] % 3 == 0
    t = b.reshape(b.shape[:-1] + (b.shape[-1] // 3, 3))
    w = t[..., 0] | (t[..., 1] << np.uint64(8)) | (t[..., 2] << np.uint64(16))
    out = np.stack([(w >> np.uint64(6 * i)) & np.uint64(63) for i in range(4)], -1).astype(np.int64)
    return out.reshape(b.shape[:-1] + (b.shape[-1] // 3 * 4,))


def row_scale_exp(absmax):
    """Per-row scale exponent s: absmax * 2^s in [4, 8) (values above 7.5 saturate: at most half a step of that range); rows of zeros
    get s = 0.  Integer array."""
    m = np.asarray(absmax, np.float64)
    return np.where(m > 0, 2 - np.floor(np.log2(np.where(m > 0, m, 1.0))), 0).astype(np.int64)


def e2m3_rows(a, exp=None):
    """The e2m3 image of a [R][K] under one power-of-two scale per row, back in a's units; exp: given per-row exponents (e.g. derived
    from a bound instead of the row's own maximum) or None = row_scale_exp(max |a| of the row)."""
    a = np.asarray(a, np.float64)
    if exp is None:
        exp = row_scale_exp(np.max(np.abs(a), axis=-1))
    sc = 2.0 ** np.asarray(exp, np.float64)[..., None]
    return e2m3_round(a * sc) / sc


def corrected_product6(x, w, correct_x=True):
    """x [M][K], w [N][K]: the f16 product plus the low-order corrections on e2m3 operands, every operand row under its own scale."""
    x, w = np.asarray(x, np.float64), np.asarray(w, np.float64)
    x16, w16 = f16_round(x), f16_round(w)
    z = x16 @ w16.T + e2m3_rows(x) @ e2m3_rows(w - w16).T
    if correct_x:
        z = z + e2m3_rows(x - x16) @ e2m3_rows(w).T
    return z


# ---- time-dithered f16 images (round 5) ---------------------------------------------------------------------------------------
DITHER_PHI32 = 0x9E3779B9          # round(2^32 / golden ratio): the per-step rotation of an element's phase


def _fmix32(h):
    """murmur3's 32-bit finaliser (uint32 arrays)."""
    h = np.asarray(h, np.uint32).copy()
    h ^= h >> np.uint32(16)
    h *= np.uint32(0x85EBCA6B)
    h ^= h >> np.uint32(13)
    h *= np.uint32(0xC2B2AE35)
    h ^= h >> np.uint32(16)
    return h


def f16_neighbours(w):
    """(dn, up) float16 arrays with dn <= w <= up, adjacent (or equal where w is an f16 value) - w taken as float32."""
    w32 = np.asarray(w, np.float32)
    h = w32.astype(np.float16)
    hf = h.astype(np.float32)
    with np.errstate(over="ignore"):
        up = np.where(hf >= w32, h, np.nextafter(h, np.float16(np.inf)))
        dn = np.where(hf <= w32, h, np.nextafter(h, np.float16(-np.inf)))
    return dn.astype(np.float16), up.astype(np.float16)


def f16_dither_threshold(w):
    """uint32 threshold of an element: round-up share frac = (w - dn) / (up - dn) as frac * 2^32 (f32 arithmetic, saturating below 2^32)."""
    w32 = np.asarray(w, np.float32)
    dn, up = f16_neighbours(w32)
    gap = up.astype(np.float32) - dn.astype(np.float32)
    frac = np.where(gap > 0, (w32 - dn.astype(np.float32)) / np.where(gap > 0, gap, np.float32(1)), np.float32(0)).astype(np.float32)
    return np.minimum(frac * np.float32(4294967296.0), np.float32(4294967040.0)).astype(np.uint32)


def f16_dither_images(w, T, seed=0, col0=0):
    """w [...] float32 -> float16 [T, ...]: image t of element i (flat row-major index) = up if uint32(fmix32(i ^ seed * PHI32) + t * PHI32) < thr else dn.
    col0 > 0 (w 2-D): the columns below col0 hold their round-to-nearest f16 value in every image."""
    w32 = np.ascontiguousarray(np.asarray(w, np.float32))
    dn, up = f16_neighbours(w32)
    if col0:
        rtn = w32.astype(np.float16)
        dn, up = dn.copy(), up.copy()
        dn[:, :col0] = rtn[:, :col0]
        up[:, :col0] = rtn[:, :col0]
    thr = f16_dither_threshold(w32).reshape(-1)
    idx = np.arange(w32.size, dtype=np.uint32)
    with np.errstate(over="ignore"):
        phase = _fmix32(idx ^ np.uint32((int(seed) * DITHER_PHI32) & 0xFFFFFFFF))
        out = np.empty((T,) + w32.shape, np.float16)
        for t in range(T):
            u = phase + np.uint32((t * DITHER_PHI32) & 0xFFFFFFFF)
            out[t] = np.where((u < thr).reshape(w32.shape), up, dn)
    return out
