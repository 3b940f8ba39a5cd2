"""Thin torch-tensor wrappers over the C ABI (one call per kernel family).

Tensors are only containers for device memory here: every wrapper extracts raw
device pointers and enqueues on torch's current HIP stream.  No wrapper has a
CPU or eager-PyTorch fallback; CPU tensors are rejected.
"""
from __future__ import annotations

import ctypes as C

import os

import torch

from . import _lib

BF16 = torch.bfloat16
F16 = torch.float16
F32 = torch.float32


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.EvcError("evc ops need device tensors (got a CPU tensor); there is no CPU path")
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


MARKS = None     # set to a list to collect (name, stream handle, timing event) at the engine's mark() points (scripts/step_timeline.py)


def mark(name):
    """Timeline aid: a timing event on the current stream, kept in ops.MARKS when that is a list (else nothing happens)."""
    if MARKS is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        MARKS.append((name, torch.cuda.current_stream().cuda_stream, ev))


def round_up(x, m):
    return (x + m - 1) // m * m


def check_device(dev=0):
    _lib.call("evc_check_device", dev)


# ---------------------------------------------------------------------------
def gemm_nt(A, B, M, N, K, out, bias=None, accumulate=False, lda=None, ldb=None, ldc=None):
    """out[M,N] (+)= A[M,K] @ B[N,K]^T (+ bias).  A, B bf16 K-contiguous."""
    assert A.dtype == BF16 and B.dtype == BF16
    lda = A.stride(0) if lda is None else lda
    ldb = B.stride(0) if ldb is None else ldb
    ldc = out.stride(0) if ldc is None else ldc
    _lib.call("evc_gemm_nt", _p(A), lda, _p(B), ldb, _p(out), ldc, M, N, K, _p(bias),
              1 if out.dtype == BF16 else 0, 1 if accumulate else 0, _stream())
    return out


def _slab_rows_arg(live_rows, K):
    """(slab_rows, nslabs, host int32 array) of evc_gemm_tn2_rows from live_rows = (rows per slab [T], rows of one slab)."""
    rows, slab = live_rows
    assert slab % 32 == 0 and K == slab * len(rows), (K, slab, len(rows))
    return slab, len(rows), (C.c_int32 * len(rows))(*[int(r) for r in rows])


def gemm_nt_sqnorm(A, B, M, N, K, out, p, l2_coeff, sums, ws=None):
    """out [M,N] f32 = A[M,K] @ B[N,K]^T and sums[0] += |out + l2_coeff * p|^2 from the same pass (evc_gemm_nt_sqnorm; p laid out as out, or None with
    l2_coeff 0; sums zeroed by the caller; ws: gemm_nt_sqnorm_ws(M, N) floats of scratch - per-wave partials summed in index order, no atomics)."""
    assert A.dtype == BF16 and B.dtype == BF16 and out.dtype == F32 and out.is_contiguous() and (p is None or (p.dtype == F32 and p.shape == out.shape and p.is_contiguous()))
    need = gemm_nt_sqnorm_ws(M, N)
    if ws is None:                      # (callers on several streams keep their own: engine.MoeHead.sq_part)
        ws = torch.empty(need, dtype=F32, device=out.device)
    assert ws.dtype == F32 and ws.numel() >= need and ws.is_contiguous()
    _lib.call("evc_gemm_nt_sqnorm", _p(A), A.stride(0), _p(B), B.stride(0), _p(out), out.stride(0), M, N, K, _p(p), float(l2_coeff), _p(sums), _p(ws), ws.numel(),
              _stream())
    return out


def gemm_nt_sqnorm_ws(M, N):
    """Floats of scratch evc_gemm_nt_sqnorm needs: one {|C + l2 P|^2, |P|^2} slot per wave of every tile."""
    return 16 * ((M + 127) // 128) * ((N + 127) // 128)


def gemm_nt_sqnorm_ok(M, N, K):
    """Shapes evc_gemm_nt_sqnorm takes (one pass of ring tiles storing whole rows) - and whether to use it: OFF unless EVC_FUSED_GRAD_NORM=1.  Measured on
    cfg 5 (the one configuration that materialises its MoE gradient), same box, alternating (profiles/r06_fused_grad_norm_ab.txt): 3.72-3.75 ms per step
    with the norm from the product's stores against 3.61-3.63 with the separate evc_grad_sqnorm pass - reading P and squaring in the store epilogue costs
    the two products more than the 0.19 ms pass it removes (with same-address atomics instead of per-wave partials: 3.95)."""
    return M > 512 and N % 256 == 0 and K % 64 == 0 and K < 8192 and os.environ.get("EVC_FUSED_GRAD_NORM", "0") == "1"


def gemm_tn(A, B, M, N, K, out, row_interleave_H=0, accumulate=False, lda=None, ldb=None, ldc=None, live_rows=None):
    """out[M,N] (+)= A[K,M]^T @ B[K,N] (both row-major over K), f32 out.  live_rows = (rows [T], slab_rows): K = T slabs of slab_rows rows
    of which the first rows[t] are live (a row-planned level) - the dead rows are skipped (evc_gemm_tn2_rows)."""
    assert A.dtype == BF16 and B.dtype == BF16 and out.dtype == F32
    if live_rows is not None:
        slab, ns, arr = _slab_rows_arg(live_rows, K)
        _lib.call("evc_gemm_tn2_rows", _p(A), A.stride(0) if lda is None else lda, _p(B), B.stride(0) if ldb is None else ldb, N, None, 0, 0, 0,
                  _p(out), out.stride(0) if ldc is None else ldc, M, slab, ns, arr, row_interleave_H, 1 if accumulate else 0, _stream())
        return out
    _lib.call("evc_gemm_tn", _p(A), A.stride(0) if lda is None else lda, _p(B), B.stride(0) if ldb is None else ldb,
              _p(out), out.stride(0) if ldc is None else ldc, M, N, K, row_interleave_H, 1 if accumulate else 0, _stream())
    return out


def gemm_tn_det(A, B, M, N, K, out, row_interleave_H=0, accumulate=False, ldc=None, B2=None, N2=0, c_col2=None):
    """out[:, :N] (and out[:, c_col2:c_col2+N2] from B2) (+)= A^T @ B without atomics and with the K split kept (EVC_DETERMINISTIC=1): partial
    products into slabs (evc_gemm_tn2_slabs), added in slab order (evc_sum_slabs).  Scratch from the stream-aware caching allocator."""
    assert A.dtype == BF16 and B.dtype == BF16 and out.dtype == F32
    ldc = out.stride(0) if ldc is None else ldc
    ntot = (c_col2 + N2) if B2 is not None else N
    tiles = ((M + 255) // 256) * ((N + N2 + 255) // 256)
    nslab = max(1, min(256 // max(1, tiles), K // 1024))
    while nslab > 1 and ((K // 32 + nslab - 1) // nslab) * (nslab - 1) >= K // 32:
        nslab -= 1
    ws = torch.empty((nslab, M, ntot), dtype=F32, device=out.device)
    _lib.call("evc_gemm_tn2_slabs", _p(A), A.stride(0), _p(B), B.stride(0), N, _p(B2), B2.stride(0) if B2 is not None else 0, N2,
              (c_col2 if c_col2 is not None else N), _p(ws), ntot, M * ntot, M, K, row_interleave_H, nslab, _stream())
    for c0, w in ((0, N),) + (((c_col2, N2),) if B2 is not None else ()):
        _lib.call("evc_sum_slabs", _p(ws[0, :, c0:]), M * ntot, nslab, M, w, ntot, _p(out[:, c0:]), ldc, 1 if accumulate else 0, _stream())
    return out


def gemm_tn2(A, B1, N1, B2, N2, M, K, out, row_interleave_H=0, accumulate=False, ldc=None, c_col2=None, live_rows=None):
    """out[:, :N1] (+)= A[K,M]^T @ B1[K,N1], out[:, c_col2:c_col2+N2] (+)= A^T @ B2[K,N2] in one launch (N1 % 256 == 0;
    c_col2 defaults to N1: adjacent segments).  live_rows: as in gemm_tn."""
    assert A.dtype == BF16 and B1.dtype == BF16 and B2.dtype == BF16 and out.dtype == F32
    if live_rows is not None:
        slab, ns, arr = _slab_rows_arg(live_rows, K)
        _lib.call("evc_gemm_tn2_rows", _p(A), A.stride(0), _p(B1), B1.stride(0), N1, _p(B2), B2.stride(0), N2, N1 if c_col2 is None else c_col2,
                  _p(out), out.stride(0) if ldc is None else ldc, M, slab, ns, arr, row_interleave_H, 1 if accumulate else 0, _stream())
        return out
    _lib.call("evc_gemm_tn2", _p(A), A.stride(0), _p(B1), B1.stride(0), N1, _p(B2), B2.stride(0), N2, N1 if c_col2 is None else c_col2,
              _p(out), out.stride(0) if ldc is None else ldc, M, K, row_interleave_H, 1 if accumulate else 0, _stream())
    return out


def fill_f32(t, value):
    """Contiguous f32 fill (one streaming kernel; hipMemset2D on a pitched view is several times slower)."""
    assert t.dtype == F32 and t.is_contiguous()
    _lib.call("evc_fill_f32", _p(t), t.numel(), float(value), _stream())


DETERMINISTIC = os.environ.get("EVC_DETERMINISTIC", "0") not in ("", "0")     # (the library reads the same variable: csrc/evc_common.h)


def colsum_bf16(x, R, C, out, deinterleave_H=0):
    if DETERMINISTIC and R >= 128:      # no atomics: partial rows + a fixed-order finish (evc_colsum_bf16_det)
        # (workspace from the caching allocator, which is stream-aware: the two towers call this concurrently on two streams)
        ws = torch.empty((128, C), dtype=F32, device=x.device)
        _lib.call("evc_colsum_bf16_det", _p(x), x.stride(0), R, C, deinterleave_H, _p(out), _p(ws), 128, _stream())
        return out
    _lib.call("evc_colsum_bf16", _p(x), x.stride(0), R, C, deinterleave_H, _p(out), _stream())
    return out


def transpose_to_bf16(x, R, C, out, Rpad, ld_in=None, interleave_H=0):
    """out[c][r] = x[r][c]; out is [C, >=Rpad] bf16 with columns [R,Rpad) zeroed.
    interleave_H: write input row g*H+u to output column u*4+g (gate-interleaved K order)."""
    ld_in = x.stride(0) if ld_in is None else ld_in
    _lib.call("evc_transpose_to_bf16", _p(x), 1 if x.dtype == F32 else 0, ld_in, R, C, _p(out), out.stride(0), Rpad,
              interleave_H, _stream())
    return out


def cast_bf16(x, out=None):
    x2 = x.reshape(-1, x.shape[-1]) if x.dim() > 1 else x.reshape(1, -1)
    if out is None:
        out = torch.empty(x.shape, dtype=BF16, device=x.device)
    o2 = out.reshape(x2.shape)
    _lib.call("evc_cast_f32_to_bf16", _p(x2), x2.stride(0), x2.shape[0], x2.shape[1], _p(o2), o2.stride(0), _stream())
    return out


def cast_f16(x, out):
    """out = f16(x), round to nearest even, for a 2-D f32 tensor (the f16 weight shadows of lstm_layer_fwd_f16)."""
    assert x.dtype == F32 and out.dtype == F16 and x.dim() == 2
    _lib.call("evc_cast_f32_to_f16", _p(x), x.stride(0), x.shape[0], x.shape[1], _p(out), out.stride(0), _stream())
    return out


def cast_f16_wide(w, Kin, H, nseg, out, h_ext=False):
    """f16 image of an LSTM kernel w [R][Kin+H] f32 for K-extended operands: out [R][nseg*Kin + (2 if h_ext else 1)*H] =
    [f16(Wx) | f16(Wx)/64 | (Wx - f16(Wx))*64 | f16(Wh) | (Wh - f16(Wh))*64] (first nseg x blocks; the last block with h_ext)."""
    assert w.dtype == F32 and out.dtype == F16 and w.shape[1] == Kin + H and out.is_contiguous()
    assert out.shape == (w.shape[0], nseg * Kin + (2 if h_ext else 1) * H)
    _lib.call("evc_cast_f32_to_f16_wide", _p(w), w.stride(0), w.shape[0], Kin, H, nseg, 1 if h_ext else 0, _p(out), _stream())
    return out


def cast_f16_segs(x, nseg, out):
    """K-extended f16 image of a 2-D f32 activation matrix: out [R][nseg*C] = [f16(x) | (x - f16(x))*64 | f16(x)/64] (first nseg)."""
    assert x.dtype == F32 and out.dtype == F16 and out.shape == (x.shape[0], nseg * x.shape[1]) and out.is_contiguous()
    _lib.call("evc_cast_f32_to_f16_segs", _p(x), x.stride(0), x.shape[0], x.shape[1], nseg, _p(out), _stream())
    return out


def cast_f16_wlo(w, Kin, H, out):
    """f16 image of an LSTM kernel w [R][Kin+H] f32 with both parts K-extended by the weights' low-order halves:
    out [R][2Kin + 2H] = [f16(Wx) | (Wx - f16(Wx))*64 | f16(Wh) | (Wh - f16(Wh))*64] (upper layer of lstm_stack2_fwd_f16)."""
    assert w.dtype == F32 and out.dtype == F16 and w.shape[1] == Kin + H and out.shape == (w.shape[0], 2 * (Kin + H)) and out.is_contiguous()
    _lib.call("evc_cast_f32_to_f16_wlo", _p(w), w.stride(0), w.shape[0], Kin, H, _p(out), _stream())
    return out


FP8_W_SCALE_EXP = 17       # e4m3((W - f16(W)) * 2^17): |W| < 4 never clamps, weights above ~2^-13 keep 3-4 bits of their low-order half
FP8_WX_HI_EXP = 6          # e4m3(Wx * 2^6) against e4m3((x - f16(x)) * 2^18): the same 2^24 as 2^7 * 2^17


def cast_fp8_lo(w, out, hi_cols=0, scale_exp=FP8_W_SCALE_EXP, hi_exp=FP8_WX_HI_EXP, hi_tail=False):
    """out uint8 = e4m3(clamp((w - f16(w)) * 2^scale_exp)): the low-order halves of a weight matrix next to its f16 image (the wT8 operand of
    lstm_layer_fwd_f16_fp8lo).  hi_cols > 0: out [R][C + hi_cols] = [lo(W[:, :hi_cols]) | e4m3(W[:, :hi_cols] * 2^hi_exp) | lo(W[:, hi_cols:])] -
    the layer that reads the input frames contracts the input's low-order half against the full-value image.
    hi_tail (round 6): EVERY column gets its full-value image - out [R][2C] = [lo(A) | hi(A) | lo(B) | hi(B)], A = W[:, :hi_cols], B = the rest
    (evc_cast_f32_to_fp8_lohi: the weight rows of the h_lo forms, whose activations' low-order halves are corrected too)."""
    assert w.dtype == F32 and out.dtype == torch.uint8 and w.dim() == 2
    if hi_tail:
        assert out.shape == (w.shape[0], 2 * w.shape[1])
        _lib.call("evc_cast_f32_to_fp8_lohi", _p(w), w.stride(0), w.shape[0], w.shape[1], scale_exp, hi_cols, hi_exp, _p(out), out.stride(0), _stream())
        return out
    assert out.shape == (w.shape[0], w.shape[1] + hi_cols)
    _lib.call("evc_cast_f32_to_fp8_lo", _p(w), w.stride(0), w.shape[0], w.shape[1], scale_exp, hi_cols, hi_exp, _p(out), out.stride(0), _stream())
    return out


FP8_MOE = dict(x_hi_exp=6, x_lo_exp=17, w_lo_exp=18, w_hi_exp=7)     # e4m3 scales of the "high" MoE head: |state| < 7, |W| < 3.5 never clamp; 6 + 18 = 17 + 7 = 24


AMAX_SLOTS = 64     # EVC_AMAX_SLOTS (csrc/evc_common.h)


def absmax_partials(x, ws):
    """ws [64] f32 = partial maxima of |x| over a 2-D f32 tensor (evc_absmax_partials: plain stores, nothing to zero)."""
    assert x.dtype == F32 and x.dim() == 2 and ws.dtype == F32 and ws.numel() == AMAX_SLOTS and ws.is_contiguous()
    _lib.call("evc_absmax_partials", _p(x), x.stride(0), x.shape[0], x.shape[1], _p(ws), _stream())
    return ws


def cast_f16_fp8x(x, out, hi_exp=FP8_MOE["x_hi_exp"], lo_exp=FP8_MOE["x_lo_exp"], amax_ws=None):
    """out [R][2C] f16 containers = rows [f16(x) | e4m3(x 2^hi_exp) (C bytes) | e4m3((x - f16(x)) 2^lo_exp) (C bytes)]: the A operands of gemm_nt_f16_fp8.
    amax_ws (absmax_partials(x)): dynamic range - both e4m3 images are shifted down by the d bits the largest |x| needs to stay below 448
    (evc_cast_f32_to_f16_fp8x_dyn; gemm_nt_f16_fp8 takes the same amax_ws / hi_exp and scales its products back up)."""
    assert x.dtype == F32 and x.dim() == 2 and out.dtype == F16 and out.shape == (x.shape[0], 2 * x.shape[1]) and out.is_contiguous()
    if amax_ws is not None:
        _lib.call("evc_cast_f32_to_f16_fp8x_dyn", _p(x), x.stride(0), x.shape[0], x.shape[1], hi_exp, lo_exp, _p(amax_ws), _p(out), _stream())
        return out
    _lib.call("evc_cast_f32_to_f16_fp8x", _p(x), x.stride(0), x.shape[0], x.shape[1], hi_exp, lo_exp, _p(out), _stream())
    return out


def gemm_nt_f16_fp8(a_rows, w16, w8, M, N, K, out, bias=None, scale_exp=-24, amax_ws=None, a_hi_exp=FP8_MOE["x_hi_exp"]):
    """out [M][N] f32 = f16(x) . f16(W)^T + 2^scale_exp [e4m3(x ..) | e4m3(x_lo ..)] . w8^T (+ bias): a_rows [M][2K] from cast_f16_fp8x, w16 [N][K]
    f16, w8 [N][2K] uint8 = [e4m3(W_lo ..) | e4m3(W ..)] (evc_gemm_nt_f16_fp8: the "high" precision MoE head).  amax_ws / a_hi_exp: what
    cast_f16_fp8x(amax_ws=...) wrote a_rows with (evc_gemm_nt_f16_fp8_dyn)."""
    assert a_rows.dtype == F16 and a_rows.shape == (M, 2 * K) and w16.dtype == F16 and w16.shape == (N, K) and w8.dtype == torch.uint8 and w8.shape == (N, 2 * K)
    assert out.dtype == F32 and a_rows.is_contiguous() and w16.is_contiguous() and w8.is_contiguous()
    if amax_ws is not None:
        _lib.call("evc_gemm_nt_f16_fp8_dyn", _p(a_rows), 2 * K, a_rows.data_ptr() + 2 * K, 4 * K, _p(w16), K, _p(w8), 2 * K, _p(out), out.stride(0),
                  M, N, K, 2 * K, scale_exp, _p(amax_ws), a_hi_exp, _p(bias), _stream())
        return out
    _lib.call("evc_gemm_nt_f16_fp8", _p(a_rows), 2 * K, a_rows.data_ptr() + 2 * K, 4 * K, _p(w16), K, _p(w8), 2 * K, _p(out), out.stride(0),
              M, N, K, 2 * K, scale_exp, _p(bias), _stream())
    return out


def cast_bf16_wide(x, out, lo_first):
    """Wide split-bf16 image of a 2-D f32 tensor: out rows [lo | hi] (lo_first: the A operand of gemm_nt_split_wide) or
    [hi | lo] (its B operand)."""
    R, Cc = x.shape
    assert x.dtype == F32 and out.dtype == BF16 and out.shape[0] == R and out.shape[1] >= 2 * Cc
    _lib.call("evc_cast_f32_to_bf16_wide", _p(x), x.stride(0), R, Cc, _p(out), out.stride(0), 1 if lo_first else 0, _stream())
    return out


def gemm_nt_split_wide(A_lohi, B_hilo, M, N, K, out, bias=None):
    """out[M,N] f32 = (A_hi + A_lo) @ (B_hi + B_lo)^T (+ bias) to ~2^-16 as ONE K-extended launch (evc_gemm_nt_split):
    A_lohi rows [lo(K) | hi(K)], B_hilo rows [hi(K) | lo(K)]."""
    assert A_lohi.dtype == BF16 and B_hilo.dtype == BF16 and out.dtype == F32
    _lib.call("evc_gemm_nt_split", _p(A_lohi), A_lohi.stride(0), _p(B_hilo), B_hilo.stride(0), _p(out), out.stride(0), M, N, K,
              _p(bias), _stream())
    return out


def cast_bf16_split(x, hi, lo):
    """hi = bf16(x), lo = bf16(x - hi) for a 2-D f32 tensor (split-bf16 operands)."""
    R, Cc = x.shape
    _lib.call("evc_cast_f32_to_bf16_split", _p(x), x.stride(0), R, Cc, _p(hi), _p(lo), hi.stride(0), _stream())
    return hi, lo


def gemm_nt_split(A_hi, A_lo, B_hi, B_lo, M, N, K, out, bias=None):
    """out = (A_hi + A_lo) @ (B_hi + B_lo)^T to ~2^-16: hi.hi + hi.lo + lo.hi (the lo.lo term is below f32 noise)."""
    gemm_nt(A_hi, B_hi, M, N, K, out, bias=bias)
    gemm_nt(A_hi, B_lo, M, N, K, out, accumulate=True)
    gemm_nt(A_lo, B_hi, M, N, K, out, accumulate=True)
    return out


def rowsum_bf16(x, R, C, out):
    _lib.call("evc_rowsum_bf16", _p(x), x.stride(0), R, C, _p(out), _stream())
    return out


# ---------------------------------------------------------------------------
class RowPlan:
    """Row order of one LSTM stack for one batch (evc_sort_rows_by_len): rows sorted by sequence length,
    longest first, so that the rows active at step t are the prefix [0, rows[t]) and the length-0 rows
    (frames beyond num_frames) drop out of every kernel.

    pos / inv / lens  device int32 [M]: slot of each row, row of each slot, length of each slot
    P                 rows kept per time slab (rows[0] rounded up to 32, at most M): all [T][M][..] buffers
                      of the stack are used as [T][P][..]
    rows              host list [T]: rows[t] = #{len > t}, computed from the HOST copy of the lengths (the
                      launch geometry depends on it; a device->host read would stall the stream)
    """

    def __init__(self, lens_dev, lens_host, T):
        import numpy as np
        lens_host = np.asarray(lens_host)
        M = int(lens_dev.shape[0])
        assert lens_host.shape == (M,)
        self.M, self.T = M, T
        hist = np.bincount(np.clip(lens_host, 0, T), minlength=T + 1)
        self.rows = [int(M - hist[:t + 1].sum()) for t in range(T)]         # rows with len > t
        self.P = min(M, max(32, round_up(self.rows[0], 32)))
        self.rows_c = (C.c_int32 * T)(*self.rows)
        dev = lens_dev.device
        self.pos = torch.empty(M, dtype=torch.int32, device=dev)
        self.inv = torch.empty(M, dtype=torch.int32, device=dev)
        self.lens = torch.empty(M, dtype=torch.int32, device=dev)
        _lib.call("evc_sort_rows_by_len", _p(lens_dev), M, T, _p(self.pos), _p(self.inv), _p(self.lens), _stream())


def host_frame_counts(num_frames_host, every_n, num_chunks, chunk_len, max_frames=300, subsampled=None):
    """Host (numpy) twin of evc_frame_counts, bit-identical: (n_used int64 [B], len_l1 int32 [C*B], len_l2 int32 [B]).
    subsampled: the student formula of cs/train.py:264 (default: every_n > 1; the student graph passes True also at
    every_n = 1, where float64 (n/300)*300 truncates to n-1 for some n)."""
    import numpy as np
    n = np.asarray(num_frames_host).astype(np.int64)
    if (every_n > 1) if subsampled is None else subsampled:
        S = max_frames // every_n
        n = np.trunc(n.astype(np.float64) / float(max_frames) * float(S)).astype(np.int64)
    l1 = np.clip(n[None, :] - chunk_len * np.arange(num_chunks, dtype=np.int64)[:, None], 0, chunk_len).astype(np.int32).reshape(-1)
    l2 = np.ceil(n.astype(np.float32) / np.float32(chunk_len)).astype(np.int32)
    return n, l1, l2


def l2norm_chunk(x_raw, num_chunks, every_n=None, num_chunks_student=None, num_frames=None, normalize=True, split=False,
                 plan1=None, plan2=None, f16_segments=1, fp8_tail=False, teacher_view=True):
    """a1+a2.  x_raw [B,T,F] f32 (or uint8 with num_frames).  Returns the
    teacher view [Lc][C*B][F] bf16 and (if every_n) the student view; with row plans the views are
    [Lc][plan.P][F] in slot order.  split: True -> (bf16, bf16 low half) pairs; "f16" -> (bf16, IEEE f16 image) pairs (the
    operands of the "high" precision L1 forward; the bf16 image stays the operand of the backward products) - the f16 image
    has rows of f16_segments*F: [f16(x) | (x - f16(x))*64 | f16(x)/64], the K-extended x operand of lstm_layer_fwd_f16;
    "wide" -> (bf16, wide bf16 image with rows [lo | hi] of 2F) pairs, the input of lstm_layer_fwd_hp.
    fp8_tail (with split "f16", f16_segments 1): the f16 image's rows are 2F halfwords = [f16(x) | e4m3(x 2^7) (F bytes) | e4m3((x - f16(x)) 2^18)
    (F bytes)], the x rows of lstm_layer_fwd_f16_fp8lo (evc_l2norm_chunk_fwd aux_mode 5).
    teacher_view=False (student-only graphs; needs every_n): the first view is None and only the sub-sampled frames of x_raw are read."""
    B, T, F = x_raw.shape
    dev = x_raw.device
    assert teacher_view or every_n, "l2norm_chunk: no view requested"
    rows1 = (plan1.P if plan1 is not None else num_chunks * B) if teacher_view else 0
    out1 = torch.empty((T // num_chunks, rows1, F), dtype=BF16, device=dev) if teacher_view else None
    out2 = None
    rows2 = 0
    if every_n:
        S = T // every_n
        rows2 = plan2.P if plan2 is not None else num_chunks_student * B
        out2 = torch.empty((S // num_chunks_student, rows2, F), dtype=BF16, device=dev)
    is_u8 = x_raw.dtype == torch.uint8
    aux_dt = F16 if split == "f16" else BF16
    nseg = f16_segments if split == "f16" else (2 if split == "wide" else 1)
    aux_mode = nseg if split == "f16" else (4 if split == "wide" else 0)
    wrow = nseg * F
    if fp8_tail:
        assert split == "f16" and f16_segments == 1 and F % 32 == 0, "fp8_tail: the f16 image + two e4m3 images, F % 32 == 0"
        aux_mode, wrow = 5, 2 * F
    lo1 = torch.empty(out1.shape[:2] + (wrow,), dtype=aux_dt, device=dev) if (split and out1 is not None) else None
    lo2 = torch.empty(out2.shape[:2] + (wrow,), dtype=aux_dt, device=dev) if (split and out2 is not None) else None
    _lib.call("evc_l2norm_chunk_fwd", None if is_u8 else _p(x_raw), _p(x_raw) if is_u8 else None, _p(num_frames),
              B, T, F, num_chunks, _p(out1), every_n or 1, num_chunks_student or 1, _p(out2), 1 if normalize else 0,
              _p(lo1), _p(lo2), aux_mode, _p(plan1.pos) if (plan1 is not None and teacher_view) else None, rows1,
              _p(plan2.pos) if plan2 is not None else None, rows2, _stream())
    if split:      # image pairs for the "high" / "split" precision forward
        return ((out1, lo1) if out1 is not None else None), ((out2, lo2) if out2 is not None else None)
    return out1, out2


def frame_counts(num_frames, every_n, num_chunks, chunk_len, max_frames=300, subsampled=None):
    """a2 integer part.  Returns (n_used int64 [B], len_l1 int32 [C*B], len_l2 int32 [B])."""
    sub = (every_n > 1) if subsampled is None else subsampled
    B = num_frames.shape[0]
    dev = num_frames.device
    n_out = torch.empty(B, dtype=torch.int64, device=dev)
    l1 = torch.empty(num_chunks * B, dtype=torch.int32, device=dev)
    l2 = torch.empty(B, dtype=torch.int32, device=dev)
    _lib.call("evc_frame_counts", _p(num_frames), B, every_n, 1 if sub else 0, max_frames, num_chunks, chunk_len, _p(n_out), _p(l1), _p(l2), _stream())
    return n_out, l1, l2


# ---------------------------------------------------------------------------
def _plan_args(plan):
    return (_p(plan.inv), plan.rows_c) if plan is not None else (None, None)


def lstm_layer_fwd(x, wT, bias, lens, T, M, Kin, H, hbuf, c_state, h_state, ld_state,
                   gates=None, c_all=None, hoist=False, zx_ws=None, plan=None):
    """With a RowPlan: M = plan.P, lens = plan.lens, all [T][M][..] operands in slot order."""
    _lib.call("evc_lstm_layer_fwd", _p(x), _p(wT), _p(bias), _p(lens), T, M, Kin, H, 1 if hoist else 0, _p(zx_ws),
              _p(hbuf), _p(c_state), _p(h_state), ld_state, _p(gates), _p(c_all), *_plan_args(plan), _stream())


def lstm_layer_fwd_f16(x16, wT16, bias, lens, T, M, Kin, H, hbuf16, hbuf_bf, c_state, h_state, ld_state,
                       gates=None, c_all=None, plan=None, ldx=None, h_wide=False):
    """lstm_layer_fwd on IEEE f16 operands (one f16 MFMA product per depth); hbuf16 f16 (h_wide: rows [h | h/64] of 2H against a
    kernel whose h-part is [Wh | Wh_lo*64]), hbuf_bf the bf16 copy of h; ldx: row stride of x16 (default Kin)."""
    assert x16.dtype == F16 and wT16.dtype == F16 and hbuf16.dtype == F16 and hbuf_bf.dtype == BF16
    assert wT16.shape[1] == Kin + (2 if h_wide else 1) * H
    _lib.call("evc_lstm_layer_fwd_f16", _p(x16), Kin if ldx is None else ldx, _p(wT16), _p(bias), _p(lens), T, M, Kin, H, _p(hbuf16),
              1 if h_wide else 0, _p(hbuf_bf), _p(c_state), _p(h_state), ld_state, _p(gates), _p(c_all), *_plan_args(plan), _stream())


def lstm_layer_fwd_f16_fp8lo(x16, ldx, kx16, x8_off, kx8, wT16, wT8, bias, lens, T, M, H, hbuf16, hbuf_bf, c_state, h_state, ld_state,
                             gates=None, c_all=None, plan=None, w8_scale_exp=FP8_W_SCALE_EXP, h_lo=False, x_int=None, b8_gap=0):
    """lstm_layer_fwd_f16 with the weights' low-order halves contracted in fp8 (evc_lstm_layer_fwd_f16_fp8lo): x16 rows of ldx halfwords
    (kx16 halfwords of f16 operand at the row start, kx8 e4m3 bytes at byte offset x8_off), wT16 [4H][kx16 + H] f16, wT8 [4H][kx8 + H]
    uint8 (cast_fp8_lo), hbuf16 [(T+1)][M][3H/2] f16 containers = rows [f16(h) | e4m3(h 2^7)], hbuf_bf the bf16 copy of h.
    h_lo: the low-order half of h corrected too - hbuf16 [(T+1)][M][2H] containers = rows [f16(h) | e4m3(h 2^7) | e4m3((h - f16(h)) 2^18)],
    wT8 [4H][kx8 + 2H] with the h-part [lo(Wh) | hi(Wh)] (cast_fp8_lo(hi_tail=True)).
    x_int = (row_scale [T][M] f32, col_const [4H] f32): the integer-frame form (x16 rows from l2norm_chunk_int: exact integers + e4m3(x_hat 2^7));
    b8_gap: bytes of wT8's rows between the x-part this launch contracts and the h-part (the hi(Wx) block)."""
    assert x16.dtype == F16 and wT16.dtype == F16 and wT8.dtype == torch.uint8 and hbuf16.dtype == F16 and hbuf_bf.dtype == BF16
    assert wT16.shape == (4 * H, kx16 + H) and wT8.shape == (4 * H, kx8 + b8_gap + (2 if h_lo else 1) * H) and wT16.is_contiguous() and wT8.is_contiguous()
    rs, cc = x_int if x_int is not None else (None, None)
    assert x_int is None or (rs.dtype == F32 and cc.dtype == F32 and rs.numel() >= T * M and cc.numel() == 4 * H and rs.is_contiguous() and cc.is_contiguous())
    assert hbuf16.shape[-1] == (2 * H if h_lo else 3 * H // 2)
    _lib.call("evc_lstm_layer_fwd_f16_fp8lo", _p(x16), ldx, kx16, x8_off, kx8, _p(wT16), _p(wT8), w8_scale_exp, 1 if h_lo else 0, _p(bias), _p(lens), T, M, H,
              _p(hbuf16), _p(hbuf_bf), _p(c_state), _p(h_state), ld_state, _p(gates), _p(c_all), *_plan_args(plan), _p(rs), _p(cc), b8_gap, _stream())


def l2norm_chunk_int(x_u8, num_frames, num_chunks, every_n=None, num_chunks_student=None, plan1=None, plan2=None, teacher_view=True):
    """l2norm_chunk for the "high" mode on the reader's uint8 frames (evc_l2norm_chunk_int): per view (bf16 image, integer image, row scales) -
    integer image rows of 3F/2 containers [f16(2q - 255) | e4m3(x_hat 2^7)], row scales [steps][rows] f32 with x_hat = rs (c + 255/256)."""
    B, T, F = x_u8.shape
    assert x_u8.dtype == torch.uint8 and F % 32 == 0 and (teacher_view or every_n)
    dev = x_u8.device
    rows1 = (plan1.P if plan1 is not None else num_chunks * B) if teacher_view else 0
    L1 = T // num_chunks
    out1 = torch.empty((L1, rows1, F), dtype=BF16, device=dev) if teacher_view else None
    int1 = torch.empty((L1, rows1, 3 * F // 2), dtype=F16, device=dev) if teacher_view else None
    rs1 = torch.zeros((L1, rows1), dtype=F32, device=dev) if teacher_view else None
    out2 = int2 = rs2 = None
    rows2 = 0
    if every_n:
        S = T // every_n
        rows2 = plan2.P if plan2 is not None else num_chunks_student * B
        L2 = S // num_chunks_student
        out2 = torch.empty((L2, rows2, F), dtype=BF16, device=dev)
        int2 = torch.empty((L2, rows2, 3 * F // 2), dtype=F16, device=dev)
        rs2 = torch.zeros((L2, rows2), dtype=F32, device=dev)
    _lib.call("evc_l2norm_chunk_int", _p(x_u8), _p(num_frames), B, T, F, num_chunks, _p(out1), every_n or 1, num_chunks_student or 1, _p(out2),
              _p(int1), _p(int2), _p(rs1), _p(rs2), _p(plan1.pos) if (plan1 is not None and teacher_view) else None, rows1,
              _p(plan2.pos) if plan2 is not None else None, rows2, _stream())
    return ((out1, int1, rs1) if teacher_view else None), ((out2, int2, rs2) if every_n else None)


def lstm_layer_fwd_f16_dith(x16, ldx, kx16, x8_off, kx8, wT16_steps, wT8, ldb8, scale8_exp, bias, lens, T, M, H, hbuf16, hbuf_bf, c_state, h_state,
                            ld_state, gates=None, c_all=None, plan=None):
    """lstm_layer_fwd_f16 on time-dithered weight images (evc_lstm_layer_fwd_f16_dith): wT16_steps [T'][4H][kx16 + H] f16 from cast_f16_dither
    (T' >= T; T' == 1: one image for every step), x16 rows of ldx halfwords with kx8 e4m3 bytes at byte offset x8_off (kx8 = 0: none) against
    the first kx8 bytes of wT8's rows (row stride ldb8; a view into a cast_fp8_lo image), every e4m3 product scaled by 2^-scale8_exp;
    hbuf16 [(T+1)][M][H] plain f16 rows, hbuf_bf the bf16 copy of h."""
    assert x16.dtype == F16 and wT16_steps.dtype == F16 and hbuf16.dtype == F16 and hbuf_bf.dtype == BF16
    assert wT16_steps.dim() == 3 and wT16_steps.shape[1:] == (4 * H, kx16 + H) and wT16_steps.is_contiguous() and wT16_steps.shape[0] in (1, ) + tuple(range(T, 4097))
    assert (kx8 == 0) == (wT8 is None) and (wT8 is None or wT8.dtype == torch.uint8)
    stride = 0 if wT16_steps.shape[0] == 1 else wT16_steps.stride(0)
    _lib.call("evc_lstm_layer_fwd_f16_dith", _p(x16), ldx, kx16, x8_off, kx8, _p(wT16_steps), stride, _p(wT8), ldb8, scale8_exp, _p(bias), _p(lens), T, M, H,
              _p(hbuf16), _p(hbuf_bf), _p(c_state), _p(h_state), ld_state, _p(gates), _p(c_all), *_plan_args(plan), _stream())


def lstm_level2_fwd_high(x16, ldx, kx16, x8_off, kx8, wT16_0, wT8_0, bias0, wT16_1_steps, bias1, lens, T, M, H, h0_rows, hbuf0, h1_rows, hbuf1, S,
                         gates=(None, None), c_all=(None, None), plan=None, w8_scale_exp=FP8_W_SCALE_EXP, h_lo=False, x_int=None, b8_gap=0):
    """The two-layer L1 level of the "high" mode as T + 1 two-tile launches (evc_lstm_level2_fwd_high): lstm_layer_fwd_f16_fp8lo for layer 0 (same
    arguments) + lstm_layer_fwd_f16_dith(kx8 = 0) for layer 1 on wT16_1_steps [T'][4H][2H], bit for bit.  h0_rows [(T+1)][M][2H or 3H/2] f16 containers,
    h1_rows [(T+1)][M][H] f16, hbuf0 / hbuf1 the bf16 copies; S [rows][4H] f32 = [c0 | h0 | c1 | h1]."""
    assert x16.dtype == F16 and wT16_0.dtype == F16 and wT8_0.dtype == torch.uint8 and wT16_1_steps.dtype == F16
    assert h0_rows.dtype == F16 and h1_rows.dtype == F16 and hbuf0.dtype == BF16 and hbuf1.dtype == BF16
    assert wT16_0.shape == (4 * H, kx16 + H) and wT8_0.shape == (4 * H, kx8 + b8_gap + (2 if h_lo else 1) * H) and wT16_0.is_contiguous() and wT8_0.is_contiguous()
    assert wT16_1_steps.dim() == 3 and wT16_1_steps.shape[1:] == (4 * H, 2 * H) and wT16_1_steps.is_contiguous() and (wT16_1_steps.shape[0] == 1 or wT16_1_steps.shape[0] >= T)
    assert h0_rows.shape[-1] == (2 * H if h_lo else 3 * H // 2) and h1_rows.shape[-1] == H
    rs, cc = x_int if x_int is not None else (None, None)
    assert x_int is None or (rs.dtype == F32 and cc.dtype == F32 and rs.numel() >= T * M and cc.numel() == 4 * H and rs.is_contiguous() and cc.is_contiguous())
    stride = 0 if wT16_1_steps.shape[0] == 1 else wT16_1_steps.stride(0)
    _lib.call("evc_lstm_level2_fwd_high", _p(x16), ldx, kx16, x8_off, kx8, _p(wT16_0), _p(wT8_0), w8_scale_exp, 1 if h_lo else 0, _p(bias0), _p(rs), _p(cc), b8_gap,
              _p(wT16_1_steps), stride, _p(bias1), _p(lens), T, M, H, _p(h0_rows), _p(hbuf0), _p(h1_rows), _p(hbuf1),
              _p(S[:, 0:]), _p(S[:, H:]), _p(S[:, 2 * H:]), _p(S[:, 3 * H:]), S.stride(0),
              _p(gates[0]), _p(c_all[0]), _p(gates[1]), _p(c_all[1]), *_plan_args(plan), _stream())


def cast_f16_dither(p, out, seed, col0=0):
    """out [T][...p.shape] f16: the T time-dithered f16 images of the f32 tensor p (evc_cast_f32_to_f16_dither; oracle/lowprec.py::f16_dither_images).
    col0 > 0 (p 2-D): only the columns from col0 on are dithered, the others hold their round-to-nearest value in every image."""
    assert p.dtype == F32 and p.is_contiguous() and out.dtype == F16 and out.is_contiguous() and out.shape[1:] == p.shape
    assert col0 == 0 or (p.dim() == 2 and 0 < col0 <= p.shape[1])
    _lib.call("evc_cast_f32_to_f16_dither", _p(p), p.numel(), out.shape[0], out.stride(0) if out.shape[0] > 1 else p.numel(), int(seed) & 0xFFFFFFFF, _p(out),
              p.shape[1] if col0 else 0, col0, _stream())


def lstm_level2_fwd(x, wT0, bias0, wT1, bias1, lens, T, M, Kin, H, hbuf0, hbuf1, S, gates=(None, None), c_all=(None, None), plan=None):
    """Two-layer L1 level in bf16, layer 0's step s and layer 1's step s-1 per launch (evc_lstm_level2_fwd): the results of two
    lstm_layer_fwd calls, bit for bit.  S [rows][4H] f32 = [c0 | h0 | c1 | h1]; with a RowPlan M = plan.P, lens = plan.lens."""
    assert x.dtype == BF16 and wT0.dtype == BF16 and wT1.dtype == BF16 and hbuf0.dtype == BF16 and hbuf1.dtype == BF16
    _lib.call("evc_lstm_level2_fwd", _p(x), _p(wT0), _p(bias0), _p(wT1), _p(bias1), _p(lens), T, M, Kin, H, _p(hbuf0), _p(hbuf1),
              _p(S[:, 0:]), _p(S[:, H:]), _p(S[:, 2 * H:]), _p(S[:, 3 * H:]), S.stride(0),
              _p(gates[0]), _p(c_all[0]), _p(gates[1]), _p(c_all[1]), *_plan_args(plan), _stream())


def lstm_stack2_fwd(x, wT0, bias0, wT1, bias1, lens, T, M, Kin, H, zx_ws, hbuf0, hbuf1, S, gates=(None, None), c_all=(None, None)):
    """Two-layer stack, M ~ batch rows, wavefront order (evc_lstm_stack2_fwd).  S [M][4H] f32 = [c0 | h0 | c1 | h1]."""
    _lib.call("evc_lstm_stack2_fwd", _p(x), _p(wT0), _p(bias0), _p(wT1), _p(bias1), _p(lens), T, M, Kin, H, _p(zx_ws),
              _p(hbuf0), _p(hbuf1), _p(S[:, 0:]), _p(S[:, H:]), _p(S[:, 2 * H:]), _p(S[:, 3 * H:]), S.stride(0),
              _p(gates[0]), _p(c_all[0]), _p(gates[1]), _p(c_all[1]), _stream())


def lstm_stack2_fwd_f16(x16, wT0_16, bias0, wT1_wlo, bias1, lens, T, M, Kin, H, zx_ws, h0_wide, h1_wide, hbuf0, hbuf1, S,
                        gates=(None, None), c_all=(None, None), x_segments=1, h0_ext=False):
    """Two-layer stack, M ~ batch rows, wavefront order, IEEE f16 operands with the upper layer's weights K-extended by their
    low-order halves (evc_lstm_stack2_fwd_f16).  h*_wide [(T+1)][M][2H] f16, hbuf* [(T+1)][M][H] bf16; S [M][4H] f32."""
    assert x16.dtype == F16 and wT0_16.dtype == F16 and wT1_wlo.dtype == F16 and h0_wide.dtype == F16 and hbuf0.dtype == BF16
    assert x16.shape[-1] == x_segments * Kin and wT0_16.shape[1] == x_segments * Kin + (2 if h0_ext else 1) * H
    _lib.call("evc_lstm_stack2_fwd_f16", _p(x16), x_segments, _p(wT0_16), 1 if h0_ext else 0, _p(bias0), _p(wT1_wlo), _p(bias1), _p(lens),
              T, M, Kin, H, _p(zx_ws),
              _p(h0_wide), _p(h1_wide), _p(hbuf0), _p(hbuf1), _p(S[:, 0:]), _p(S[:, H:]), _p(S[:, 2 * H:]), _p(S[:, 3 * H:]), S.stride(0),
              _p(gates[0]), _p(c_all[0]), _p(gates[1]), _p(c_all[1]), _stream())


def lstm_stack2_fwd_f16_fp8lo(x16, wT0_16, wT0_8, bias0, wT1_16, wT1_8, bias1, lens, T, M, Kin, H, zx_ws, h0_rows, h1_rows, hbuf0, hbuf1, S,
                              gates=(None, None), c_all=(None, None), x_segments=1, w8_scale_exp=FP8_W_SCALE_EXP, h_lo=False):
    """lstm_stack2_fwd_f16 with the low-order halves of layer 0's recurrent weights and of layer 1's weights as e4m3 operands behind the f16
    stages of the same launches (evc_lstm_stack2_fwd_f16_fp8lo).  wT0_16 [4H][x_segments Kin + H], wT0_8 [4H][H], wT1_16 / wT1_8 [4H][2H];
    h*_rows [(T+1)][M][3H/2] f16 containers = rows [f16(h) | e4m3(h 2^7)]; hbuf* [(T+1)][M][H] bf16; S [M][4H] f32.
    h_lo: rows [f16(h) | e4m3(h 2^7) | e4m3((h - f16(h)) 2^18)] (2H containers), wT0_8 [4H][2H] = [lo(Wh0) | hi(Wh0)], wT1_8 [4H][4H] = [lo(Wx1) | hi(Wx1)
    | lo(Wh1) | hi(Wh1)] (cast_fp8_lo(hi_tail=True)): the activations' low-order halves corrected as well."""
    k8 = 2 if h_lo else 1
    assert x16.dtype == F16 and wT0_16.dtype == F16 and wT1_16.dtype == F16 and wT0_8.dtype == torch.uint8 and wT1_8.dtype == torch.uint8
    assert x16.shape[-1] == x_segments * Kin and wT0_16.shape == (4 * H, x_segments * Kin + H) and wT0_8.shape == (4 * H, k8 * H)
    assert wT1_16.shape == (4 * H, 2 * H) and wT1_8.shape == (4 * H, 2 * k8 * H) and h0_rows.dtype == F16 and hbuf0.dtype == BF16
    assert h0_rows.shape[-1] == h1_rows.shape[-1] == (2 * H if h_lo else 3 * H // 2)
    assert all(t.is_contiguous() for t in (wT0_16, wT0_8, wT1_16, wT1_8))
    _lib.call("evc_lstm_stack2_fwd_f16_fp8lo", _p(x16), x_segments, _p(wT0_16), _p(wT0_8), _p(bias0), _p(wT1_16), _p(wT1_8), w8_scale_exp, 1 if h_lo else 0, _p(bias1),
              _p(lens), T, M, Kin, H, _p(zx_ws), _p(h0_rows), _p(h1_rows), _p(hbuf0), _p(hbuf1),
              _p(S[:, 0:]), _p(S[:, H:]), _p(S[:, 2 * H:]), _p(S[:, 3 * H:]), S.stride(0),
              _p(gates[0]), _p(c_all[0]), _p(gates[1]), _p(c_all[1]), _stream())


def lstm_layer_fwd_hp(x_lohi, wx_hilo, wh_hilo, bias, lens, T, M, Kin, H, zx_ws, hbuf, hbuf_lohi, c_state, h_state, ld_state,
                      gates=None, c_all=None):
    """Split-bf16 layer for the M ~ batch stacks (evc_lstm_layer_fwd_hp): wide [lo | hi] activations, [hi | lo] weights."""
    _lib.call("evc_lstm_layer_fwd_hp", _p(x_lohi), _p(wx_hilo), wx_hilo.stride(0), _p(wh_hilo), wh_hilo.stride(0), _p(bias), _p(lens),
              T, M, Kin, H, _p(zx_ws), _p(hbuf), _p(hbuf_lohi), _p(c_state), _p(h_state), ld_state, _p(gates), _p(c_all), _stream())


def lstm_layer_bwd(w_il, lens, T, M, Kin, H, gates, c_all, dS_c, dS_h, ld_dS, dh_above, dc_ws, dz4, plan=None, db=None,
                   dz_above=None, w_above=None):
    """db [4H] f32: the bias gradient is accumulated into it (zero it first) - no separate column-sum pass.
    dz_above / w_above (instead of dh_above): the upper layer's gate gradients and backward-layout kernel - its dX is
    contracted inside this layer's steps."""
    _lib.call("evc_lstm_layer_bwd", _p(w_il), _p(lens), T, M, Kin, H, _p(gates), _p(c_all), _p(dS_c), _p(dS_h), ld_dS,
              _p(dh_above), _p(dc_ws), _p(dz4), _p(db), *_plan_args(plan), _p(dz_above), _p(w_above), _stream())


def lstm_stack2_bwd(w_il0, w_il1, lens, T, M, Kin0, H, gates, c_all, dS, dc_ws, dz, db, plan=None):
    """Two-layer stack, BPTT in wavefront order (evc_lstm_stack2_bwd).  gates / c_all / dc_ws / dz / db: per-layer pairs."""
    _lib.call("evc_lstm_stack2_bwd", _p(w_il0), _p(w_il1), _p(lens), T, M, Kin0, H, _p(gates[0]), _p(c_all[0]), _p(gates[1]),
              _p(c_all[1]), _p(dS), dS.stride(0), _p(dc_ws[0]), _p(dc_ws[1]), _p(dz[0]), _p(dz[1]), _p(db[0]), _p(db[1]),
              *_plan_args(plan), _stream())


# ---------------------------------------------------------------------------
def moe_tail_fwd(gate_logits, expert_logits, B, V, M, pred, rowsum):
    _lib.call("evc_moe_tail_fwd", _p(gate_logits), _p(expert_logits), B, V, M, _p(pred), _p(rowsum), _stream())


def moe_tail_bwd(gate_logits, expert_logits, dpred, B, V, M, dgate, dexpert):
    _lib.call("evc_moe_tail_bwd", _p(gate_logits), _p(expert_logits), _p(dpred), B, V, M, _p(dgate), dgate.stride(0),
              _p(dexpert), dexpert.stride(0), _stream())


def moe_grad_update(dlogits, x, rows, V, K, p, m, v, p_bf16, pT_bf16, l2_coeff, sums, partial_ws, clip_norm, lr_t,
                    beta1=0.9, beta2=0.999, eps=1e-8, phase=0, p_wide=None, p_f16=None, p_fp8=None):
    """Fused weight-gradient + per-tensor clip + TF-Adam of one MoE weight matrix (evc_moe_grad_update); phase 1 / 2:
    the norm pass / the update pass alone, for a row slab of a matrix sharded over ranks (evc_moe_grad_update_phase).
    p_wide [V][2K] bf16 / p_f16 [V][K] f16 + p_fp8 [V][2K] uint8 (phase 0 only): also receive the forward operand images of the new weights
    (evc_moe_grad_update_wide: the wide [hi | lo] split image / the f16 image and [e4m3(W_lo) | e4m3(W)])."""
    if p_wide is not None or p_f16 is not None:
        assert phase == 0 and (p_wide is None or (p_wide.dtype == BF16 and p_wide.shape == (V, 2 * K) and p_wide.is_contiguous()))
        assert (p_f16 is None) == (p_fp8 is None)
        if p_f16 is not None:
            assert p_f16.dtype == F16 and p_f16.shape == (V, K) and p_fp8.dtype == torch.uint8 and p_fp8.shape == (V, 2 * K) and p_f16.is_contiguous() and p_fp8.is_contiguous()
        _lib.call("evc_moe_grad_update_wide", _p(dlogits), dlogits.stride(0), _p(x), x.stride(0), rows, V, K, _p(p), _p(m), _p(v),
                  _p(p_bf16), _p(pT_bf16), pT_bf16.stride(0), _p(p_wide), _p(p_f16), _p(p_fp8), FP8_MOE["w_lo_exp"], FP8_MOE["w_hi_exp"],
                  l2_coeff, _p(sums), _p(partial_ws), clip_norm, lr_t, beta1, beta2, eps, _stream())
        return
    _lib.call("evc_moe_grad_update_phase", _p(dlogits), dlogits.stride(0), _p(x), x.stride(0), rows, V, K, _p(p), _p(m), _p(v),
              _p(p_bf16), _p(pT_bf16), pT_bf16.stride(0), l2_coeff, _p(sums), _p(partial_ws), clip_norm, lr_t, beta1, beta2, eps,
              phase, _stream())


def gram_slabs(A, R, Kc, S, slabs):
    """slabs[s] [R][R] f32 = A[:, slab s] . A[:, slab s]^T over S slabs of the first Kc columns of A (bf16, evc_gram_slabs)."""
    assert A.dtype == BF16 and slabs.dtype == F32 and slabs.numel() >= S * R * R
    _lib.call("evc_gram_slabs", _p(A), A.stride(0), R, Kc, S, _p(slabs), _stream())


def moe_grad_norms(gram_a, SA, gram_x, SX, R, dlogits, logits, bias, B, V, l2_coeff, wsq, part_ws, sums):
    """sums[0] += |dlogits^T x + l2 W|^2 from the two Gram matrices, the forward logits and the carried |W|^2 (evc_moe_grad_norms);
    sums[1] += |W|^2."""
    assert part_ws.numel() >= 256 + 4 * B and logits.dtype == F32
    _lib.call("evc_moe_grad_norms", _p(gram_a), SA, _p(gram_x), SX, R, _p(dlogits), dlogits.stride(0), _p(logits), logits.stride(0),
              _p(bias), B, V, l2_coeff, _p(wsq), _p(part_ws), _p(sums), _stream())


def moe_grad_update_apply(dlogits, x, rows, V, K, p, m, v, p_bf16, pT_bf16, l2_coeff, sums, partial_ws, clip_norm, lr_t, wsq_out,
                          beta1=0.9, beta2=0.999, eps=1e-8, p_wide=None, p_f16=None, p_fp8=None):
    """The update pass of evc_moe_grad_update alone (clip scale from sums[0]) + wsq_out[0] = sum of the new weights squared."""
    assert (p_f16 is None) == (p_fp8 is None)
    _lib.call("evc_moe_grad_update_apply", _p(dlogits), dlogits.stride(0), _p(x), x.stride(0), rows, V, K, _p(p), _p(m), _p(v),
              _p(p_bf16), _p(pT_bf16), pT_bf16.stride(0), _p(p_wide), _p(p_f16), _p(p_fp8), FP8_MOE["w_lo_exp"], FP8_MOE["w_hi_exp"],
              l2_coeff, _p(sums), _p(partial_ws), clip_norm, lr_t, beta1, beta2, eps, _p(wsq_out), _stream())


def ce_loss(pred, labels_u8, loss, dpred=None, grad_scale=1.0, accumulate_grad=False):
    B, V = pred.shape
    if DETERMINISTIC:      # fixed-order loss sum from per-block partials (scratch from the stream-aware caching allocator)
        ws = torch.empty(256, dtype=F32, device=pred.device)
        _lib.call("evc_ce_loss_ordered", _p(pred), _p(labels_u8), B, V, grad_scale, _p(loss), _p(dpred), 1 if accumulate_grad else 0, _p(ws), _stream())
        return
    _lib.call("evc_ce_loss", _p(pred), _p(labels_u8), B, V, grad_scale, _p(loss), _p(dpred), 1 if accumulate_grad else 0, _stream())


def kl_pred_loss(pred_t, rowsum_t, pred_s, rowsum_s, loss, dpred_s=None, grad_scale=1.0, accumulate_grad=False):
    B, V = pred_t.shape
    _lib.call("evc_kl_pred_loss", _p(pred_t), _p(rowsum_t), _p(pred_s), _p(rowsum_s), B, V, grad_scale, _p(loss), _p(dpred_s),
              1 if accumulate_grad else 0, _stream())


def rep_loss(state_t, state_s, loss, dstate_s=None, grad_scale=1.0, accumulate_grad=False):
    B, D = state_t.shape
    if DETERMINISTIC:
        ws = torch.empty(256, dtype=F32, device=state_t.device)
        _lib.call("evc_rep_loss_ordered", _p(state_t), _p(state_s), B, D, grad_scale, _p(loss), _p(dstate_s), 1 if accumulate_grad else 0, _p(ws), _stream())
        return
    _lib.call("evc_rep_loss", _p(state_t), _p(state_s), B, D, grad_scale, _p(loss), _p(dstate_s), 1 if accumulate_grad else 0, _stream())


def clip_adam_small(ps, gs, ms, vs, sums, clip_norm, lr_t, beta1=0.9, beta2=0.999, eps=1e-8):
    """Per-tensor clip + TF-Adam of up to 16 small tensors (no l2 term) in one launch (evc_clip_adam_small): sums[i] receives {|g_i|^2, 0}."""
    import ctypes as C
    k = len(ps)
    assert 1 <= k <= 16 and len(gs) == len(ms) == len(vs) == len(sums) == k
    arr = lambda ts: (C.c_void_p * k)(*[_p(t) for t in ts])
    n = (C.c_int64 * k)(*[t.numel() for t in ps])
    _lib.call("evc_clip_adam_small", k, arr(ps), arr(gs), arr(ms), arr(vs), n, arr(sums), clip_norm, lr_t, beta1, beta2, eps, _stream())


def grad_sqnorm(g, p, l2_coeff, sums):
    _lib.call("evc_grad_sqnorm", _p(g), _p(p), l2_coeff, g.numel(), _p(sums), _stream())


def lstm_adam_fused(p, g, m, v, pb, gb, mb, vb, part_ws, sums_w, sums_b, clip_norm, lr_t, p_bf16, pT_bf16, beta1=0.9, beta2=0.999, eps=1e-8,
                    p_f16=None, nin=0, nseg=1, p_fp8=None, fp8_col0=0, fp8_hi_cols=0, fp8_lo_exp=FP8_W_SCALE_EXP, fp8_hi_exp=FP8_WX_HI_EXP, fp8_hi_tail=False):
    """Clip + TF-Adam of one LSTM layer's kernel p [4H][C] and bias pb [4H] with every operand image of the new kernel written from the same
    pass (evc_sqnorm2_partials + evc_lstm_adam_fused): bf16 forward shadow, gate-interleaved transposed bf16 backward shadow, and in "high"
    precision the f16 image (cast_f16 / cast_f16_wide without h_ext) and the e4m3 low-order image (cast_fp8_lo of the columns from fp8_col0)."""
    R, C = p.shape
    assert R % 64 == 0 and p.is_contiguous() and g.is_contiguous() and part_ws.numel() >= 1025 and pT_bf16.shape[0] == C
    assert p_f16 is None or (p_f16.dtype == F16 and p_f16.shape == (R, nseg * nin + (C - nin)) and p_f16.is_contiguous())
    assert p_fp8 is None or (p_fp8.dtype == torch.uint8 and p_fp8.shape == (R, 2 * (C - fp8_col0) if fp8_hi_tail else C - fp8_col0 + fp8_hi_cols) and p_fp8.is_contiguous())
    _lib.call("evc_sqnorm2_partials", _p(g), g.numel(), _p(gb), gb.numel(), _p(part_ws), _stream())
    _lib.call("evc_lstm_adam_fused", _p(p), _p(g), _p(m), _p(v), _p(pb), _p(gb), _p(mb), _p(vb), R // 4, C, _p(part_ws), _p(sums_w), _p(sums_b),
              clip_norm, lr_t, beta1, beta2, eps, _p(p_bf16), _p(pT_bf16), pT_bf16.stride(0), _p(p_f16), p_f16.stride(0) if p_f16 is not None else 0,
              nin, nseg, _p(p_fp8), p_fp8.stride(0) if p_fp8 is not None else 0, fp8_col0, fp8_hi_cols, fp8_lo_exp, fp8_hi_exp, 1 if fp8_hi_tail else 0, _stream())


def adam2d_fused(p, g, m, v, part_ws, sums_w, clip_norm, lr_t, p_bf16, pT_bf16, beta1=0.9, beta2=0.999, eps=1e-8,
                 p_f16=None, p_fp8=None, fp8_hi_cols=0, fp8_lo_exp=FP8_W_SCALE_EXP, fp8_hi_exp=FP8_WX_HI_EXP):
    """lstm_adam_fused for a plain 2-D weight p [R][C] without a bias (evc_sqnorm2_partials + evc_adam2d_fused): clip + TF-Adam, bf16 forward shadow,
    transposed bf16 backward shadow (pad columns zeroed) and the optional f16 / e4m3 images, one pass over the weights."""
    R, C = p.shape
    assert p.is_contiguous() and g.is_contiguous() and part_ws.numel() >= 1025 and pT_bf16.shape[0] == C and pT_bf16.stride(0) >= round_up(R, 64)
    assert p_f16 is None or (p_f16.dtype == F16 and p_f16.shape == (R, C) and p_f16.is_contiguous())
    assert p_fp8 is None or (p_fp8.dtype == torch.uint8 and p_fp8.shape == (R, C + fp8_hi_cols) and p_fp8.is_contiguous())
    _lib.call("evc_sqnorm2_partials", _p(g), g.numel(), None, 0, _p(part_ws), _stream())
    _lib.call("evc_adam2d_fused", _p(p), _p(g), _p(m), _p(v), R, C, _p(part_ws), _p(sums_w), clip_norm, lr_t, beta1, beta2, eps, _p(p_bf16), _p(pT_bf16),
              pT_bf16.stride(0), _p(p_f16), C if p_f16 is not None else 0, _p(p_fp8), p_fp8.stride(0) if p_fp8 is not None else 0, fp8_hi_cols,
              fp8_lo_exp, fp8_hi_exp, _stream())


def clip_adam_step(p, g, m, v, l2_coeff, sums, clip_norm, lr_t, beta1=0.9, beta2=0.999, eps=1e-8, p_bf16=None):
    _lib.call("evc_clip_adam_step", _p(p), _p(g), _p(m), _p(v), p.numel(), l2_coeff, _p(sums), clip_norm, lr_t, beta1, beta2, eps,
              _p(p_bf16), _stream())


def meanpool(x, num_frames, avg_f32, avg_bf16=None, normalize=False):
    B, T, F = x.shape
    is_u8 = x.dtype == torch.uint8
    _lib.call("evc_meanpool_fwd", None if is_u8 else _p(x), _p(x) if is_u8 else None, _p(num_frames), B, T, F, 1 if normalize else 0, _p(avg_f32), _p(avg_bf16), _stream())


def sigmoid_(z):
    _lib.call("evc_sigmoid_fwd", _p(z), z.numel(), _stream())
    return z


def sigmoid_bwd(p, dp, dz):
    _lib.call("evc_sigmoid_bwd", _p(p), _p(dp), p.numel(), _p(dz), _stream())


def sample_frames_gather(x, u, num_frames, out, idx_out=None, normalize=False):
    B, T, F = x.shape
    S = u.shape[1]
    is_u8 = x.dtype == torch.uint8
    _lib.call("evc_sample_frames_gather", None if is_u8 else _p(x), _p(x) if is_u8 else None, _p(u), _p(num_frames), B, T, F, S,
              1 if normalize else 0, _p(out), _p(idx_out), _stream())


def sample_sequence_gather(x, u, num_frames, S, out, idx_out=None, normalize=False):
    """SampleRandomSequence (cs/model_utils.py:11-36): u [B] one draw per video, S consecutive frames."""
    B, T, F = x.shape
    is_u8 = x.dtype == torch.uint8
    _lib.call("evc_sample_sequence_gather", None if is_u8 else _p(x), _p(x) if is_u8 else None, _p(u), _p(num_frames), B, T, F, S,
              1 if normalize else 0, _p(out), _p(idx_out), _stream())


def relu6_fwd(x, y_f32=None, y_bf16=None):
    _lib.call("evc_relu6_fwd", _p(x), x.numel(), _p(y_f32), _p(y_bf16), _stream())


def relu6_bwd(x, dy, dx_f32=None, dx_bf16=None):
    _lib.call("evc_relu6_bwd", _p(x), _p(dy), x.numel(), _p(dx_f32), _p(dx_bf16), _stream())


def framepool_mean_fwd(y, B, S, Cc, pooled_f32=None, pooled_bf16=None):
    _lib.call("evc_framepool_mean_fwd", _p(y), B, S, Cc, _p(pooled_f32), _p(pooled_bf16), _stream())


def framepool_mean_bwd(dpooled, B, S, Cc, dy):
    _lib.call("evc_framepool_mean_bwd", _p(dpooled), B, S, Cc, _p(dy), _stream())


def bn_stats(x, R, Cc, ws, mean, var):
    _lib.call("evc_bn_stats", _p(x), R, Cc, _p(ws), _p(mean), _p(var), _stream())


def bn_stats_partial(x, R, Cc, ws):
    _lib.call("evc_bn_stats_partial", _p(x), R, Cc, _p(ws), _stream())


def bn_stats_finalize(ws, R_total, Cc, mean, var):
    _lib.call("evc_bn_stats_finalize", _p(ws), R_total, Cc, _p(mean), _p(var), _stream())


def ema_update(moving, batch_value, decay=0.999):
    _lib.call("evc_ema_update", _p(moving), _p(batch_value), decay, moving.numel(), _stream())


def bn_apply(x, R, Cc, mean, var, gamma, beta, relu6, y_f32=None, y_bf16=None):
    _lib.call("evc_bn_apply", _p(x), R, Cc, _p(mean), _p(var), _p(gamma), _p(beta), 1 if relu6 else 0, _p(y_f32), _p(y_bf16), _stream())


def bn_bwd_partial(x, dy, R, Cc, mean, var, gamma, beta, relu6, ws, argmax=None, S=1):
    _lib.call("evc_bn_bwd_partial", _p(x), _p(dy), R, Cc, _p(mean), _p(var), _p(gamma), _p(beta), 1 if relu6 else 0,
              _p(argmax), S, _p(ws), _stream())


def bn_bwd_finalize(x, dy, R, R_total, Cc, mean, var, gamma, beta, relu6, ws, argmax=None, S=1,
                    dx_f32=None, dx_bf16=None, dgamma=None, dbeta=None):
    _lib.call("evc_bn_bwd_finalize", _p(x), _p(dy), R, R_total, Cc, _p(mean), _p(var), _p(gamma), _p(beta),
              1 if relu6 else 0, _p(argmax), S, _p(ws), _p(dx_f32), _p(dx_bf16), _p(dgamma), _p(dbeta), _stream())


def bn_relu6_framepool_fwd(act, B, S, Cc, mean, var, gamma, beta, pooled_f32, pooled_bf16, argmax):
    _lib.call("evc_bn_relu6_framepool_fwd", _p(act), B, S, Cc, _p(mean), _p(var), _p(gamma), _p(beta), _p(pooled_f32),
              _p(pooled_bf16), _p(argmax), _stream())


def framepool_max_fwd(y, B, S, Cc, pooled_f32, pooled_bf16, argmax):
    _lib.call("evc_framepool_max_fwd", _p(y), B, S, Cc, _p(pooled_f32), _p(pooled_bf16), _p(argmax), _stream())


def framepool_max_bwd(dpooled, argmax, B, S, Cc, dy):
    _lib.call("evc_framepool_max_bwd", _p(dpooled), _p(argmax), B, S, Cc, _p(dy), _stream())


# ---------------------------------------------------------------------------
# DbofModel fused path (csrc/evc_dbof.hip)
def dbof_workspace(B, S):
    """(padded_rows, gather_part_rows, gemm_part_rows) of the padded frame layout for B videos x S sampled frames."""
    a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
    _lib.call("evc_dbof_workspace", B, S, C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def dbof_row_index(B, S, device=None):
    """int64 [B, S]: row of frame s of video b in the padded frame layout (tests / debugging; the kernels compute it)."""
    b = torch.arange(B, device=device)[:, None]
    s = torch.arange(S, device=device)[None, :]
    return (b >> 2) * 128 + (s >> 2) * 16 + (b & 3) * 4 + (s & 3)


def dbof_gather(x, u, num_frames, r, idx_out=None, part=None, normalize=True):
    B, T, F = x.shape
    S = u.shape[1]
    is_u8 = x.dtype == torch.uint8
    assert is_u8 or x.dtype == F32
    _lib.call("evc_dbof_gather", None if is_u8 else _p(x), _p(x) if is_u8 else None, _p(u), _p(num_frames), B, T, F, S,
              1 if normalize else 0, _p(r), _p(idx_out), _p(part), _stream())


def bn_partials_reduce(part, P, Cc, ws):
    _lib.call("evc_bn_partials_reduce", _p(part), P, Cc, _p(ws), _stream())


def bn_finalize_ema(ws, R_total, Cc, mean, var, moving_mean=None, moving_var=None, decay=0.999):
    _lib.call("evc_bn_finalize_ema", _p(ws), R_total, Cc, _p(mean), _p(var), _p(moving_mean), _p(moving_var), decay, _stream())


def dbof_input_bn_apply(r, B, S, F, mean, var, gamma, beta, r_bn, r_bn_lo=None, xhat=None):
    _lib.call("evc_dbof_input_bn_apply", _p(r), B, S, F, _p(mean), _p(var), _p(gamma), _p(beta), _p(r_bn), _p(r_bn_lo), _p(xhat),
              _stream())


def dbof_cluster_pool_fwd(r_bn, wT, B, S, F, Cc, gamma, xsel, arg, act=None, part=None, r_bn_lo=None, wT_lo=None):
    _lib.call("evc_dbof_cluster_pool_fwd", _p(r_bn), _p(r_bn_lo), _p(wT), _p(wT_lo), B, S, F, Cc, _p(gamma), _p(act), _p(part),
              _p(xsel), _p(arg), _stream())


FP8_DBOF_CLUSTER = dict(x_hi_exp=5, x_lo_exp=16, w_lo_exp=19, w_hi_exp=8)   # batch-normalised frames |y| < 14, cluster weights |W| < 1.75: 5 + 19 = 16 + 8 = 24


def dbof_input_bn_apply_f16fp8(r, B, S, F, mean, var, gamma, beta, r_rows, xhat=None, e=FP8_DBOF_CLUSTER):
    """dbof_input_bn_apply writing rows [f16(y) | e4m3(y 2^x_hi_exp) | e4m3((y - f16(y)) 2^x_lo_exp)] (r_rows [Mp][2F] f16 containers)."""
    assert r_rows.dtype == F16 and r_rows.shape[1] == 2 * F and r_rows.is_contiguous()
    _lib.call("evc_dbof_input_bn_apply_f16fp8", _p(r), B, S, F, _p(mean), _p(var), _p(gamma), _p(beta), _p(r_rows), e["x_hi_exp"], e["x_lo_exp"],
              _p(xhat), _stream())


def dbof_cluster_pool_fwd_f16fp8(r_rows, wT16, wT8, B, S, F, Cc, gamma, xsel, arg, act=None, part=None, e=FP8_DBOF_CLUSTER):
    """dbof_cluster_pool_fwd on f16 + e4m3 operands (evc_dbof_cluster_pool_fwd_f16fp8): wT16 [C][F] f16, wT8 [C][2F] uint8 from
    cast_fp8_lo(W, hi_cols=F, scale_exp=w_lo_exp, hi_exp=w_hi_exp)."""
    assert r_rows.dtype == F16 and wT16.dtype == F16 and wT16.shape == (Cc, F) and wT8.dtype == torch.uint8 and wT8.shape == (Cc, 2 * F)
    assert e["x_hi_exp"] + e["w_lo_exp"] == e["x_lo_exp"] + e["w_hi_exp"]
    _lib.call("evc_dbof_cluster_pool_fwd_f16fp8", _p(r_rows), _p(wT16), _p(wT8), -(e["x_hi_exp"] + e["w_lo_exp"]), B, S, F, Cc, _p(gamma),
              _p(act), _p(part), _p(xsel), _p(arg), _stream())


def dbof_pool_finish(xsel, B, Cc, mean, var, gamma, beta, pooled_f32, pooled_bf16=None, pooled_lo=None):
    _lib.call("evc_dbof_pool_finish", _p(xsel), B, Cc, _p(mean), _p(var), _p(gamma), _p(beta), _p(pooled_f32), _p(pooled_bf16),
              _p(pooled_lo), _stream())


def dbof_dact(act, dpooled, pooled, arg, mean, var, gamma, ws, R_total, B, S, Cc, dgamma=None, dbeta=None):
    _lib.call("evc_dbof_dact", _p(act), _p(dpooled), _p(pooled), _p(arg), _p(mean), _p(var), _p(gamma), _p(ws), R_total, B, S, Cc,
              _p(dgamma), _p(dbeta), _stream())


def gemm_tn_slabs(A, B, M, N, K, slabs, nslab):
    """slabs[s] [M,N] f32 = partial product of A[K,M]^T @ B[K,N] over the s-th K range."""
    assert A.dtype == BF16 and B.dtype == BF16 and slabs.dtype == F32
    _lib.call("evc_gemm_tn_slabs", _p(A), A.stride(0), _p(B), B.stride(0), _p(slabs), M, N, K, nslab, _stream())


def dbof_wgrad_finish(slabs, nslab, Cc, F, W, gamma_in, dW, dgamma_in, dbeta_in=None, part_ws=None):
    if part_ws is None:
        part_ws = torch.empty(((Cc + 7) // 8, F), dtype=F32, device=dW.device)
    _lib.call("evc_dbof_wgrad_finish", _p(slabs), nslab, Cc, F, _p(W), _p(gamma_in), _p(dW), _p(dgamma_in), _p(dbeta_in), _p(part_ws),
              _stream())


# ---------------------------------------------------------------------------
# NetVLAD aggregation (csrc/evc_netvlad.hip; extension, see towers.NetVladTower)
def netvlad_softmax_fwd(act, R, K, mean, var, gamma, beta, a):
    _lib.call("evc_netvlad_softmax_fwd", _p(act), R, K, _p(mean), _p(var), _p(gamma), _p(beta), _p(a), _stream())


def netvlad_softmax_bwd(a, da, R, K, dz):
    _lib.call("evc_netvlad_softmax_bwd", _p(a), _p(da), R, K, _p(dz), _stream())


def netvlad_aggregate_fwd(a, r, B, S, K, F, mean, var, gamma, beta, c2, V, asum):
    _lib.call("evc_netvlad_aggregate_fwd", _p(a), _p(r), B, S, K, F, _p(mean), _p(var), _p(gamma), _p(beta), _p(c2), _p(V), _p(asum), _stream())


def netvlad_aggregate_bwd(a, r, B, S, K, F, mean, var, gamma, beta, c2, dV, da, dx):
    _lib.call("evc_netvlad_aggregate_bwd", _p(a), _p(r), B, S, K, F, _p(mean), _p(var), _p(gamma), _p(beta), _p(c2), _p(dV), _p(da), _p(dx),
              _stream())


def netvlad_dcenters(asum, dV, B, K, F, dc2):
    _lib.call("evc_netvlad_dcenters", _p(asum), _p(dV), B, K, F, _p(dc2), _stream())


def netvlad_normalize_fwd(V, B, K, F, n1, n2, Y_bf16, Y_f32=None):
    _lib.call("evc_netvlad_normalize_fwd", _p(V), B, K, F, _p(n1), _p(n2), _p(Y_f32), _p(Y_bf16), _stream())


def netvlad_normalize_bwd(V, n1, n2, dY, B, K, F, dV):
    _lib.call("evc_netvlad_normalize_bwd", _p(V), _p(n1), _p(n2), _p(dY), B, K, F, _p(dV), _stream())
