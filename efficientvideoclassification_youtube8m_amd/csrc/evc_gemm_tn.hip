// TN products (weight gradients without transposes) and the fused MoE weight update (4.4b).
#include "gemm_shared.h"

// ===========================================================================
// TN GEMM: C[M,N] (+)= A^T . B with A [K][lda], B [K][ldb] (weight gradients without transposes)
// ===========================================================================
struct StoreParamsT {
  float* C; long ldc; int M, N;
  int row_il_H;                   // > 0: row m = u*4+g of the product is stored at row g*H+u (gate de-interleave)
  int accumulate, splits, ksteps_per_split;
  long slab_stride;               // > 0: split s stores its partial tile plainly at C + s*slab_stride (no atomics; the caller sums the slabs)
};

template <class Cfg, bool SEG = false>
__global__ __launch_bounds__(Cfg::NT) void gemm_tn_kernel(GemmOperandsT p, StoreParamsT s, int tiles_m, int tiles_n) {
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x, split = 0;
  if (s.splits > 1) {
    split = bid / nwg;
    bid -= split * nwg;
  }
  const int id = xcd_remap(bid, nwg);
  int tm, tn;
  tile_of(id, tiles_m, tiles_n, tm, tn, s.splits > 1 ? patch_rows(nwg, tiles_n) : 8);
  const int m0 = tm * Cfg::BM;
  int n0 = tn * Cfg::BU;
  int nb = n0;                                                 // first column within the B segment this workgroup reads
  if (p.B2) {                                                  // two column segments (workgroup-uniform choice)
    if (n0 >= p.N1) {
      p.B = p.B2; p.ldb = p.ldb2; nb = n0 - p.N1; p.N = s.N - p.N1;
      s.N = p.c_col2 + p.N;                                    // the segment's columns in C: [c_col2, c_col2 + N2)
      n0 = p.c_col2 + nb;
    } else {
      p.N = s.N = p.N1;
    }
  }
  if (SEG) {                                                   // K steps are counted over the LIVE steps of the segments (evc_gemm_tn2_rows)
    int off = split * s.ksteps_per_split, sg = 0;
    p.nk = min(s.ksteps_per_split, p.nk - off);
#pragma unroll
    for (int i = 0; i < 15; ++i) {                             // (constant indices: the table stays in scalar registers)
      const bool adv = sg == i && i + 1 < p.nseg && off >= (int)p.seg_len[i];
      off -= adv ? (int)p.seg_len[i] : 0;
      sg += adv ? 1 : 0;
    }
    p.seg0 = sg; p.seg_off0 = off;
  } else if (s.splits > 1) {
    const int k0 = split * s.ksteps_per_split;
    p.A += (long)k0 * 32 * p.lda;
    p.B += (long)k0 * 32 * p.ldb;
    p.nk = min(s.ksteps_per_split, p.nk - k0);
  }
  f32x4 acc[Cfg::MI][1][Cfg::NI];
  gemm_mainloop_tn<Cfg, true, EVC_TN_LOOP_MODE, SEG>(p, m0, nb, lds_dyn, acc);      // transposed accumulators: lane = one row, 4 consecutive columns
  // Through the per-wave LDS transpose (store_tile_via_lds): whole sub-tile rows for the plain / slab stores, contiguous
  // row runs for the split-K atomics ("accumulate" is the same join onto what C already holds).
  float* C = s.C + split * s.slab_stride;
  const int wave = threadIdx.x >> 6, wc = wave % Cfg::WC;
  const bool plain = s.slab_stride > 0 || (s.splits == 1 && !s.accumulate);
  const bool aligned = (s.ldc % 4) == 0 && ((uintptr_t)C % 16) == 0 && n0 + wc * Cfg::WU + Cfg::WU <= s.N;
  __syncthreads();                                             // every wave has read its last ring slot
#ifdef EVC_ABLATE_TN_ATOMICS     // debug build: plain stores instead of the split-K atomics (wrong sums, timing only)
  store_tile_via_lds<Cfg, 4, false>(acc, lds_dyn, C, s.ldc, s.M, s.N, m0, n0, nullptr, s.row_il_H);
#else
  if (plain && aligned) {
    store_tile_via_lds<Cfg, 4, false>(acc, lds_dyn, C, s.ldc, s.M, s.N, m0, n0, nullptr, s.row_il_H);
  } else if (s.splits == 1 && s.accumulate && aligned) {     // one workgroup per tile: C += tile needs no atomics
    store_tile_via_lds<Cfg, 4, false, true>(acc, lds_dyn, C, s.ldc, s.M, s.N, m0, n0, nullptr, s.row_il_H);
  } else if (plain) {            // ragged right edge / unaligned rows: element-wise
    TileCoordsT<Cfg> tc;
#pragma unroll
    for (int mi = 0; mi < Cfg::MI; ++mi) {
      const int m = m0 + tc.row0 + mi * 16;
      if (m >= s.M) continue;
      const long mo = s.row_il_H > 0 ? (long)(m & 3) * s.row_il_H + (m >> 2) : m;
#pragma unroll
      for (int ni = 0; ni < Cfg::NI; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + tc.unit0 + ni * 16 + r;
          if (n < s.N) C[mo * s.ldc + n] = acc[mi][0][ni][r];
        }
    }
  } else {
    store_tile_via_lds<Cfg, 4, true>(acc, lds_dyn, C, s.ldc, s.M, s.N, m0, n0, nullptr, s.row_il_H);
  }
#endif
}

static int gemm_tn_impl(const evc_bf16* A, int64_t lda, const evc_bf16* B, int64_t ldb, int N1, const evc_bf16* B2, int64_t ldb2,
                        int c_col2, float* C, int64_t ldc, int M, int N, int K, int row_interleave_H, int accumulate, void* stream,
                        int slab_rows = 0, const int32_t* rows_per_slab = nullptr) {
  EVC_REQUIRE(M >= 8 && N >= 8 && K > 0 && M % 8 == 0 && N % 8 == 0 && K % 32 == 0, EVC_ERR_BAD_SHAPE,
              "evc_gemm_tn: needs M %% 8 == 0, N %% 8 == 0, K %% 32 == 0 (M=%d N=%d K=%d)", M, N, K);
  EVC_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, EVC_ERR_BAD_ALIGN,
              "evc_gemm_tn: operands must be 16-byte aligned (lda=%ld ldb=%ld)", (long)lda, (long)ldb);
  EVC_REQUIRE(row_interleave_H == 0 || M == 4 * row_interleave_H, EVC_ERR_BAD_SHAPE, "evc_gemm_tn: row_interleave_H needs M == 4*H");
  EVC_REQUIRE(!B2 || (N1 > 0 && N1 < N && N1 % 256 == 0 && ldb2 % 8 == 0 && ((uintptr_t)B2 % 16) == 0 && c_col2 >= N1), EVC_ERR_BAD_SHAPE,
              "evc_gemm_tn2: N1=%d must be a multiple of 256 inside (0, N=%d), B2 16-byte aligned with ldb2 %% 8 == 0, c_col2=%d >= N1", N1, N, c_col2);
  hipStream_t st = (hipStream_t)stream;
  GemmOperandsT p{A, lda, B, ldb, M, N, K / 32};
  if (B2) { p.B2 = B2; p.ldb2 = ldb2; p.N1 = N1; p.c_col2 = c_col2; }
  // live rows per time slab (evc_gemm_tn2_rows): the K walk covers ceil(rows / 32) steps of every slab.  More than 16 non-empty slabs, nothing
  // dead or nothing live: the plain walk over all K rows (the dead rows of A are zeros: the same sums).
  bool seg = false;
  if (rows_per_slab && slab_rows > 0 && slab_rows % 32 == 0 && K % slab_rows == 0 && !evc_deterministic()) {
    const int nslabs = K / slab_rows, slab_steps = slab_rows / 32;
    int n = 0, total = 0;
    // the segment walk indexes both operands by 32-bit BYTE offsets from their un-advanced bases (gemm_core_tn.h: (start + off) * 64 * ld): the
    // last K row must stay below 4 GiB in the wider operand - beyond that the plain walk (which advances the base pointers per split) runs
    const long ld_max = lda > ldb ? (long)lda : (long)ldb;
    bool ok = (long)nslabs * slab_steps < 65536 && (long)K * (B2 && ldb2 > ld_max ? (long)ldb2 : ld_max) * 2 < (1L << 32);
    for (int t = 0; t < nslabs && ok; ++t) {
      int r = rows_per_slab[t] < 0 ? 0 : (rows_per_slab[t] > slab_rows ? slab_rows : rows_per_slab[t]);
      const int steps = (r + 31) / 32;
      if (steps == 0) continue;
      if (n == 16) { ok = false; break; }
      p.seg_start[n] = (unsigned short)(t * slab_steps); p.seg_len[n] = (unsigned short)steps;
      ++n; total += steps;
    }
    if (ok && total > 0 && total < K / 32) { seg = true; p.nseg = n; p.nk = total; K = total * 32; }
  }
  typedef TileCfg2<128, 1, 128, 2, 4, 5, true> CfgTn128;
  // short contractions (the student's L2: K = 5 x 256 rows) on 128x128 tiles without split-K: the atomic join of
  // 256x256 partial tiles costs more than the product itself there (81 -> 36 us at 4096 x 1024 x 1280); from
  // K ~ 5000 on the 256x256 split-K form is faster again (92 vs 99 us)
  if (forced_tile() == 11 || (forced_tile() == 0 && K <= 2048 && (long)ceil_div(M, 128) * ceil_div(N, 128) >= 192)) {
    const int tm1 = ceil_div(M, 128), tn1 = ceil_div(N, 128);
    StoreParamsT s1{C, ldc, M, N, row_interleave_H, accumulate, 1, p.nk, 0};
    if (seg) launch_cfg<CfgTn128>(gemm_tn_kernel<CfgTn128, true>, tm1 * tn1, st, p, s1, tm1, tn1);
    else launch_cfg<CfgTn128>(gemm_tn_kernel<CfgTn128>, tm1 * tn1, st, p, s1, tm1, tn1);
    EVC_LAUNCH_CHECK();
    return EVC_OK;
  }
  // a narrow strip (N <= 128: the last 128 input columns of an L1 layer-0 kernel gradient, see engine._wgrad_tn): 128x128 tiles,
  // K split until ~256 workgroups exist - a 256-column tile would do half of its MFMAs on columns that do not exist
  if (forced_tile() == 0 && N <= 128 && !B2) {
    const int tm1 = ceil_div(M, 128);
    int splits = 256 / tm1;
    if (splits > K / 1024) splits = K / 1024;
    if (splits < 1 || evc_deterministic()) splits = 1;
    while (splits > 1 && (long)ceil_div(p.nk, splits) * (splits - 1) >= p.nk) --splits;     // no empty split
    StoreParamsT s1{C, ldc, M, N, row_interleave_H, accumulate, splits, ceil_div(p.nk, splits), 0};
    if (splits > 1 && !accumulate) EVC_CHECK_HIP(hipMemset2DAsync(C, ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, st));
    if (seg) launch_cfg<CfgTn128>(gemm_tn_kernel<CfgTn128, true>, tm1 * splits, st, p, s1, tm1, 1);
    else launch_cfg<CfgTn128>(gemm_tn_kernel<CfgTn128>, tm1 * splits, st, p, s1, tm1, 1);
    EVC_LAUNCH_CHECK();
    return EVC_OK;
  }
  const int tm = ceil_div(M, CfgPlainV2::BM), tn = ceil_div(N, CfgPlainV2::BU);
  int splits = 256 / (tm * tn);
  if (splits > K / 1024) splits = K / 1024;     // keep >= 32 K steps per split
  if (splits < 1 || evc_deterministic()) splits = 1;
  StoreParamsT s{C, ldc, M, N, row_interleave_H, accumulate, splits, ceil_div(p.nk, splits), 0};
  if (splits > 1 && !accumulate) EVC_CHECK_HIP(hipMemset2DAsync(C, ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, st));
  if (seg) launch_cfg<CfgPlainV2>(gemm_tn_kernel<CfgPlainV2, true>, tm * tn * splits, st, p, s, tm, tn);
  else launch_cfg<CfgPlainV2>(gemm_tn_kernel<CfgPlainV2>, tm * tn * splits, st, p, s, tm, tn);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// evc_gemm_tn / evc_gemm_tn2 over TIME SLABS with a live prefix (round 5): the contraction index runs over K = nslabs * slab_rows rows, slab t
// holding rows_per_slab[t] live rows at its start (row plans: rows sorted by length, the weight-gradient products dW^T = dz^T . [x | h] of a
// row-planned LSTM level) - the K walk covers ceil(rows / 32) steps of every slab and jumps over the rest (the teacher's L1 level: 5 % of
// the rows).  The skipped rows of A must be zeros or the skipped rows of A and B finite garbage that the caller does not want summed; rows
// between rows_per_slab[t] and the next multiple of 32 ARE contracted.  B2 == NULL: one column segment of N1 columns.  rows_per_slab is HOST
// memory (read before the launch).  More than 16 non-empty slabs or EVC_DETERMINISTIC: the plain product over all K rows.
extern "C" int evc_gemm_tn2_rows(const evc_bf16* A, int64_t lda, const evc_bf16* B1, int64_t ldb1, int N1, const evc_bf16* B2, int64_t ldb2,
                                 int N2, int c_col2, float* C, int64_t ldc, int M, int slab_rows, int nslabs, const int32_t* rows_per_slab,
                                 int row_interleave_H, int accumulate, void* stream) {
  EVC_REQUIRE(B1 && N1 > 0 && slab_rows > 0 && nslabs > 0 && rows_per_slab && (long)slab_rows * nslabs < (1L << 31), EVC_ERR_BAD_ARG,
              "evc_gemm_tn2_rows: B1, N1, slab_rows, nslabs, rows_per_slab are required");
  EVC_REQUIRE(!B2 || N2 > 0, EVC_ERR_BAD_ARG, "evc_gemm_tn2_rows: N2=%d", N2);
  EVC_REQUIRE(!B2 || accumulate || c_col2 == N1, EVC_ERR_BAD_ARG, "evc_gemm_tn2_rows: segments that are not adjacent in C (c_col2=%d, N1=%d) need accumulate", c_col2, N1);
  return gemm_tn_impl(A, lda, B1, ldb1, B2 ? N1 : 0, B2, ldb2, B2 ? c_col2 : 0, C, ldc, M, N1 + (B2 ? N2 : 0), slab_rows * nslabs, row_interleave_H, accumulate,
                      stream, slab_rows, rows_per_slab);
}

extern "C" int evc_gemm_tn(const evc_bf16* A, int64_t lda, const evc_bf16* B, int64_t ldb, float* C, int64_t ldc,
                           int M, int N, int K, int row_interleave_H, int accumulate, void* stream) {
  return gemm_tn_impl(A, lda, B, ldb, 0, nullptr, 0, 0, C, ldc, M, N, K, row_interleave_H, accumulate, stream);
}

extern "C" int evc_gemm_tn2(const evc_bf16* A, int64_t lda, const evc_bf16* B1, int64_t ldb1, int N1, const evc_bf16* B2, int64_t ldb2,
                            int N2, int c_col2, float* C, int64_t ldc, int M, int K, int row_interleave_H, int accumulate, void* stream) {
  EVC_REQUIRE(B1 && B2 && N1 > 0 && N2 > 0, EVC_ERR_BAD_ARG, "evc_gemm_tn2: two column segments are required");
  EVC_REQUIRE(accumulate || c_col2 == N1, EVC_ERR_BAD_ARG, "evc_gemm_tn2: segments that are not adjacent in C (c_col2=%d, N1=%d) need accumulate "
              "(the split-K join adds into a C the caller has zeroed)", c_col2, N1);
  return gemm_tn_impl(A, lda, B1, ldb1, N1, B2, ldb2, c_col2, C, ldc, M, N1 + N2, K, row_interleave_H, accumulate, stream);
}

// Split-K into slabs: slab s (s < nslab) = the partial product over K rows [s*ceil(K/32/nslab)*32, ...), stored plainly at
// slabs + s*M*N (row stride N).  For products whose 256x256 tiles do not fill the chip and whose result is read once by a
// pass that can add the slabs on the way (DBoF cluster-weight gradient: 8192 x 1152 x 16384 = 160 tiles): no atomics, no memset.
extern "C" int evc_gemm_tn_slabs(const evc_bf16* A, int64_t lda, const evc_bf16* B, int64_t ldb, float* slabs, int M, int N, int K,
                                 int nslab, void* stream) {
  EVC_REQUIRE(M >= 8 && N >= 8 && K > 0 && M % 8 == 0 && N % 8 == 0 && K % 32 == 0 && nslab >= 1 && nslab <= K / 32, EVC_ERR_BAD_SHAPE,
              "evc_gemm_tn_slabs: needs M %% 8 == 0, N %% 8 == 0, K %% 32 == 0, 1 <= nslab <= K/32 (M=%d N=%d K=%d nslab=%d)", M, N, K, nslab);
  EVC_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, EVC_ERR_BAD_ALIGN,
              "evc_gemm_tn_slabs: operands must be 16-byte aligned (lda=%ld ldb=%ld)", (long)lda, (long)ldb);
  GemmOperandsT p{A, lda, B, ldb, M, N, K / 32};
  const int tm = ceil_div(M, CfgPlainV2::BM), tn = ceil_div(N, CfgPlainV2::BU);
  const int per = ceil_div(p.nk, nslab);
  EVC_REQUIRE((long)per * (nslab - 1) < p.nk, EVC_ERR_BAD_SHAPE, "evc_gemm_tn_slabs: nslab=%d leaves an empty slab at K=%d", nslab, K);
  StoreParamsT s{slabs, N, M, N, 0, 0, nslab, per, (long)M * N};
  launch_cfg<CfgPlainV2>(gemm_tn_kernel<CfgPlainV2>, tm * tn * nslab, (hipStream_t)stream, p, s, tm, tn);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// Two column segments into slabs (EVC_DETERMINISTIC=1 weight gradients, round 5): evc_gemm_tn2's product with the K split stored as nslab plain partial
// images instead of joined with atomics - slab s at slabs + s*slab_stride, each an [M][ldc] image like C (rows de-interleaved, the second segment
// at column c_col2) - that evc_sum_slabs adds in slab order.  B2 == NULL: one segment of N1 columns.
extern "C" int evc_gemm_tn2_slabs(const evc_bf16* A, int64_t lda, const evc_bf16* B1, int64_t ldb1, int N1, const evc_bf16* B2, int64_t ldb2,
                                  int N2, int c_col2, float* slabs, int64_t ldc, int64_t slab_stride, int M, int K, int row_interleave_H,
                                  int nslab, void* stream) {
  const int N = N1 + (B2 ? N2 : 0);
  EVC_REQUIRE(M >= 8 && N1 >= 8 && K > 0 && M % 8 == 0 && N % 8 == 0 && K % 32 == 0 && nslab >= 1 && nslab <= K / 32, EVC_ERR_BAD_SHAPE,
              "evc_gemm_tn2_slabs: needs M %% 8 == 0, N %% 8 == 0, K %% 32 == 0, 1 <= nslab <= K/32 (M=%d N=%d K=%d nslab=%d)", M, N, K, nslab);
  EVC_REQUIRE(lda % 8 == 0 && ldb1 % 8 == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B1 % 16) == 0 && slabs && slab_stride >= (int64_t)M * ldc, EVC_ERR_BAD_ALIGN,
              "evc_gemm_tn2_slabs: operands must be 16-byte aligned, slab_stride >= M * ldc");
  EVC_REQUIRE(row_interleave_H == 0 || M == 4 * row_interleave_H, EVC_ERR_BAD_SHAPE, "evc_gemm_tn2_slabs: row_interleave_H needs M == 4*H");
  EVC_REQUIRE(!B2 || (N1 % 256 == 0 && N2 > 0 && ldb2 % 8 == 0 && ((uintptr_t)B2 % 16) == 0 && c_col2 >= N1), EVC_ERR_BAD_SHAPE,
              "evc_gemm_tn2_slabs: N1=%d must be a multiple of 256, B2 16-byte aligned with ldb2 %% 8 == 0, c_col2=%d >= N1", N1, c_col2);
  GemmOperandsT p{A, lda, B1, ldb1, M, N, K / 32};
  if (B2) { p.B2 = B2; p.ldb2 = ldb2; p.N1 = N1; p.c_col2 = c_col2; }
  const int tm = ceil_div(M, CfgPlainV2::BM), tn = ceil_div(N, CfgPlainV2::BU);
  const int per = ceil_div(p.nk, nslab);
  EVC_REQUIRE((long)per * (nslab - 1) < p.nk, EVC_ERR_BAD_SHAPE, "evc_gemm_tn2_slabs: nslab=%d leaves an empty slab at K=%d", nslab, K);
  StoreParamsT s{slabs, ldc, M, N, row_interleave_H, 0, nslab, per, slab_stride};
  launch_cfg<CfgPlainV2>(gemm_tn_kernel<CfgPlainV2>, tm * tn * nslab, (hipStream_t)stream, p, s, tm, tn);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// C[r][c] (+)= sum over slabs s, in slab order, of slabs[s*slab_stride + r*ld + c] for r < M, c < N (N % 4 == 0, 16-byte aligned rows)
__global__ __launch_bounds__(256) void sum_slabs_kernel(const float* __restrict__ slabs, long slab_stride, int nslab, int M, int N4, long ld,
                                                        float* __restrict__ C, long ldc, int accumulate) {
  const long n = (long)M * N4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long r = i / N4, c = (i - r * N4) * 4;
    float4 a = accumulate ? *(const float4*)(C + r * ldc + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < nslab; ++s) {
      const float4 v = *(const float4*)(slabs + s * slab_stride + r * ld + c);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *(float4*)(C + r * ldc + c) = a;
  }
}
extern "C" int evc_sum_slabs(const float* slabs, int64_t slab_stride, int nslab, int M, int N, int64_t ld, float* C, int64_t ldc, int accumulate,
                             void* stream) {
  EVC_REQUIRE(slabs && C && nslab >= 1 && M > 0 && N > 0 && N % 4 == 0 && ld % 4 == 0 && ldc % 4 == 0 && slab_stride % 4 == 0 &&
              ((uintptr_t)slabs % 16) == 0 && ((uintptr_t)C % 16) == 0, EVC_ERR_BAD_SHAPE, "evc_sum_slabs: N, ld, ldc, slab_stride multiples of 4, 16-byte aligned");
  const long n = (long)M * (N / 4);
  long nb = (n + 255) / 256;
  nb = nb > 2048 ? 2048 : nb;
  hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, slabs, (long)slab_stride, nslab, M, N / 4, (long)ld, C, (long)ldc,
                     accumulate);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// ===========================================================================
// MoE weight update without materialising the gradient.
// The gradient of a MoE weight matrix W [V][K] (stored as the forward GEMM's B operand) is the outer
// product dlogits^T . x over the batch rows: rank = batch.  Writing it (4 B/param), reading it for the
// norm (4+4) and again for Adam, then transposing the updated weights for the backward shadow costs 46
// bytes per parameter of HBM traffic for 96.6 M parameters per tower.  Here the [256 x 256] gradient tile is
// recomputed from the factors (8 K steps of the TN loop) in each of two passes:
//   pass 1: sum (g + l2 p)^2 and sum p^2 per workgroup -> partials (summed in a fixed order afterwards)
//   pass 2: per-tensor clip + TF-Adam in the epilogue: reads p, m, v, writes p, m, v, the bf16 forward
//           shadow and - through an LDS transpose - the bf16 transposed shadow: 30 bytes per parameter.
// Under data parallelism the factors of all ranks are all-gathered (14 MB per rank) instead of all-reducing
// the 386 MB gradient; the contraction then simply runs over world x batch rows.
// ===========================================================================
struct MoeUpdateParams {
  float* p; float* m; float* v;        // [V][K] f32, row stride K
  bf16_t* p_bf16;                      // forward shadow [V][K], or NULL (round 5: a "high" tower's forward reads the f16 + e4m3 images, not this one)
  bf16_t* pT_bf16; long ldT;           // transposed shadow [K][ldT], ldT >= V
  bf16_t* p_wide;                      // or NULL: wide split-bf16 image [V][2K] = [hi | lo] of the new weights (the "split" forward's operand)
  bf16_t* p_f16; uint8_t* p_fp8;       // or NULL: IEEE f16 image [V][K] and e4m3 image [V][2K] = [e4m3((w - f16(w)) lo_scale) | e4m3(w hi_scale)] of the
  float lo_scale, hi_scale;            // new weights (the "high" forward's operands: evc_gemm_nt_f16_fp8)
  float* partial;                      // pass 1 out: [workgroups][2]
  float* wsq_partial;                  // or NULL; pass 2 out: [workgroups][2] = {sum of the new weights squared, 0}
  const float* sums;                   // pass 2 in: sums[0] = sum (g + l2 p)^2 of this tensor
  int V, K;
  float l2, clip, lr_t, b1, b2, eps;
};

#ifndef EVC_ADAM_NT
#define EVC_ADAM_NT 1       // (round 5: 1 = W, m, v of the fused MoE update - read once, written once, 5.9 GB per step - as non-temporal accesses: same box, alternated three times, 10.00 -> 9.93 ms per step; 0 = plain accesses; 2 = the bf16 shadows too - they are read again soon: 9.82 -> 9.93)
#endif
__device__ __forceinline__ float4 ld_stream_moe(const float* p) {
#if EVC_ADAM_NT
  const f32x4 v = __builtin_nontemporal_load((const f32x4*)p);
  return make_float4(v[0], v[1], v[2], v[3]);
#else
  return *(const float4*)p;
#endif
}
__device__ __forceinline__ void st_stream_moe(float* p, float a, float b, float c, float d) {
#if EVC_ADAM_NT
  __builtin_nontemporal_store(f32x4{a, b, c, d}, (f32x4*)p);
#else
  *(float4*)p = make_float4(a, b, c, d);
#endif
}
#ifndef EVC_MOE_UPD_EARLY_MV
#define EVC_MOE_UPD_EARLY_MV 0          // (A/B, round 5: 1 = m and v asked for ahead of the factor product like p - 219 VGPRs, still one workgroup per CU;
                                        //  measured 459-462 / 309-316 us against 442-456 / 302-305: nothing, the pass is not waiting for those loads)
#endif
#ifndef EVC_MOE_UPD_WAVES_PER_EU
#define EVC_MOE_UPD_WAVES_PER_EU 2      // (A/B: 4 = cap the kernel at 128 VGPRs so that two of its 80 KB workgroups share a CU)
#endif
template <class Cfg, int PASS>
__global__ __launch_bounds__(Cfg::NT, EVC_MOE_UPD_WAVES_PER_EU) void moe_update_kernel(GemmOperandsT p, MoeUpdateParams u, int tiles_m, int tiles_n) {
  const int nwg = tiles_m * tiles_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  int tm, tn;
  tile_of(id, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * Cfg::BM, n0 = tn * Cfg::BU;
  f32x4 acc[Cfg::MI][1][Cfg::NI];
  TileCoordsT<Cfg> tc;
  const int K = u.K;
  // Epilogue loads first, all of them (the stores of one fragment and the loads of the next go to the same
  // arrays, so hipcc keeps them in program order and every fragment would wait for the previous one's stores:
  // 8 x (load latency + store acknowledge) per workgroup; issued up front they overlap - 3.4 -> see DESIGN.md).
  // The weights themselves are asked for BEFORE the factor product: they do not depend on it, and their HBM
  // latency then runs under the 8-step loop instead of after it.
  float4 pv[Cfg::MI][Cfg::NI];
#pragma unroll
  for (int mi = 0; mi < Cfg::MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < Cfg::NI; ++ni) {
      const int vr = m0 + tc.row0 + mi * 16, k = n0 + tc.unit0 + ni * 16;
      pv[mi][ni] = (vr < u.V && k < K) ? ld_stream_moe(u.p + (long)vr * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#if EVC_MOE_UPD_EARLY_MV
  // (A/B) m and v asked for up front too: the update pass owns its CU either way (184 -> 219 VGPRs, still one 8-wave workgroup)
  float4 mv[Cfg::MI][Cfg::NI], vv[Cfg::MI][Cfg::NI];
  if (PASS == 2) {
#pragma unroll
    for (int mi = 0; mi < Cfg::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < Cfg::NI; ++ni) {
        const int vr = m0 + tc.row0 + mi * 16, k = n0 + tc.unit0 + ni * 16;
        const bool ok = vr < u.V && k < K;
        mv[mi][ni] = ok ? *(const float4*)(u.m + (long)vr * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        vv[mi][ni] = ok ? *(const float4*)(u.v + (long)vr * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
  }
#endif
  gemm_mainloop_tn<Cfg, true>(p, m0, n0, lds_dyn, acc);
  if (PASS == 1) {
    float sg = 0.f, sp = 0.f;
#pragma unroll
    for (int mi = 0; mi < Cfg::MI; ++mi) {
      const int vr = m0 + tc.row0 + mi * 16;
#pragma unroll
      for (int ni = 0; ni < Cfg::NI; ++ni) {
        const int k = n0 + tc.unit0 + ni * 16;
        if (vr >= u.V || k >= K) continue;
        const float pa[4] = {pv[mi][ni].x, pv[mi][ni].y, pv[mi][ni].z, pv[mi][ni].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float w = acc[mi][0][ni][r] + u.l2 * pa[r];
          sg += w * w;
          sp += pa[r] * pa[r];
        }
      }
    }
    sg = wave_sum(sg);
    sp = wave_sum(sp);
    __syncthreads();                                   // the LDS ring is free now
    float* red = (float*)lds_dyn;
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave * 2] = sg; red[wave * 2 + 1] = sp; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float a = 0.f, b = 0.f;
      for (int w = 0; w < Cfg::NT / 64; ++w) { a += red[w * 2]; b += red[w * 2 + 1]; }
      u.partial[2 * blockIdx.x] = a;
      u.partial[2 * blockIdx.x + 1] = b;
    }
    return;
  }
#if !EVC_MOE_UPD_EARLY_MV
  float4 mv[Cfg::MI][Cfg::NI], vv[Cfg::MI][Cfg::NI];
#pragma unroll
  for (int mi = 0; mi < Cfg::MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < Cfg::NI; ++ni) {
      const int vr = m0 + tc.row0 + mi * 16, k = n0 + tc.unit0 + ni * 16;
      const bool ok = vr < u.V && k < K;
      mv[mi][ni] = ok ? ld_stream_moe(u.m + (long)vr * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      vv[mi][ni] = ok ? ld_stream_moe(u.v + (long)vr * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#endif
  float scale = 1.f;
  if (u.clip > 0.f) scale = u.clip / fmaxf(sqrtf(u.sums[0]), u.clip);      // tf.clip_by_norm
  float wsq = 0.f;                                     // sum of the NEW weights squared (the next update's |W|^2: evc_moe_grad_norms)
  __syncthreads();                                     // every wave is done with the ring: reuse it for the transpose
  constexpr int PITCH = Cfg::BM + 8;                   // bf16 elements per k row of the [BU k][BM v] image (+16 B: bank spread)
  bf16_t* tile = (bf16_t*)lds_dyn;
  static_assert((long)Cfg::BU * PITCH * 2 <= Cfg::LDS_BYTES, "transpose image must fit the ring");
#pragma unroll
  for (int mi = 0; mi < Cfg::MI; ++mi) {
    const int vl = tc.row0 + mi * 16, vr = m0 + vl;
#pragma unroll
    for (int ni = 0; ni < Cfg::NI; ++ni) {
      const int kl = tc.unit0 + ni * 16, k = n0 + kl;
      bf16_t pb[4] = {0, 0, 0, 0};
      if (vr < u.V && k < K) {
        const long o = (long)vr * K + k;
        const float pa[4] = {pv[mi][ni].x, pv[mi][ni].y, pv[mi][ni].z, pv[mi][ni].w};
        const float ma[4] = {mv[mi][ni].x, mv[mi][ni].y, mv[mi][ni].z, mv[mi][ni].w};
        const float va[4] = {vv[mi][ni].x, vv[mi][ni].y, vv[mi][ni].z, vv[mi][ni].w};
        float pn[4], mn[4], vn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {                 // same operation order as clip_adam_kernel
          const float gc = (acc[mi][0][ni][r] + u.l2 * pa[r]) * scale;
          mn[r] = u.b1 * ma[r] + (1.f - u.b1) * gc;
          vn[r] = u.b2 * va[r] + (1.f - u.b2) * gc * gc;
          pn[r] = adam_step_(pa[r], mn[r], vn[r], u.lr_t, u.eps);
          pb[r] = f32_to_bf16(pn[r]);
          wsq += pn[r] * pn[r];
        }
        st_stream_moe(u.p + o, pn[0], pn[1], pn[2], pn[3]);
        st_stream_moe(u.m + o, mn[0], mn[1], mn[2], mn[3]);
        st_stream_moe(u.v + o, vn[0], vn[1], vn[2], vn[3]);
        if (u.p_bf16) {
          const u32x2_t sb = {(uint32_t)pb[0] | ((uint32_t)pb[1] << 16), (uint32_t)pb[2] | ((uint32_t)pb[3] << 16)};
#if EVC_ADAM_NT >= 2
          __builtin_nontemporal_store(sb, (u32x2_t*)(u.p_bf16 + o));
#else
          *(u32x2_t*)(u.p_bf16 + o) = sb;
#endif
        }
        if (u.p_f16) {                                // f16 + e4m3 images: saves the passes over the f32 weights (evc_cast_f32_to_f16 / _fp8_lo) per update
          const uint32_t h01 = pack_f16x2_hw(pn[0], pn[1]), h23 = pack_f16x2_hw(pn[2], pn[3]);
          *(uint2*)(u.p_f16 + o) = make_uint2(h01, h23);
          const float hf[4] = {f16_to_f32((f16_t)(h01 & 0xffffu)), f16_to_f32((f16_t)(h01 >> 16)), f16_to_f32((f16_t)(h23 & 0xffffu)), f16_to_f32((f16_t)(h23 >> 16))};
          float lo8[4], hi8[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            lo8[r] = fminf(fmaxf((pn[r] - hf[r]) * u.lo_scale, -448.f), 448.f);
            hi8[r] = fminf(fmaxf(pn[r] * u.hi_scale, -448.f), 448.f);
          }
          int wl = __builtin_amdgcn_cvt_pk_fp8_f32(lo8[0], lo8[1], 0, false);
          wl = __builtin_amdgcn_cvt_pk_fp8_f32(lo8[2], lo8[3], wl, true);
          int wh = __builtin_amdgcn_cvt_pk_fp8_f32(hi8[0], hi8[1], 0, false);
          wh = __builtin_amdgcn_cvt_pk_fp8_f32(hi8[2], hi8[3], wh, true);
          uint8_t* w8 = u.p_fp8 + (long)vr * 2 * K + k;
          *(int*)w8 = wl;
          *(int*)(w8 + K) = wh;
        }
        if (u.p_wide) {                               // [hi | lo]: saves a pass over the f32 weights (evc_cast_f32_to_bf16_wide) per update
          bf16_t* w = u.p_wide + (long)vr * 2 * K + k;
          *(uint2*)w = make_uint2((uint32_t)pb[0] | ((uint32_t)pb[1] << 16), (uint32_t)pb[2] | ((uint32_t)pb[3] << 16));
          *(uint2*)(w + K) = make_uint2(pack_bf16x2_hw(pn[0] - bf16_to_f32(pb[0]), pn[1] - bf16_to_f32(pb[1])),
                                        pack_bf16x2_hw(pn[2] - bf16_to_f32(pb[2]), pn[3] - bf16_to_f32(pb[3])));
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[(kl + r) * PITCH + vl] = pb[r];
    }
  }
  __syncthreads();
  // rows k of the transposed shadow: 4 bf16 per lane, BM/4 lanes per row, 64/(BM/4) rows per wave-instruction
  constexpr int LPR = Cfg::BM / 4, RPW = 64 / LPR;
  static_assert(LPR <= 64 && 64 % LPR == 0, "row of the transposed image must fit a wave");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int v4 = m0 + (lane % LPR) * 4;
  for (int kl = wave * RPW + lane / LPR; kl < Cfg::BU; kl += (Cfg::NT / 64) * RPW) {
    const int k = n0 + kl;
    if (k >= K || v4 >= u.V) continue;                 // V % 4 == 0: a lane's 4 rows are all valid or all not
    const uint2 q = *(const uint2*)(tile + kl * PITCH + (lane % LPR) * 4);
#if EVC_ADAM_NT >= 2
    __builtin_nontemporal_store(u32x2_t{q.x, q.y}, (u32x2_t*)(u.pT_bf16 + (long)k * u.ldT + v4));
#else
    *(uint2*)(u.pT_bf16 + (long)k * u.ldT + v4) = q;
#endif
  }
  if (u.wsq_partial) {                                 // per-workgroup partial, summed in a fixed order by moe_update_finalize_kernel
    wsq = wave_sum(wsq);
    __syncthreads();                                   // the transpose image has been read
    float* red = (float*)lds_dyn;
    if ((threadIdx.x & 63) == 0) red[wave] = wsq;
    __syncthreads();
    if (threadIdx.x == 0) {
      float a = 0.f;
      for (int w = 0; w < Cfg::NT / 64; ++w) a += red[w];
      u.wsq_partial[2 * blockIdx.x] = a;
      u.wsq_partial[2 * blockIdx.x + 1] = 0.f;
    }
  }
}

__global__ __launch_bounds__(1024) void moe_update_finalize_kernel(const float* partial, int n, float* sums, int assign) {
  // one workgroup, fixed summation order (thread-strided partial sums, wave butterflies, then the 16 wave totals
  // in order): run-to-run identical
  __shared__ float wa[16], wb[16];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) { a += partial[2 * i]; b += partial[2 * i + 1]; }
  a = wave_sum(a);
  b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) { wa[threadIdx.x >> 6] = a; wb[threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float sa = 0.f, sb = 0.f;
    for (int w = 0; w < 16; ++w) { sa += wa[w]; sb += wb[w]; }
    if (assign) { sums[0] = sa; sums[1] = sb; }
    else { sums[0] += sa; sums[1] += sb; }
  }
}

static int moe_grad_update_impl(const evc_bf16* dlogits, int64_t ld_dlogits, const evc_bf16* x, int64_t ldx, int rows,
                                int V, int K, float* p, float* m, float* v, evc_bf16* p_bf16, evc_bf16* pT_bf16, int64_t ldT,
                                float l2_coeff, float* sums, float* partial_ws, float clip_norm, float lr_t,
                                float beta1, float beta2, float eps, int phase, evc_bf16* p_wide, evc_f16* p_f16, uint8_t* p_fp8, int lo_exp, int hi_exp,
                                void* stream, float* wsq_out = nullptr);

extern "C" int evc_moe_grad_update_wide(const evc_bf16* dlogits, int64_t ld_dlogits, const evc_bf16* x, int64_t ldx, int rows,
                                        int V, int K, float* p, float* m, float* v, evc_bf16* p_bf16, evc_bf16* pT_bf16, int64_t ldT,
                                        evc_bf16* p_wide_hilo, evc_f16* p_f16, uint8_t* p_fp8, int fp8_lo_exp, int fp8_hi_exp,
                                        float l2_coeff, float* sums, float* partial_ws, float clip_norm, float lr_t,
                                        float beta1, float beta2, float eps, void* stream) {
  EVC_REQUIRE(p_wide_hilo || p_f16, EVC_ERR_BAD_ARG, "evc_moe_grad_update_wide: p_wide_hilo [V][2K] or p_f16 [V][K] + p_fp8 [V][2K] is required");
  EVC_REQUIRE(!p_wide_hilo || ((uintptr_t)p_wide_hilo % 8) == 0, EVC_ERR_BAD_ALIGN, "evc_moe_grad_update_wide: p_wide_hilo must be 8-byte aligned");
  EVC_REQUIRE((p_f16 == nullptr) == (p_fp8 == nullptr) && (!p_f16 || (((uintptr_t)p_f16 % 8) == 0 && ((uintptr_t)p_fp8 % 4) == 0)), EVC_ERR_BAD_ARG,
              "evc_moe_grad_update_wide: p_f16 (8-byte aligned) and p_fp8 (4-byte aligned) go together");
  EVC_REQUIRE(!p_f16 || (fp8_lo_exp >= 0 && fp8_lo_exp <= 60 && fp8_hi_exp >= -30 && fp8_hi_exp <= 30), EVC_ERR_BAD_ARG,
              "evc_moe_grad_update_wide: fp8_lo_exp=%d fp8_hi_exp=%d", fp8_lo_exp, fp8_hi_exp);
  return moe_grad_update_impl(dlogits, ld_dlogits, x, ldx, rows, V, K, p, m, v, p_bf16, pT_bf16, ldT, l2_coeff, sums, partial_ws, clip_norm, lr_t,
                              beta1, beta2, eps, 0, p_wide_hilo, p_f16, p_fp8, fp8_lo_exp, fp8_hi_exp, stream);
}

extern "C" int evc_moe_grad_update_phase(const evc_bf16* dlogits, int64_t ld_dlogits, const evc_bf16* x, int64_t ldx, int rows,
                                         int V, int K, float* p, float* m, float* v, evc_bf16* p_bf16, evc_bf16* pT_bf16, int64_t ldT,
                                         float l2_coeff, float* sums, float* partial_ws, float clip_norm, float lr_t,
                                         float beta1, float beta2, float eps, int phase, void* stream) {
  return moe_grad_update_impl(dlogits, ld_dlogits, x, ldx, rows, V, K, p, m, v, p_bf16, pT_bf16, ldT, l2_coeff, sums, partial_ws, clip_norm, lr_t,
                              beta1, beta2, eps, phase, nullptr, nullptr, nullptr, 0, 0, stream);
}

static int moe_grad_update_impl(const evc_bf16* dlogits, int64_t ld_dlogits, const evc_bf16* x, int64_t ldx, int rows,
                                int V, int K, float* p, float* m, float* v, evc_bf16* p_bf16, evc_bf16* pT_bf16,
                                int64_t ldT, float l2_coeff, float* sums, float* partial_ws, float clip_norm,
                                float lr_t, float beta1, float beta2, float eps, int phase, evc_bf16* p_wide, evc_f16* p_f16, uint8_t* p_fp8,
                                int lo_exp, int hi_exp, void* stream, float* wsq_out) {
  EVC_REQUIRE(rows > 0 && rows % 32 == 0 && V > 0 && V % 4 == 0 && K > 0 && K % 8 == 0, EVC_ERR_BAD_SHAPE,
              "evc_moe_grad_update: rows=%d (%%32), V=%d (%%4), K=%d (%%8)", rows, V, K);
  EVC_REQUIRE(phase >= 0 && phase <= 2, EVC_ERR_BAD_ARG, "evc_moe_grad_update_phase: phase=%d (0 both, 1 norms, 2 update)", phase);
  EVC_REQUIRE(ld_dlogits % 8 == 0 && ld_dlogits >= V && ldx % 8 == 0 && ldT % 4 == 0 && ldT >= V &&
              ((uintptr_t)dlogits % 16) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)p % 16) == 0 && ((uintptr_t)m % 16) == 0 &&
              ((uintptr_t)v % 16) == 0 && ((uintptr_t)p_bf16 % 8) == 0 && ((uintptr_t)pT_bf16 % 8) == 0, EVC_ERR_BAD_ALIGN,
              "evc_moe_grad_update: operand alignment / leading dimensions");
  hipStream_t st = (hipStream_t)stream;
  // 128x128 tiles, two workgroups per CU: the kernel is a stream over W, m, v with an 8-step GEMM in front - what
  // counts is how many epilogue loads are in flight per CU (256x256 tiles, one workgroup per CU: 3.4 TB/s)
  typedef CfgTn128 Cfg;
  const int Vp = (int)(ld_dlogits < ((V + 7) / 8) * 8 ? ld_dlogits : ((V + 7) / 8) * 8);   // A columns the loop may touch (%8)
  GemmOperandsT g{dlogits, ld_dlogits, x, ldx, Vp, K, rows / 32};
  const int tm = ceil_div(V, Cfg::BM), tn = ceil_div(K, Cfg::BU);
  EVC_REQUIRE(wsq_out == nullptr || phase == 2, EVC_ERR_BAD_ARG, "evc_moe_grad_update_apply: wsq_out goes with the update pass alone");
  MoeUpdateParams u{p, m, v, p_bf16, pT_bf16, ldT, p_wide, (bf16_t*)p_f16, p_fp8, ldexpf(1.0f, lo_exp), ldexpf(1.0f, hi_exp),
                    partial_ws, wsq_out ? partial_ws : nullptr, sums, V, K, l2_coeff, clip_norm, lr_t, beta1, beta2, eps};
  if (phase != 2) {
    launch_cfg<Cfg>(moe_update_kernel<Cfg, 1>, tm * tn, st, g, u, tm, tn);
    hipLaunchKernelGGL(moe_update_finalize_kernel, dim3(1), dim3(1024), 0, st, (const float*)partial_ws, tm * tn, sums, 0);
  }
  if (phase != 1) launch_cfg<Cfg>(moe_update_kernel<Cfg, 2>, tm * tn, st, g, u, tm, tn);
  if (wsq_out) hipLaunchKernelGGL(moe_update_finalize_kernel, dim3(1), dim3(1024), 0, st, (const float*)partial_ws, tm * tn, wsq_out, 1);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// The update pass alone (clip scale from sums[0], which evc_moe_grad_norms has filled), with any of the forward operand images of
// evc_moe_grad_update_wide (all three may be NULL: plain bf16) and wsq_out[0] = sum of the NEW weights squared (wsq_out[1] = 0):
// the |W|^2 term of the next update's norm.
extern "C" int evc_moe_grad_update_apply(const evc_bf16* dlogits, int64_t ld_dlogits, const evc_bf16* x, int64_t ldx, int rows,
                                         int V, int K, float* p, float* m, float* v, evc_bf16* p_bf16, evc_bf16* pT_bf16, int64_t ldT,
                                         evc_bf16* p_wide_hilo, evc_f16* p_f16, uint8_t* p_fp8, int fp8_lo_exp, int fp8_hi_exp,
                                         float l2_coeff, const float* sums, float* partial_ws, float clip_norm, float lr_t,
                                         float beta1, float beta2, float eps, float* wsq_out, void* stream) {
  EVC_REQUIRE(wsq_out != nullptr, EVC_ERR_BAD_ARG, "evc_moe_grad_update_apply: wsq_out is required");
  EVC_REQUIRE(!p_wide_hilo || ((uintptr_t)p_wide_hilo % 8) == 0, EVC_ERR_BAD_ALIGN, "evc_moe_grad_update_apply: p_wide_hilo must be 8-byte aligned");
  EVC_REQUIRE((p_f16 == nullptr) == (p_fp8 == nullptr) && (!p_f16 || (((uintptr_t)p_f16 % 8) == 0 && ((uintptr_t)p_fp8 % 4) == 0)), EVC_ERR_BAD_ARG,
              "evc_moe_grad_update_apply: p_f16 (8-byte aligned) and p_fp8 (4-byte aligned) go together");
  EVC_REQUIRE(!p_f16 || (fp8_lo_exp >= 0 && fp8_lo_exp <= 60 && fp8_hi_exp >= -30 && fp8_hi_exp <= 30), EVC_ERR_BAD_ARG,
              "evc_moe_grad_update_apply: fp8_lo_exp=%d fp8_hi_exp=%d", fp8_lo_exp, fp8_hi_exp);
  return moe_grad_update_impl(dlogits, ld_dlogits, x, ldx, rows, V, K, p, m, v, p_bf16, pT_bf16, ldT, l2_coeff, (float*)sums, partial_ws, clip_norm, lr_t,
                              beta1, beta2, eps, 2, p_wide_hilo, p_f16, p_fp8, fp8_lo_exp, fp8_hi_exp, stream, wsq_out);
}

extern "C" int evc_moe_grad_update(const evc_bf16* dlogits, int64_t ld_dlogits, const evc_bf16* x, int64_t ldx, int rows,
                                   int V, int K, float* p, float* m, float* v, evc_bf16* p_bf16, evc_bf16* pT_bf16, int64_t ldT,
                                   float l2_coeff, float* sums, float* partial_ws, float clip_norm, float lr_t,
                                   float beta1, float beta2, float eps, void* stream) {
  return evc_moe_grad_update_phase(dlogits, ld_dlogits, x, ldx, rows, V, K, p, m, v, p_bf16, pT_bf16, ldT, l2_coeff, sums,
                                   partial_ws, clip_norm, lr_t, beta1, beta2, eps, 0, stream);
}

