// Generic NT GEMM entry points (bf16 / f16 / split-bf16 / f16 + e4m3 low-order stages); DESIGN.md 4.1-4.2a.
#include "gemm_shared.h"

template <class Cfg, bool F16 = false, bool FP8 = false>
__global__ __launch_bounds__(Cfg::NT) void gemm_nt_kernel(GemmOperands p, StoreParams s, int tiles_m, int tiles_n) {
  static_assert(!FP8 || (F16 && is_v3<Cfg>::value), "the e4m3 tail rides behind f16 stages of the 64-wide ring loop");
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x, split = 0;
  if (s.splits > 1) {               // this workgroup's K range (wave-uniform)
    split = bid / nwg;
    bid -= split * nwg;
    const int k0 = split * s.ksteps_per_split;
    const int kstep = (is_v2<Cfg>::value && !is_v3<Cfg>::value) ? 32 : 64;
    p.A1 += (long)k0 * kstep;
    p.B += (long)k0 * kstep;
    p.nk1 = min(s.ksteps_per_split, p.nk1 - k0);
    if constexpr (FP8) {
      const int k8 = split * s.ksteps8_per_split;
      p.A3 += (long)k8 * 128;
      p.B8 += (long)k8 * 128;
      p.nk3 = min(s.ksteps8_per_split, p.nk3 - k8);
    }
  }
  const int id = xcd_remap(bid, nwg);
  int tm, tn;
  tile_of(id, tiles_m, tiles_n, tm, tn, s.splits > 1 ? patch_rows(nwg, tiles_n) : 8);
  const int m0 = tm * Cfg::BM, u0 = tn * Cfg::BU;
  f32x4 acc[Cfg::MI][Cfg::G][Cfg::NI];
  constexpr bool V2 = is_v2<Cfg>::value;
  // loop options (gemm_core_v2.h): producer waves + LDS-DMA first + no priority flips for every ring tile but the 320-row one
  // (same box: L1 dX 477 -> 431 us, 1280 x 4096 x 4096 63 -> 56.5 us, MoE gates forward 48.3 -> 46 us; 5120 x 4096 x 4096 on
  // the 320-row tile 153 -> 162 us with them)
#ifndef EVC_TALL_B_NT
#define EVC_TALL_B_NT 1      // the weight rows of the batch-row products (256 x 64 tiles: every weight row is read by ONE workgroup) as non-temporal LDS-DMA loads
                             // (round 5; same box, alternated three times: the MoE gates product ALONE 42.9 -> 49.3 us, the training step 9.96 -> 9.92 ms -
                             // 193 MB of weights per product no longer evict what the other streams' kernels re-read; 0 = default policy)
#endif
  constexpr int NT_MODE = (Cfg::BM == 320 ? 0 : (LOOP_PRODUCER | LOOP_DMA_FIRST | LOOP_NO_PRIO)) | (F16 ? LOOP_F16 : 0) | (FP8 ? LOOP_FP8_TAIL : 0) |
                          ((EVC_TALL_B_NT && is_v3<Cfg>::value && Cfg::BM == 256 && Cfg::BU == 64) ? LOOP_B_NT : 0);
  run_mainloop<Cfg, Cfg::G, V2, true, NT_MODE>(p, m0, u0, acc);     // ring tiles: transposed accumulators (lane = one row, 4 consecutive columns)
  if constexpr (V2) {
    // plain overwrite with 16-byte-aligned rows, or the split-K join: through LDS (kernel-uniform conditions: one barrier)
    const int es = s.out_bf16 ? 2 : 4;
    const bool lds_store = s.splits == 1 && !s.accumulate && (s.ldc * es) % 16 == 0 && ((uintptr_t)s.C % 16) == 0 &&
                           (!s.bias || ((uintptr_t)s.bias % 16) == 0);
    const int wave = threadIdx.x >> 6, wc = wave % Cfg::WC;
    const bool wave_cols_in = u0 + wc * Cfg::WU + Cfg::WU <= s.N;         // this wave's column span lies inside C
    if constexpr (Cfg::WU <= 64) {                                       // (wider wave tiles are never launched with a K split)
      if (s.splits > 1) {
        __syncthreads();                                                 // every wave has read its last ring slot
        store_tile_via_lds<Cfg, 4, true>(acc, lds_dyn, s.C, s.ldc, s.M, s.N, m0, u0, split == 0 ? s.bias : nullptr);
        return;
      }
    }
    if (lds_store) {
      __syncthreads();
      if (wave_cols_in) {
        if (s.out_bf16) store_tile_via_lds<Cfg, 2>(acc, lds_dyn, s.C, s.ldc, s.M, s.N, m0, u0, s.bias);
        else store_tile_via_lds<Cfg, 4>(acc, lds_dyn, s.C, s.ldc, s.M, s.N, m0, u0, s.bias, 0, s.sq_p, s.sq_l2, s.sq_out);
        return;
      }
    }
    TileCoordsT<Cfg> tc;                               // element-wise: accumulate, unaligned rows, the ragged right edge
#pragma unroll
    for (int mi = 0; mi < Cfg::MI; ++mi) {
      const int m = m0 + tc.row0 + mi * 16;
      if (m >= s.M) continue;
#pragma unroll
      for (int ni = 0; ni < Cfg::NI; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = u0 + tc.unit0 + ni * 16 + r;
          if (n >= s.N) continue;
          float v = acc[mi][0][ni][r] + ((s.bias && split == 0) ? s.bias[n] : 0.f);
          const long o = (long)m * s.ldc + n;
          if (s.splits > 1) {
            atomicAdd((float*)s.C + o, v);          // one global_atomic_add_f32 per element, executed at the memory side
          } else if (s.out_bf16) {
            ((bf16_t*)s.C)[o] = f32_to_bf16(v);
          } else {
            float* cp = (float*)s.C + o;
            if (s.accumulate) v += *cp;
            *cp = v;
          }
        }
    }
  } else {
    TileCoords<Cfg> tc;
#pragma unroll
    for (int mi = 0; mi < Cfg::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < Cfg::NI; ++ni) {
        const int n = u0 + tc.unit0 + ni * 16;
        if (n >= s.N) continue;
        const float b = (s.bias && split == 0) ? s.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + tc.row0 + mi * 16 + r;
          if (m >= s.M) continue;
          float v = acc[mi][0][ni][r] + b;
          const long o = (long)m * s.ldc + n;
          if (s.splits > 1) {
            atomicAdd((float*)s.C + o, v);          // one global_atomic_add_f32 per element, executed at the memory side
          } else if (s.out_bf16) {
            ((bf16_t*)s.C)[o] = f32_to_bf16(v);
          } else {
            float* cp = (float*)s.C + o;
            if (s.accumulate) v += *cp;
            *cp = v;
          }
        }
      }
  }
}


template <class Cfg>
static inline void launch_gemm(GemmOperands p, StoreParams s, int K, int splits, hipStream_t st) {
  p.nk1 = K / kdiv<Cfg>();
  const int tm = ceil_div(s.M, Cfg::BM), tn = ceil_div(s.N, Cfg::BU);
  s.splits = splits;
  s.ksteps_per_split = ceil_div(p.nk1, splits);
  launch_cfg<Cfg>(gemm_nt_kernel<Cfg>, tm * tn * splits, st, p, s, tm, tn);
}

static int gemm_nt_impl(const evc_bf16* A, int64_t lda, const evc_bf16* B, int64_t ldb, void* C, int64_t ldc,
                        int M, int N, int K, const float* bias, int out_bf16, int accumulate, void* stream,
                        const float* sq_p = nullptr, float sq_l2 = 0.f, float* sq_out = nullptr) {
  EVC_REQUIRE(M > 0 && N > 0 && K >= 0, EVC_ERR_BAD_SHAPE, "evc_gemm_nt: bad shape M=%d N=%d K=%d", M, N, K);
  // fused squared norm: only where the product runs as ONE pass of ring tiles that store whole rows through LDS (no K split, no ragged edge)
  EVC_REQUIRE(!sq_out || (!out_bf16 && !accumulate && !bias && M > 512 && N % 256 == 0 && K < 8192 && ldc % 4 == 0 && ((uintptr_t)C % 16) == 0 &&
                          (!sq_p || ((uintptr_t)sq_p % 16) == 0) && forced_tile() == 0), EVC_ERR_BAD_ARG,
              "evc_gemm_nt_sqnorm: needs a plain f32 product with M > 512, N %% 256 == 0, K < 8192, 16-byte aligned C / P (M=%d N=%d K=%d ldc=%ld)", M, N, K, (long)ldc);
  EVC_REQUIRE(K % 64 == 0, EVC_ERR_BAD_SHAPE, "evc_gemm_nt: K=%d must be a multiple of 64", K);
  EVC_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0,
              EVC_ERR_BAD_ALIGN, "evc_gemm_nt: operands must be 16-byte aligned (lda=%ld ldb=%ld)", (long)lda, (long)ldb);
  EVC_REQUIRE(!(out_bf16 && accumulate), EVC_ERR_BAD_ARG, "evc_gemm_nt: accumulate needs f32 output");
  EVC_REQUIRE(ring_operand_ok(M, lda) && ring_operand_ok(N, ldb), EVC_ERR_BAD_SHAPE,
              "evc_gemm_nt: an operand spans 4 GiB or more (M=%d lda=%ld, N=%d ldb=%ld): split the product", M, (long)lda, N, (long)ldb);
  GemmOperands p;
  p.A1 = A; p.lda1 = lda; p.nk1 = 0; p.A2 = A; p.lda2 = lda; p.nk2 = 0;
  p.B = B; p.ldb = ldb; p.group_stride = 0; p.M = M; p.Nu = N;
  p.A1lo = p.A2lo = p.Blo = nullptr;
  StoreParams s{C, ldc, M, N, bias, out_bf16, accumulate, 1, 0};
  s.sq_p = sq_p; s.sq_l2 = sq_l2; s.sq_out = sq_out;
  hipStream_t st = (hipStream_t)stream;
  // M <= 256 (one row tile: the MoE head on a batch of videos, [B, K] x [N, K]^T with N or K ~ 14k): the product
  // streams the weight matrix once from HBM, so it wants ~256 workgroups pulling at the same time and a deep
  // load pipeline rather than a square tile: 256x64 tiles, K split until ~256 workgroups exist.
  if (M <= 256 && N >= 1024 && K >= 1024 && forced_tile() == 0) {
    const int tn = ceil_div(N, 64);
    int splits = out_bf16 ? 1 : 256 / tn;
    if (splits > K / 1024) splits = K / 1024;
    if (splits < 1 || evc_deterministic()) splits = 1;
    if (splits > 1 && !accumulate) EVC_CHECK_HIP(hipMemset2DAsync(C, ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, st));
    static const bool tall_v2 = getenv("EVC_TALL_V2") != nullptr;      // A/B: the 32-wide K stages
    if (tall_v2) launch_gemm<CfgTallV2>(p, s, K, splits, st);
    else launch_gemm<TileCfg3<256, 1, 64, 2, 4, 4>>(p, s, K, splits, st);   // 64-wide K stages: the [256][K] row operand is re-read from L2 by every workgroup
    EVC_LAUNCH_CHECK();
    return EVC_OK;
  }
  // A few hundred rows against a long K with few output columns (DBoF: hidden layer [512 x 8192] . [1024 x 8192]^T, the MoE
  // head's dX at batch 512 with K = 14148 / 9432): eight 256x256 tiles cannot be split far enough to fill the chip (K/2048
  // splits = 32-48 workgroups, measured 105 us for 8.6 GFLOP).  128x128 ring tiles at two workgroups per CU instead, K split
  // until ~512 workgroups exist (>= 16 K steps each): the partial tiles are joined by f32 atomics into a zeroed C.
  // (M <= 512 only: streams.concurrent_streams probes with a 1024 x 1024 x 4096 product that must stay on 64 workgroups.)
  {
    const long t128 = (long)ceil_div(M, 128) * ceil_div(N, 128);
    if (!out_bf16 && M > 256 && M <= 512 && t128 <= 128 && K >= 4096 && forced_tile() == 0 && !evc_deterministic()) {
      const int nk = K / 32;
      int splits = (int)(512 / t128);
      if (splits > nk / 16) splits = nk / 16;
      while (splits > 1 && (long)ceil_div(nk, splits) * (splits - 1) >= nk) --splits;     // no empty split
      if (splits > 1) {
        if (!accumulate) EVC_CHECK_HIP(hipMemset2DAsync(C, ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, st));
        launch_gemm<CfgTn128>(p, s, K, splits, st);
        EVC_LAUNCH_CHECK();
        return EVC_OK;
      }
    }
  }
  // Split-K: a long-K product with too few 256x256 tiles to fill the 256 CUs (the weight-gradient
  // GEMMs: M=4H, N~1-2K, K = T*M rows) is cut along K; partial tiles are summed with f32 atomics
  // into a zeroed C (63 MB of atomic traffic at ~1.3 TB/s << the ~0.7 ms it saves per GEMM).
  {
    const long t2 = (long)ceil_div(M, 256) * ceil_div(N, 256);
    if (!out_bf16 && t2 <= 128 && K >= 8192 && forced_tile() == 0 && !evc_deterministic()) {
      int splits = (int)(256 / t2);
      const int max_by_k = K / 2048;                  // keep >= 64 K steps per split
      if (splits > max_by_k) splits = max_by_k;
      if (splits > 1) {
        if (!accumulate) EVC_CHECK_HIP(hipMemset2DAsync(C, ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, st));
        launch_gemm<CfgPlainV2>(p, s, K, splits, st);
        EVC_LAUNCH_CHECK();
        return EVC_OK;
      }
    }
  }
  // per-flop cost factors measured on MI355X (scripts/gemm_bench.py): v2 ~1000 TF/s, v1 128^2 ~800, v1 64^2 ~400
  const double c_v2 = tile_cost((long)ceil_div(M, 256) * ceil_div(N, 256), 256, 256, 1, 1.0);
  const double c_v2b = tile_cost((long)ceil_div(M, 224) * ceil_div(N, 256), 224, 256, 1, 1.01);   // e.g. 56 640 rows x 1024: 1012 tiles = 3.95 rounds instead of 888 = 3.47
  const double c_big = tile_cost((long)ceil_div(M, 128) * ceil_div(N, 128), 128, 128, 2, 1.3);
  const double c_small = tile_cost((long)ceil_div(M, 64) * ceil_div(N, 64), 64, 64, 4, 2.6);
  const double c_v2c = tile_cost((long)ceil_div(M, 320) * ceil_div(N, 256), 320, 256, 1, 1.03);   // the L2 hoist / dX: 5120 x 4096 x 4096 = 16 x 16 tiles
  double c_ring = c_v2;
  int ring = 1;
  if (c_v2b < c_ring) { c_ring = c_v2b; ring = 4; }
  if (c_v2c < c_ring) { c_ring = c_v2c; ring = 6; }
  // 160 x 128 ring tiles: 1280 rows x 4096 columns (the student's L2 hoist / dX at batch 256) = 8 x 32 = 256 tiles, one round
  const double c_160 = tile_cost((long)ceil_div(M, 160) * ceil_div(N, 128), 160, 128, 1, 1.5);
  if (K >= 2048 && c_160 < c_ring) { c_ring = c_160; ring = 7; }
  int pick = (c_ring <= c_big && c_ring <= c_small) ? ring : (c_big <= c_small ? 2 : 3);
  // 128x128 ring tiles (LDS-DMA ring instead of the v1 two-stage loop): 20-25 % faster than the v1 128x128 tile while the
  // whole product is one round of <= 256 tiles (measured: 1024 x 4096 x 4096 50 vs 63 us, 2048^3 28 vs 38, 1280 x 1024 x 4096
  // 45 vs 58); beyond that two workgroups share a CU's L2 ingest and the v1 / 256x256 tiles win again
  if (pick == 2 && K >= 2048 && (long)ceil_div(M, 128) * ceil_div(N, 128) <= 256) pick = 5;
  if (sq_out && (pick == 2 || pick == 3)) pick = ring;            // (the fused norm lives in the ring tiles' LDS store)
  if (forced_tile()) pick = forced_tile();
  static const bool nt_v3 = getenv("EVC_NT_BIG_V2") == nullptr;          // the 224/256-row tiles on two 64-wide stages (A/B switch: the five 32-wide ones)
  static const bool v3 = getenv("EVC_NT_V2_LOOP") == nullptr;      // 64-wide K stages for the 128-column ring tiles (A/B switch)
#ifdef EVC_EXPERIMENT_4WAVE
  if (pick == 12) { launch_gemm<TileCfg3<256, 1, 256, 2, 2, 2>>(p, s, K, 1, st); EVC_LAUNCH_CHECK(); return EVC_OK; }   // 4 waves, 128 x 128 per wave
#endif
  if (pick == 5 && v3) launch_gemm<TileCfg3<128, 1, 128, 2, 4, 4>>(p, s, K, 1, st);
  else if (pick == 5) launch_gemm<CfgTn128>(p, s, K, 1, st);
  else if (pick == 4 && nt_v3) launch_gemm<TileCfg3<224, 1, 256, 2, 4, 2>>(p, s, K, 1, st);
  else if (pick == 4) launch_gemm<CfgPlainV2_224>(p, s, K, 1, st);
  else if (pick == 6) launch_gemm<CfgPlainV2_320>(p, s, K, 1, st);
  else if (pick == 1 && nt_v3) launch_gemm<TileCfg3<256, 1, 256, 2, 4, 2>>(p, s, K, 1, st);
  else if (pick == 7 && v3) launch_gemm<TileCfg3<160, 1, 128, 2, 4, 4>>(p, s, K, 1, st);
  else if (pick == 7) launch_gemm<TileCfg2<160, 1, 128, 2, 4, 5, true>>(p, s, K, 1, st);
  else if (pick == 1) launch_gemm<CfgPlainV2>(p, s, K, 1, st);
  else if (pick == 2) launch_gemm<CfgPlainBig>(p, s, K, 1, st);
  else launch_gemm<CfgPlainSmall>(p, s, K, 1, st);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}
extern "C" int evc_gemm_nt(const evc_bf16* A, int64_t lda, const evc_bf16* B, int64_t ldb, void* C, int64_t ldc,
                           int M, int N, int K, const float* bias, int out_bf16, int accumulate, void* stream) {
  return gemm_nt_impl(A, lda, B, ldb, C, ldc, M, N, K, bias, out_bf16, accumulate, stream);
}
// evc_gemm_nt (plain f32 output) + evc_grad_sqnorm in one pass (round 6): sums[0] += sum of (C + l2_coeff * P)^2, sums[1] += sum of P^2 over the product's elements, from
// the tiles' stores (one f32 atomic per wave; P [M][N] f32 laid out as C, NULL with l2_coeff 0).  A weight gradient that is materialised anyway -
// the MoE head at 1024 rows, cfg 5 - then needs no separate norm pass before its clip + Adam (8 of 40 bytes per parameter).  sums must be zeroed
// by the caller (the launch ADDS to them); every wave stores its partial pair into part_ws (>= 16 * ceil(M/128) * ceil(N/128) floats) and a one-block
// finishing launch adds the slots in index order - no atomics, the same bits every run.
__global__ __launch_bounds__(1024) void sum_pairs_kernel(const float* __restrict__ part, int n, float* __restrict__ sums) {
  __shared__ float sh[2][16];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) { a += part[2 * i]; b += part[2 * i + 1]; }     // fixed order per thread, fixed tree below
  a = wave_sum(a); b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = a; sh[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float ta = 0.f, tb = 0.f;
    for (int i = 0; i < 16; ++i) { ta += sh[0][i]; tb += sh[1][i]; }
    sums[0] += ta; sums[1] += tb;
  }
}
extern "C" int evc_gemm_nt_sqnorm(const evc_bf16* A, int64_t lda, const evc_bf16* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K,
                                  const float* P, float l2_coeff, float* sums, float* part_ws, int64_t part_ws_floats, void* stream) {
  EVC_REQUIRE(sums && part_ws && (P || l2_coeff == 0.f), EVC_ERR_BAD_ARG, "evc_gemm_nt_sqnorm: sums / part_ws must not be NULL; P == NULL needs l2_coeff == 0");
  // every wave of every tile owns one {|C + l2 P|^2, |P|^2} slot: at most ceil(M / 128) * ceil(N / 128) workgroups of 8 waves whatever tile is picked
  const long slots = (long)ceil_div(M, 128) * ceil_div(N, 128) * 8;
  EVC_REQUIRE(part_ws_floats >= 2 * slots, EVC_ERR_BAD_ARG, "evc_gemm_nt_sqnorm: part_ws holds %ld floats, %ld needed", (long)part_ws_floats, 2 * slots);
  EVC_CHECK_HIP(hipMemsetAsync(part_ws, 0, (size_t)(2 * slots) * sizeof(float), (hipStream_t)stream));
  const int rc = gemm_nt_impl(A, lda, B, ldb, C, ldc, M, N, K, nullptr, 0, 0, stream, P, l2_coeff, part_ws);
  if (rc) return rc;
  hipLaunchKernelGGL(sum_pairs_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, part_ws, (int)slots, sums);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// ---------------------------------------------------------------------------
// Split-bf16 product C = (A_hi + A_lo) . (B_hi + B_lo)^T (+ bias) to ~2^-16 as ONE launch: the three products
// lo.hi + hi.lo + hi.hi are a K-extension of the plain loop.  A rows are the wide image [lo(K) | hi(K)]
// (evc_cast_f32_to_bf16_wide, lo first), B rows [hi(K) | lo(K)] (hi first): segment 1 walks A's whole row against B's whole row
// (k < K: lo.hi, k >= K: hi.lo), segment 2 walks A's hi half again against B2 = B's hi half (GemmOperands::B2 restarts B's k
// index).  Same accumulator, one epilogue, no split-K join - against three launches that each re-staged both operands and
// joined through C.  Ring tiles only (B2 lives in the v2 / v3 loops).
// ---------------------------------------------------------------------------
// C[M,N] f32 = A . B^T on IEEE f16 operands (one f16 MFMA product per depth): the hoisted input projection of the "high" mode's L2
// level (evc_lstm_stack2_fwd_f16).  Ring tiles, no K split.
template <class Cfg>
static inline void launch_gemm_f16(GemmOperands p, StoreParams s, int K, hipStream_t st) {
  p.nk1 = K / kdiv<Cfg>();
  const int tm = ceil_div(s.M, Cfg::BM), tn = ceil_div(s.N, Cfg::BU);
  s.splits = 1; s.ksteps_per_split = p.nk1;
  launch_cfg<Cfg>(gemm_nt_kernel<Cfg, true>, tm * tn, st, p, s, tm, tn);
}
int gemm_nt_f16(const evc_f16* A, int64_t lda, const evc_f16* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K, void* stream) {
  EVC_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0,
              EVC_ERR_BAD_SHAPE, "f16 product: M=%d N=%d K=%d (K %% 64), operands 16-byte aligned", M, N, K);
  EVC_REQUIRE(ring_operand_ok(M, lda) && ring_operand_ok(N, ldb), EVC_ERR_BAD_SHAPE, "f16 product: an operand spans 4 GiB or more");
  GemmOperands p;
  p.A1 = (const bf16_t*)A; p.lda1 = lda; p.nk1 = 0; p.A2 = p.A1; p.lda2 = lda; p.nk2 = 0;
  p.B = (const bf16_t*)B; p.ldb = ldb; p.group_stride = 0; p.M = M; p.Nu = N;
  p.A1lo = p.A2lo = p.Blo = nullptr;
  StoreParams s{C, ldc, M, N, nullptr, 0, 0, 1, 0};
  hipStream_t st = (hipStream_t)stream;
  const double c320 = tile_cost((long)ceil_div(M, 320) * ceil_div(N, 256), 320, 256, 1, 1.03);
  const double c256 = tile_cost((long)ceil_div(M, 256) * ceil_div(N, 256), 256, 256, 1, 1.0);
  const double c160 = tile_cost((long)ceil_div(M, 160) * ceil_div(N, 128), 160, 128, 1, 1.5);
  const double c128 = tile_cost((long)ceil_div(M, 128) * ceil_div(N, 128), 128, 128, 2, 1.3);
  const double best = fmin(fmin(c320, c256), fmin(c160, c128));
  if (best == c320) launch_gemm_f16<CfgPlainV2_320>(p, s, K, st);
  else if (best == c256) launch_gemm_f16<TileCfg3<256, 1, 256, 2, 4, 2>>(p, s, K, st);
  else if (best == c160) launch_gemm_f16<TileCfg3<160, 1, 128, 2, 4, 4>>(p, s, K, st);
  else launch_gemm_f16<TileCfg3<128, 1, 128, 2, 4, 4>>(p, s, K, st);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

template <class Cfg>
static inline void launch_gemm_seg(GemmOperands p, StoreParams s, int K, hipStream_t st) {
  p.nk1 = 2 * K / kdiv<Cfg>(); p.nk2 = K / kdiv<Cfg>();
  const int tm = ceil_div(s.M, Cfg::BM), tn = ceil_div(s.N, Cfg::BU);
  s.splits = 1; s.ksteps_per_split = p.nk1;
  launch_cfg<Cfg>(gemm_nt_kernel<Cfg>, tm * tn, st, p, s, tm, tn);
}

extern "C" int evc_gemm_nt_split(const evc_bf16* A_lohi, int64_t lda, const evc_bf16* B_hilo, int64_t ldb, float* C, int64_t ldc,
                                 int M, int N, int K, const float* bias, void* stream) {
  EVC_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0, EVC_ERR_BAD_SHAPE, "evc_gemm_nt_split: bad shape M=%d N=%d K=%d (K %% 64)", M, N, K);
  EVC_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && lda >= 2L * K && ldb >= 2L * K && ((uintptr_t)A_lohi % 16) == 0 && ((uintptr_t)B_hilo % 16) == 0,
              EVC_ERR_BAD_ALIGN, "evc_gemm_nt_split: wide operands [2K] per row, 16-byte aligned (lda=%ld ldb=%ld K=%d)", (long)lda, (long)ldb, K);
  EVC_REQUIRE(ring_operand_ok(M, lda) && ring_operand_ok(N, ldb), EVC_ERR_BAD_SHAPE,
              "evc_gemm_nt_split: an operand spans 4 GiB or more (M=%d lda=%ld, N=%d ldb=%ld)", M, (long)lda, N, (long)ldb);
  GemmOperands p;
  p.A1 = A_lohi; p.lda1 = lda; p.nk1 = 0;             // [lo | hi] against [hi | lo]
  p.A2 = A_lohi + K; p.lda2 = lda; p.nk2 = 0;         // hi against ...
  p.B = B_hilo; p.ldb = ldb; p.B2 = B_hilo;           // ... hi (k index restarts)
  p.group_stride = 0; p.M = M; p.Nu = N;
  p.A1lo = p.A2lo = p.Blo = nullptr;
  StoreParams s{C, ldc, M, N, bias, 0, 0, 1, 0};
  hipStream_t st = (hipStream_t)stream;
  if (M <= 256) {                                       // batch-row products (MoE head): stream the weights once
    launch_gemm_seg<TileCfg3<256, 1, 64, 2, 4, 4>>(p, s, K, st);
  } else {
    const double c256 = tile_cost((long)ceil_div(M, 256) * ceil_div(N, 256), 256, 256, 1, 1.0);
    const double c224 = tile_cost((long)ceil_div(M, 224) * ceil_div(N, 256), 224, 256, 1, 1.01);
    const double c320 = tile_cost((long)ceil_div(M, 320) * ceil_div(N, 256), 320, 256, 1, 1.03);
    const double c160 = tile_cost((long)ceil_div(M, 160) * ceil_div(N, 128), 160, 128, 1, 1.5);
    const double c128 = tile_cost((long)ceil_div(M, 128) * ceil_div(N, 128), 128, 128, 2, 1.3);
    const double best = fmin(fmin(c256, c224), fmin(fmin(c320, c160), c128));
    if (best == c320) launch_gemm_seg<CfgPlainV2_320>(p, s, K, st);
    else if (best == c256) launch_gemm_seg<TileCfg3<256, 1, 256, 2, 4, 2>>(p, s, K, st);
    else if (best == c224) launch_gemm_seg<TileCfg3<224, 1, 256, 2, 4, 2>>(p, s, K, st);
    else if (best == c160) launch_gemm_seg<TileCfg3<160, 1, 128, 2, 4, 4>>(p, s, K, st);
    else launch_gemm_seg<TileCfg3<128, 1, 128, 2, 4, 4>>(p, s, K, st);
  }
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// C [M][N] f32 = A16 . B16^T (IEEE f16, K16 deep) + 2^scale_exp A8 . B8^T (OCP e4m3 bytes, K8 deep) + bias in ONE launch: a product whose
// low-order corrections ride behind its f16 stages on the MX-scaled MFMA (gemm_core_v3.h LOOP_FP8_TAIL) - the "high" precision MoE head:
// [f16(x)] . [f16(W)]^T + 2^-24 [e4m3(x 2^6) | e4m3((x - f16(x)) 2^17)] . [e4m3((W - f16(W)) 2^18) | e4m3(W 2^7)]^T leaves ~2.5e-5 on
// logits of magnitude 8 (f16 alone: 8e-4; scripts/precision_budget.py "MOE fine") for 2/3 of the operand bytes of the split-bf16
// K-extension (evc_gemm_nt_split).  lda / ldb in halfwords, lda8 / ldb8 in bytes.
template <class Cfg>
static inline void launch_gemm_f16_fp8(GemmOperands p, StoreParams s, int K16, int K8, int splits, hipStream_t st) {
  p.nk1 = K16 / 64; p.nk2 = 0; p.nk3 = K8 / 128; p.nk4 = 0;
  const int tm = ceil_div(s.M, Cfg::BM), tn = ceil_div(s.N, Cfg::BU);
  s.splits = splits; s.ksteps_per_split = p.nk1 / splits; s.ksteps8_per_split = p.nk3 / splits;     // (splits divides both: chosen so below)
  launch_cfg<Cfg>(gemm_nt_kernel<Cfg, true, true>, tm * tn * splits, st, p, s, tm, tn);
}

static int gemm_nt_f16_fp8_impl(const evc_f16* A16, int64_t lda, const uint8_t* A8, int64_t lda8, const evc_f16* B16, int64_t ldb,
                                const uint8_t* B8, int64_t ldb8, float* C, int64_t ldc, int M, int N, int K16, int K8, int scale_exp,
                                const float* amax_ws, int a8_hi_exp, const float* bias, void* stream) {
  EVC_REQUIRE(M > 0 && N > 0 && K16 >= 64 && K16 % 64 == 0 && K8 >= 512 && K8 % 128 == 0, EVC_ERR_BAD_SHAPE,
              "evc_gemm_nt_f16_fp8: bad shape M=%d N=%d K16=%d (%%64, >= 64) K8=%d (%%128, >= 512)", M, N, K16, K8);
  EVC_REQUIRE(A16 && A8 && B16 && B8 && C, EVC_ERR_BAD_ARG, "evc_gemm_nt_f16_fp8: NULL operand");
  EVC_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && lda8 % 16 == 0 && ldb8 % 16 == 0 && lda >= K16 && ldb >= K16 && lda8 >= K8 && ldb8 >= K8 &&
              ((uintptr_t)A16 % 16) == 0 && ((uintptr_t)B16 % 16) == 0 && ((uintptr_t)A8 % 16) == 0 && ((uintptr_t)B8 % 16) == 0, EVC_ERR_BAD_ALIGN,
              "evc_gemm_nt_f16_fp8: 16-byte aligned operands and row strides (lda=%ld ldb=%ld lda8=%ld ldb8=%ld)", (long)lda, (long)ldb, (long)lda8, (long)ldb8);
  EVC_REQUIRE(scale_exp >= -60 && scale_exp <= 60, EVC_ERR_BAD_ARG, "evc_gemm_nt_f16_fp8: scale_exp=%d", scale_exp);
  EVC_REQUIRE(ring_operand_ok(M, lda) && ring_operand_ok(N, ldb) && ring_operand_ok(M, (lda8 + 1) / 2) && ring_operand_ok(N, (ldb8 + 1) / 2), EVC_ERR_BAD_SHAPE,
              "evc_gemm_nt_f16_fp8: an operand spans 4 GiB or more");
  GemmOperands p;
  p.A1 = (const bf16_t*)A16; p.lda1 = lda; p.nk1 = 0; p.A2 = p.A1; p.lda2 = lda; p.nk2 = 0;
  p.B = (const bf16_t*)B16; p.ldb = ldb; p.group_stride = 0; p.M = M; p.Nu = N;
  p.A1lo = p.A2lo = p.Blo = nullptr;
  p.A3 = A8; p.lda3 = lda8; p.A4 = A8; p.lda4 = lda8; p.B8 = B8; p.ldb8 = ldb8; p.scale8_exp = scale_exp;
  p.amax_ws = amax_ws; p.a8_hi_exp = a8_hi_exp;
  StoreParams s{C, ldc, M, N, bias, 0, 0, 1, 0};
  hipStream_t st = (hipStream_t)stream;
  if (M <= 512) {      // batch-row products (MoE head, DBoF hidden layer): 256 x 64 tiles stream the weights once; few column tiles and a long K
                       // (512 x 1024 x 8192: 32 tiles) are cut along K until ~256 workgroups exist - every split takes the same share of the f16
                       // and of the e4m3 stages (>= 4 of each: a full ring), the partial tiles are joined by f32 atomics into a zeroed C
    const int tiles = ceil_div(M, 256) * ceil_div(N, 64), nk16 = K16 / 64, nk8 = K8 / 128;
    int splits = 1;
    while (!evc_deterministic() && splits < 16 && tiles * splits * 2 <= 256 && nk16 % (splits * 2) == 0 && nk8 % (splits * 2) == 0 &&
           nk16 / (splits * 2) >= 4 && nk8 / (splits * 2) >= 4)
      splits *= 2;
    if (splits > 1) EVC_CHECK_HIP(hipMemset2DAsync(C, ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, st));
    launch_gemm_f16_fp8<TileCfg3<256, 1, 64, 2, 4, 4>>(p, s, K16, K8, splits, st);
  } else {
    launch_gemm_f16_fp8<TileCfg3<256, 1, 256, 2, 4, 2>>(p, s, K16, K8, 1, st);
  }
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}
extern "C" int evc_gemm_nt_f16_fp8(const evc_f16* A16, int64_t lda, const uint8_t* A8, int64_t lda8, const evc_f16* B16, int64_t ldb,
                                   const uint8_t* B8, int64_t ldb8, float* C, int64_t ldc, int M, int N, int K16, int K8, int scale_exp,
                                   const float* bias, void* stream) {
  return gemm_nt_f16_fp8_impl(A16, lda, A8, lda8, B16, ldb, B8, ldb8, C, ldc, M, N, K16, K8, scale_exp, nullptr, 0, bias, stream);
}
// ... with the A8 images written by evc_cast_f32_to_f16_fp8x_dyn from the same amax_ws / hi_exp: every e4m3 product is scaled by
// 2^(scale_exp + d), d = the range shift both kernels derive from the 64 partial maxima (0 while max|x| 2^a8_hi_exp <= 448: the fixed-scale product).
extern "C" int evc_gemm_nt_f16_fp8_dyn(const evc_f16* A16, int64_t lda, const uint8_t* A8, int64_t lda8, const evc_f16* B16, int64_t ldb,
                                       const uint8_t* B8, int64_t ldb8, float* C, int64_t ldc, int M, int N, int K16, int K8, int scale_exp,
                                       const float* amax_ws, int a8_hi_exp, const float* bias, void* stream) {
  EVC_REQUIRE(amax_ws && a8_hi_exp >= -30 && a8_hi_exp <= 30, EVC_ERR_BAD_ARG, "evc_gemm_nt_f16_fp8_dyn: amax_ws=%p a8_hi_exp=%d", (const void*)amax_ws, a8_hi_exp);
  return gemm_nt_f16_fp8_impl(A16, lda, A8, lda8, B16, ldb, B8, ldb8, C, ldc, M, N, K16, K8, scale_exp, amax_ws, a8_hi_exp, bias, stream);
}

