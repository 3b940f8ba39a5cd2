// GEMM-shaped kernels of the hot path: generic NT GEMM, fused LSTM step
// (forward, with the gate tail in the epilogue) and fused BPTT step.
#include "gemm_launch.h"

// Loop options per kernel (gemm_core_v2.h; same-box A/B measurements in DESIGN.md 4.2)
#ifndef EVC_FWD_STORE_POLICY
#define EVC_FWD_STORE_POLICY 0      // cache policy of the forward step's epilogue stores (evc_common.h store16<>): 0 plain, 1 sc1 (write-through), 2 nt
#endif
#ifndef EVC_FWD_LOOP_MODE
#define EVC_FWD_LOOP_MODE (LOOP_PRODUCER | LOOP_DMA_FIRST | LOOP_NO_PRIO)   // forward step: 81.7 -> 77.4 us per step
#endif
#ifndef EVC_BWD_LOOP_MODE
#define EVC_BWD_LOOP_MODE (LOOP_PRODUCER | LOOP_DMA_FIRST | LOOP_NO_PRIO)   // BPTT step: 64.8 -> 62.1 us per step on the 32-wide stages without producers; on the 64-wide ones producers give another 56.0 -> 54.1
#endif
#ifndef EVC_TN_LOOP_MODE
#define EVC_TN_LOOP_MODE LOOP_PRODUCER                                      // weight-gradient products: -2 .. -5 %
#endif

template <class Cfg, int NG, bool SWAP = false, bool INIT = true, int MODE = EVC_LOOP_MODE_DEFAULT>
__device__ __forceinline__ void run_mainloop(const GemmOperands& p, int m0, int u0, f32x4 (&acc)[Cfg::MI][NG][Cfg::NI]) {
  if constexpr (is_v2<Cfg>::value) {
    if constexpr (is_v3<Cfg>::value) gemm_mainloop_v3<Cfg, SWAP, INIT, MODE>(p, m0, u0, lds_dyn, acc);
    else gemm_mainloop_v2<Cfg, SWAP, INIT, MODE>(p, m0, u0, lds_dyn, acc);
  } else {
    __shared__ __attribute__((aligned(16))) char lds_static[Cfg::LDS_BYTES];   // static: keeps 2 workgroups per CU
    gemm_mainloop<Cfg, SWAP, INIT, (MODE & LOOP_F16) != 0>(p, m0, u0, lds_static, acc);
  }
}

// K-step granularity of a config (v1 walks 64-wide tiles, v2 32-wide)
template <class Cfg> static inline int kdiv() { return (is_v2<Cfg>::value && !is_v3<Cfg>::value) ? 32 : 64; }

// Tile choice: a CU works through ceil(tiles/256) tiles (co-resident workgroups share its matrix
// pipe, so residency does not shorten that), each costing area x a per-flop factor measured on
// MI355X with scripts/gemm_bench.py (v2 ~1000 TF/s -> 1.0, v1 128x128 ~800 -> 1.3, v1 64x64 ~400 -> 2.6).
static inline double tile_cost(long tiles, int bm, int bn, int /*occ*/, double c) {
  const long per_cu = (tiles + 255) / 256;
  return (double)per_cu * bm * bn * c;
}

// ===========================================================================
// generic GEMM: C[M,N] (+)= A.B^T (+bias)
// ===========================================================================
struct StoreParams {
  void* C; long ldc; int M, N; const float* bias; int out_bf16; int accumulate;
  int splits, ksteps_per_split;   // split-K: blockIdx = split * tiles + tile; partial sums joined by f32 atomics
  int ksteps8_per_split = 0;      // (FP8 kernels: e4m3 stages per split - a split takes the same share of both stage ranges)
};

// Epilogue of the ring-tile (v2) kernels for a plain overwrite of C: every wave transposes its WM x WU sub-tile through
// its own slice of the (idle) LDS ring and stores whole rows of the sub-tile, 16 bytes per lane.  From the accumulator
// layout itself a store instruction touches 16-64 different lines with 2-32 bytes each, and the stores of a bf16 output
// were issue-bound: 0.41 -> 0.33 ms on 16384 x 8192 x 1152 (DESIGN.md 4.6 has the same measurement on the DBoF kernel).
// acc: TRANSPOSED accumulators (lane 16g + l: row mi*16 + l, columns ni*16 + 4g .. 4g+3).  ES = bytes per output element.
template <class Cfg, int ES, bool ATOMIC = false, bool RMW = false>     // RMW: C += tile by plain 16-byte read-modify-write (f32)
__device__ __forceinline__ void store_tile_via_lds(f32x4 (&acc)[Cfg::MI][1][Cfg::NI], char* lds, void* C, long ldc, int M, int N,
                                                   int m0, int u0, const float* bias, int row_il_H = 0) {
  static_assert(!ATOMIC || ES == 4, "split-K partial tiles are joined in f32");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave / Cfg::WC, wc = wave % Cfg::WC;
  const int l = lane & 15, g = lane >> 4;
  constexpr int RS = Cfg::WU * ES + 16;                        // padded row: 16-byte aligned reads, <= 2-way conflicts on the writes
  constexpr int RP_MAX = (Cfg::LDS_BYTES / (Cfg::WR * Cfg::WC)) / RS / 16 * 16;   // rows of the sub-tile per pass (multiple of 16)
  constexpr int RP = RP_MAX >= Cfg::WM ? Cfg::WM : RP_MAX;
  static_assert(RP >= 16, "LDS slice too small for one accumulator block");
  // read-back: plain stores move 16 bytes per lane (whole sub-tile rows); the split-K join moves ONE float per lane so that a
  // wave-instruction's atomics cover contiguous runs of a row (global float atomics run at full rate on 256 contiguous bytes
  // and ~17x slower on 64 scattered dwords - which is what the transposed accumulator layout would issue directly)
  constexpr int CPR = ATOMIC ? Cfg::WU : Cfg::WU * ES / 16;    // lanes per sub-tile row
  constexpr int RPI = 64 / CPR;                                // rows per instruction
  static_assert(64 % CPR == 0 && RP % RPI == 0, "sub-tile rows must divide into whole instructions");
  char* wl = lds + wave * (RP * RS);
  const int colw = u0 + wc * Cfg::WU;
  const int rbase = m0 + wr * Cfg::WM;
  float4 bv[Cfg::NI];
#pragma unroll
  for (int ni = 0; ni < Cfg::NI; ++ni) {
    bv[ni] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) {
      const int c = colw + ni * 16 + g * 4;
      if (!ATOMIC) bv[ni] = *(const float4*)(bias + c);
      else bv[ni] = make_float4(c < N ? bias[c] : 0.f, c + 1 < N ? bias[c + 1] : 0.f, c + 2 < N ? bias[c + 2] : 0.f, c + 3 < N ? bias[c + 3] : 0.f);
    }
  }
#pragma unroll
  for (int r0 = 0; r0 < Cfg::WM; r0 += RP) {
    if (r0 > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous pass's reads have their data (same wave, in order)
#pragma unroll
    for (int mi = r0 / 16; mi < (r0 + RP) / 16 && mi < Cfg::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < Cfg::NI; ++ni) {
        const f32x4 a = acc[mi][0][ni];
        const float v0 = a[0] + bv[ni].x, v1 = a[1] + bv[ni].y, v2 = a[2] + bv[ni].z, v3 = a[3] + bv[ni].w;
        char* d = wl + (mi * 16 - r0 + l) * RS + (ni * 16 + g * 4) * ES;
        if constexpr (ES == 2) *(uint2*)d = make_uint2(pack_bf16x2_hw(v0, v1), pack_bf16x2_hw(v2, v3));
        else *(float4*)d = make_float4(v0, v1, v2, v3);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < RP / RPI; ++it) {
      const int rl = it * RPI + lane / CPR;
      const int row = rbase + r0 + rl;
      const long orow = row_il_H > 0 ? (long)(row & 3) * row_il_H + (row >> 2) : row;   // gate de-interleave of the TN weight gradients
      if constexpr (ATOMIC) {
        const int c = lane % CPR;
        const float v = *(const float*)(wl + rl * RS + c * 4);
        if (row < M && r0 + rl < Cfg::WM && colw + c < N) atomicAdd((float*)C + orow * ldc + colw + c, v);
      } else {
        uint4 q = *(const uint4*)(wl + rl * RS + (lane % CPR) * 16);
        if (row < M && r0 + rl < Cfg::WM) {
          uint4* cp = (uint4*)((char*)C + (orow * ldc + colw) * ES + (lane % CPR) * 16);
          if constexpr (RMW) {
            const float4 o = *(const float4*)cp;
            const float4 a = *(const float4*)&q;
            *(float4*)cp = make_float4(o.x + a.x, o.y + a.y, o.z + a.z, o.w + a.w);
          } else {
            *cp = q;
          }
        }
      }
    }
  }
}

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
  constexpr int NT_MODE = (Cfg::BM == 320 ? 0 : (LOOP_PRODUCER | LOOP_DMA_FIRST | LOOP_NO_PRIO)) | (F16 ? LOOP_F16 : 0) | (FP8 ? LOOP_FP8_TAIL : 0);
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
        else store_tile_via_lds<Cfg, 4>(acc, lds_dyn, s.C, s.ldc, s.M, s.N, m0, u0, s.bias);
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

typedef TileCfg<128, 1, 128, 2, 2> CfgPlainBig;   // 128x128, 4 waves, 4x4 MFMA tiles per wave
typedef TileCfg<64, 1, 64, 2, 2> CfgPlainSmall;   // 64x64 for skinny problems
typedef TileCfg<32, 1, 32, 2, 2> CfgPlainTiny;    // 32x32: M ~ batch recurrent steps (256 workgroups at M=256, H=1024)
typedef TileCfg2<256, 1, 256, 2, 4, 5, true> CfgPlainV2;   // 256x256, 8 waves (2x4), 128x64 per wave, 5-deep ring (160 KiB)
typedef TileCfg2<224, 1, 256, 2, 4, 5, true> CfgPlainV2_224;   // same, 224 rows: picked when it cuts M into fewer rounds of 256 workgroups
typedef TileCfg2<320, 1, 256, 2, 4, 4, false> CfgPlainV2_320;  // 320 rows (4-deep ring, single fragment set): 5120 rows = 16 x 16 tiles, ONE round of 256 workgroups instead of 320 tiles
typedef TileCfg2<128, 1, 128, 2, 4, 5, true> CfgTn128;     // 128x128 v2 tile (80 KB ring: two workgroups per CU)
typedef TileCfg2<256, 1, 64, 2, 4, 5, true> CfgTallV2;     // 256x64: M <= 256 (batch-row) products against a long weight matrix
// (256x128 tiles + split-K 2, to halve the re-reads of the [256][K] row operand: 80 vs 61 us at N = 14148 - not the bound)

template <class Cfg>
static inline void launch_gemm(GemmOperands p, StoreParams s, int K, int splits, hipStream_t st) {
  p.nk1 = K / kdiv<Cfg>();
  const int tm = ceil_div(s.M, Cfg::BM), tn = ceil_div(s.N, Cfg::BU);
  s.splits = splits;
  s.ksteps_per_split = ceil_div(p.nk1, splits);
  launch_cfg<Cfg>(gemm_nt_kernel<Cfg>, tm * tn * splits, st, p, s, tm, tn);
}

extern "C" int evc_gemm_nt(const evc_bf16* A, int64_t lda, const evc_bf16* B, int64_t ldb, void* C, int64_t ldc,
                           int M, int N, int K, const float* bias, int out_bf16, int accumulate, void* stream) {
  EVC_REQUIRE(M > 0 && N > 0 && K >= 0, EVC_ERR_BAD_SHAPE, "evc_gemm_nt: bad shape M=%d N=%d K=%d", M, N, K);
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
static int gemm_nt_f16(const evc_f16* A, int64_t lda, const evc_f16* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K, void* stream) {
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

extern "C" int evc_gemm_nt_f16_fp8(const evc_f16* A16, int64_t lda, const uint8_t* A8, int64_t lda8, const evc_f16* B16, int64_t ldb,
                                   const uint8_t* B8, int64_t ldb8, float* C, int64_t ldc, int M, int N, int K16, int K8, int scale_exp,
                                   const float* bias, void* stream) {
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

// ===========================================================================
// TN GEMM: C[M,N] (+)= A^T . B with A [K][lda], B [K][ldb] (weight gradients without transposes)
// ===========================================================================
struct StoreParamsT {
  float* C; long ldc; int M, N;
  int row_il_H;                   // > 0: row m = u*4+g of the product is stored at row g*H+u (gate de-interleave)
  int accumulate, splits, ksteps_per_split;
  long slab_stride;               // > 0: split s stores its partial tile plainly at C + s*slab_stride (no atomics; the caller sums the slabs)
};

template <class Cfg>
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
  if (s.splits > 1) {
    const int k0 = split * s.ksteps_per_split;
    p.A += (long)k0 * 32 * p.lda;
    p.B += (long)k0 * 32 * p.ldb;
    p.nk = min(s.ksteps_per_split, p.nk - k0);
  }
  f32x4 acc[Cfg::MI][1][Cfg::NI];
  gemm_mainloop_tn<Cfg, true, EVC_TN_LOOP_MODE>(p, m0, nb, lds_dyn, acc);      // transposed accumulators: lane = one row, 4 consecutive columns
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
                        int c_col2, float* C, int64_t ldc, int M, int N, int K, int row_interleave_H, int accumulate, void* stream) {
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
  typedef TileCfg2<128, 1, 128, 2, 4, 5, true> CfgTn128;
  // short contractions (the student's L2: K = 5 x 256 rows) on 128x128 tiles without split-K: the atomic join of
  // 256x256 partial tiles costs more than the product itself there (81 -> 36 us at 4096 x 1024 x 1280); from
  // K ~ 5000 on the 256x256 split-K form is faster again (92 vs 99 us)
  if (forced_tile() == 11 || (forced_tile() == 0 && K <= 2048 && (long)ceil_div(M, 128) * ceil_div(N, 128) >= 192)) {
    const int tm1 = ceil_div(M, 128), tn1 = ceil_div(N, 128);
    StoreParamsT s1{C, ldc, M, N, row_interleave_H, accumulate, 1, p.nk, 0};
    launch_cfg<CfgTn128>(gemm_tn_kernel<CfgTn128>, tm1 * tn1, st, p, s1, tm1, tn1);
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
    launch_cfg<CfgTn128>(gemm_tn_kernel<CfgTn128>, tm1 * splits, st, p, s1, tm1, 1);
    EVC_LAUNCH_CHECK();
    return EVC_OK;
  }
  const int tm = ceil_div(M, CfgPlainV2::BM), tn = ceil_div(N, CfgPlainV2::BU);
  int splits = 256 / (tm * tn);
  if (splits > K / 1024) splits = K / 1024;     // keep >= 32 K steps per split
  if (splits < 1 || evc_deterministic()) splits = 1;
  StoreParamsT s{C, ldc, M, N, row_interleave_H, accumulate, splits, ceil_div(p.nk, splits), 0};
  if (splits > 1 && !accumulate) EVC_CHECK_HIP(hipMemset2DAsync(C, ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, st));
  launch_cfg<CfgPlainV2>(gemm_tn_kernel<CfgPlainV2>, tm * tn * splits, st, p, s, tm, tn);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
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
  bf16_t* p_bf16;                      // forward shadow [V][K]
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

template <class Cfg, int PASS>
__global__ __launch_bounds__(Cfg::NT) void moe_update_kernel(GemmOperandsT p, MoeUpdateParams u, int tiles_m, int tiles_n) {
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
      pv[mi][ni] = (vr < u.V && k < K) ? *(const float4*)(u.p + (long)vr * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
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
  float4 mv[Cfg::MI][Cfg::NI], vv[Cfg::MI][Cfg::NI];
#pragma unroll
  for (int mi = 0; mi < Cfg::MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < Cfg::NI; ++ni) {
      const int vr = m0 + tc.row0 + mi * 16, k = n0 + tc.unit0 + ni * 16;
      const bool ok = vr < u.V && k < K;
      mv[mi][ni] = ok ? *(const float4*)(u.m + (long)vr * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      vv[mi][ni] = ok ? *(const float4*)(u.v + (long)vr * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
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
        *(float4*)(u.p + o) = make_float4(pn[0], pn[1], pn[2], pn[3]);
        *(float4*)(u.m + o) = make_float4(mn[0], mn[1], mn[2], mn[3]);
        *(float4*)(u.v + o) = make_float4(vn[0], vn[1], vn[2], vn[3]);
        *(uint2*)(u.p_bf16 + o) = make_uint2((uint32_t)pb[0] | ((uint32_t)pb[1] << 16), (uint32_t)pb[2] | ((uint32_t)pb[3] << 16));
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
    *(uint2*)(u.pT_bf16 + (long)k * u.ldT + v4) = q;
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

// ===========================================================================
// LSTM forward step: z = [x_t, h_{t-1}] . W^T (+ zx) + bias ; gate tail fused
// ===========================================================================
struct LstmFwdParams {
  const float* zx; long ldzx;        // hoisted x-projection rows for this step (or NULL)
  const float* bias;                 // [4H]
  const int* len; int t;
  float* c_state; float* h_state; long ld_state;
  bf16_t* hout;                      // [M][H] slab t+1 (row-major: next step's A operand)
  bf16_t* hout_lo;                   // SPLIT: slab t+1 of the WIDE image [M][2H] = [lo(h_t) | hi(h_t)] (next step's split A operand);
                                     // F16 (hout then holds IEEE f16): the bf16 copy of h_t the backward pass reads; else NULL
  uint2* gates;                      // [M][H] 8-byte records of slab t (or NULL): bf16 {i, j, f, o}
  bf16_t* c_hist;                    // slab t+1 of the bf16 cell-state history [M][H] (c after this step), or NULL
  const int* row_map;                // slot -> row of c_state / h_state (row plan, evc_sort_rows_by_len) or NULL
  int M, H;
  int h_wide = 0;                    // F16 only: 1 = hout rows are WIDE, [M][2H] = [f16(h_t) | f16(h_t)/64] - the activation operand of a
                                     // contraction whose weights are K-extended by their low-order halves (evc_lstm_stack2_fwd_f16);
                                     // 2 = hout rows are [f16(h_t) (H halfwords) | e4m3(h_t * 2^7) (H bytes)], row stride 3H bytes - the
                                     // operands of a step whose low-order weight halves are contracted in fp8 (evc_lstm_layer_fwd_f16_fp8lo)
};

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) { return pack_bf16x2_hw(lo, hi); }

// F16: the operands (x_t, h_{t-1}, W) are IEEE f16 and ONE v_mfma_f32_16x16x32_f16 product is issued per depth - the cost of
// the bf16 step with 8x smaller operand rounding; h_t leaves twice, as f16 (next step's / next layer's operand) and as bf16
// (what the BPTT products contract over).
template <class Cfg, bool SPLIT = false, bool F16 = false, bool FP8 = false>
__device__ __forceinline__ void lstm_fwd_step_body(const GemmOperands& p, const LstmFwdParams& e, int tiles_m, int tiles_n, int bid) {
  static_assert(Cfg::G == 4, "LSTM step needs the four gate groups");
  static_assert(!(SPLIT && F16), "split operands are bf16 halves");
  static_assert(!FP8 || (F16 && is_v3<Cfg>::value), "the e4m3 tail rides behind f16 stages of the 64-wide ring loop");
  const int nwg = tiles_m * tiles_n;
  EVC_STAMP(p.stamp_slot, 0);
  const int id = xcd_remap(bid, nwg);
  int tm, tn;
  tile_of(id, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * Cfg::BM, u0 = tn * Cfg::BU;
  f32x4 acc[Cfg::MI][4][Cfg::NI];
  {   // the accumulators start from bias (+ forget_bias 1.0): its loads fly under the loop's prologue, and the tail
      // below has no load left that hipcc could re-issue between the fragments' stores
    TileCoordsT<Cfg> tc0;
#pragma unroll
    for (int ni = 0; ni < Cfg::NI; ++ni) {
      const int u = min(u0 + tc0.unit0 + ni * 16, e.H - 4);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b = *(const float4*)(e.bias + (long)g * e.H + u);
        const float fb = (g == 2) ? 1.0f : 0.0f;       // forget_bias
#pragma unroll
        for (int mi = 0; mi < Cfg::MI; ++mi) acc[mi][g][ni] = f32x4{b.x + fb, b.y + fb, b.z + fb, b.w + fb};
      }
    }
  }
  // (SPLIT: the split-bf16 products hi.hi + hi.lo + lo.hi are a K-EXTENSION of the same loop - the caller hands A = [lo | hi] rows
  //  against B = [W_hi | W_lo] rows as segment 1 and A = hi against B2 = W_hi as segment 2, evc_lstm_layer_fwd_hp - so the loop
  //  itself is the plain one; only the epilogue differs: it writes h_t's wide [lo | hi] image for the next step.)
  run_mainloop<Cfg, 4, true, false, EVC_FWD_LOOP_MODE | (F16 ? LOOP_F16 : 0) | (FP8 ? LOOP_FP8_TAIL : 0)>(p, m0, u0, acc);   // transposed accumulators: lane = one row, 4 consecutive units
  EVC_STAMP(p.stamp_slot, 2);
#ifdef EVC_ABLATE_EPI    // debug build: main loop only (keep the accumulators alive, store nothing)
#pragma unroll
  for (int mi = 0; mi < Cfg::MI; ++mi)
#pragma unroll
    for (int g = 0; g < 4; ++g) asm volatile("" :: "v"(acc[mi][g][0]));
  return;
#endif
  TileCoordsT<Cfg> tc;
  const int H = e.H;     // H % 4 == 0 (checked on the host): a lane's 4 units never straddle H
  // Every load of the tail is issued before the first store: the stores of one fragment and the loads of the next
  // go to the same arrays (c_state is updated in place), so in program order hipcc must finish the stores before
  // the next loads - with 8 fragments per lane that was 8 serial load->store round trips (21 of the 67 us).
#pragma unroll
  for (int ni = 0; ni < Cfg::NI; ++ni) {
    const int u = u0 + tc.unit0 + ni * 16;
    if (u >= H) continue;
    int ln[Cfg::MI], rm[Cfg::MI];
    const int mi_n = wave_row_frags<Cfg>::of(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) / Cfg::WC);   // (uneven row split: the fragments this wave row owns)
#pragma unroll
    for (int mi = 0; mi < Cfg::MI; ++mi) {
      const int m = m0 + tc.row0 + mi * 16;
      const bool in = m < e.M && mi < mi_n;
      ln[mi] = in ? e.len[m] : -1;                     // -1: row outside the launch (nothing to do, not even zeros)
      rm[mi] = (in && e.row_map) ? e.row_map[m] : m;
    }
    float4 cv[Cfg::MI];
#pragma unroll
    for (int mi = 0; mi < Cfg::MI; ++mi) {
      cv[mi] = make_float4(0.f, 0.f, 0.f, 0.f);        // zero initial state (no memset of the state buffers)
      if (e.t > 0 && e.t < ln[mi]) cv[mi] = *(const float4*)(e.c_state + (long)rm[mi] * e.ld_state + u);   // running f32 cell state, in place
    }
    // stores: uniform base + 32-bit lane byte offset (a time slab is far below 4 GiB: checked by the launchers), policy EVC_FWD_STORE_POLICY
    constexpr int SP = EVC_FWD_STORE_POLICY;
#pragma unroll
    for (int mi = 0; mi < Cfg::MI; ++mi) {
      const int m = m0 + tc.row0 + mi * 16;
      if (ln[mi] < 0) continue;
      const uint32_t hu = (uint32_t)m * (uint32_t)H + (uint32_t)u;                      // element index in an [M][H] slab
      const uint32_t su4 = ((uint32_t)rm[mi] * (uint32_t)e.ld_state + (uint32_t)u) * 4u;   // byte offset in c_state / h_state
      // byte offset of this lane's 4 units in hout: FP8 rows are [f16(h) (H halfwords) | e4m3 (H bytes)] = 3H bytes, wide f16 rows 2H halfwords
      const uint32_t hw2 = FP8 ? (uint32_t)m * (uint32_t)(3 * H) + (uint32_t)u * 2u : (F16 && e.h_wide) ? ((uint32_t)m * (uint32_t)(2 * H) + (uint32_t)u) * 2u : hu * 2u;
      const uint32_t h8o = (uint32_t)m * (uint32_t)(3 * H) + (uint32_t)(2 * H) + (uint32_t)u;      // (FP8: the row's e4m3 part)
      const u32x2_t z2 = {0u, 0u};
      if (e.t >= ln[mi]) {          // dynamic_rnn: state copied through, zero output
        store8<SP>(e.hout, hw2, z2);
        if (FP8) store4<SP>(e.hout, h8o, 0u);
        else if (F16 && e.h_wide) store8<SP>(e.hout, hw2 + (uint32_t)H * 2u, z2);
        if (F16) store8<SP>(e.hout_lo, hu * 2u, z2);
        if (SPLIT) {
          const uint32_t wo = ((uint32_t)m * (uint32_t)(2 * H) + (uint32_t)u) * 2u;
          store8<SP>(e.hout_lo, wo, z2);
          store8<SP>(e.hout_lo, wo + (uint32_t)H * 2u, z2);
        }
        if (e.t == 0) {             // zero-length row: its final state is the zero initial state
          const u32x4_t z4 = {0u, 0u, 0u, 0u};
          store16<SP>(e.c_state, su4, z4);
          store16<SP>(e.h_state, su4, z4);
        }
        continue;
      }
      float zi[4], zj[4], zf[4], zo[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {                     // bias and forget_bias are already in the accumulators
        zi[r] = acc[mi][0][ni][r]; zj[r] = acc[mi][1][ni][r]; zf[r] = acc[mi][2][ni][r]; zo[r] = acc[mi][3][ni][r];
      }
      if (e.zx) {                   // hoisted x-projection (small-M stacks: one or two fragments per lane)
        const float* zr = e.zx + (long)m * e.ldzx + u;
        const float4 a = *(const float4*)zr, b = *(const float4*)(zr + H), c = *(const float4*)(zr + 2 * H), d = *(const float4*)(zr + 3 * H);
        zi[0] += a.x; zi[1] += a.y; zi[2] += a.z; zi[3] += a.w;
        zj[0] += b.x; zj[1] += b.y; zj[2] += b.z; zj[3] += b.w;
        zf[0] += c.x; zf[1] += c.y; zf[2] += c.z; zf[3] += c.w;
        zo[0] += d.x; zo[1] += d.y; zo[2] += d.z; zo[3] += d.w;
      }
      const float co[4] = {cv[mi].x, cv[mi].y, cv[mi].z, cv[mi].w};
      float cn[4], hn[4];
      uint2 rec[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gi = sigmoidf_(zi[r]), gj = tanhf_(zj[r]), gf = sigmoidf_(zf[r]), go = sigmoidf_(zo[r]);
        cn[r] = co[r] * gf + gi * gj;
        hn[r] = tanhf_(cn[r]) * go;
        rec[r] = make_uint2(pack_bf16x2(gi, gj), pack_bf16x2(gf, go));
      }
      const u32x4_t cnv = {__float_as_uint(cn[0]), __float_as_uint(cn[1]), __float_as_uint(cn[2]), __float_as_uint(cn[3])};
      store16<SP>(e.c_state, su4, cnv);               // rows stop updating at t = len: what stays is the returned state
      if (e.c_hist) store8<SP>(e.c_hist, hu * 2u, u32x2_t{pack_bf16x2(cn[0], cn[1]), pack_bf16x2(cn[2], cn[3])});
      if (e.t == ln[mi] - 1)
        store16<SP>(e.h_state, su4, u32x4_t{__float_as_uint(hn[0]), __float_as_uint(hn[1]), __float_as_uint(hn[2]), __float_as_uint(hn[3])});
      const u32x2_t hb = {pack_bf16x2(hn[0], hn[1]), pack_bf16x2(hn[2], hn[3])};
      if (F16) {
        const uint32_t p01 = pack_f16x2_hw(hn[0], hn[1]), p23 = pack_f16x2_hw(hn[2], hn[3]);
        store8<SP>(e.hout, hw2, u32x2_t{p01, p23});
        if (FP8) {                  // e4m3(h * 2^7): the activation operand of the weights' low-order halves (|h| < 1: no saturation)
          int w8 = __builtin_amdgcn_cvt_pk_fp8_f32(hn[0] * 128.0f, hn[1] * 128.0f, 0, false);
          w8 = __builtin_amdgcn_cvt_pk_fp8_f32(hn[2] * 128.0f, hn[3] * 128.0f, w8, true);
          store4<SP>(e.hout, h8o, (uint32_t)w8);
        } else if (e.h_wide) {      // f16(h)/64: the operand of the weights' low-order halves (scaled by 64)
          const float s0 = f16_to_f32((f16_t)(p01 & 0xffffu)) * (1.0f / 64.0f), s1 = f16_to_f32((f16_t)(p01 >> 16)) * (1.0f / 64.0f);
          const float s2 = f16_to_f32((f16_t)(p23 & 0xffffu)) * (1.0f / 64.0f), s3 = f16_to_f32((f16_t)(p23 >> 16)) * (1.0f / 64.0f);
          store8<SP>(e.hout, hw2 + (uint32_t)H * 2u, u32x2_t{pack_f16x2_hw(s0, s1), pack_f16x2_hw(s2, s3)});
        }
        store8<SP>(e.hout_lo, hu * 2u, hb);
      } else {
        store8<SP>(e.hout, hu * 2u, hb);
      }
      if (SPLIT) {
        float lo[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) lo[r] = hn[r] - bf16_to_f32(f32_to_bf16(hn[r]));
        const uint32_t wo = ((uint32_t)m * (uint32_t)(2 * H) + (uint32_t)u) * 2u;
        store8<SP>(e.hout_lo, wo, u32x2_t{pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3])});
        store8<SP>(e.hout_lo, wo + (uint32_t)H * 2u, hb);
      }
      if (e.gates) {                                   // 4 units x 8 bytes
        store16<SP>(e.gates, hu * 8u, u32x4_t{rec[0].x, rec[0].y, rec[1].x, rec[1].y});
        store16<SP>(e.gates, hu * 8u + 16u, u32x4_t{rec[2].x, rec[2].y, rec[3].x, rec[3].y});
      }
    }
  }
#ifdef EVC_STAMPS
  EVC_STAMP(p.stamp_slot, 3);
  wait_vmcnt<0>();                       // this wave's stores acknowledged
  EVC_STAMP(p.stamp_slot, 4);
  __syncthreads();
  EVC_STAMP(p.stamp_slot, 5);
#endif
}
#ifdef EVC_STAMPS
extern "C" int evc_debug_read_stamps(unsigned long long* out) {     // out: [8][512][8]
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(evc_stamps), sizeof(unsigned long long) * 8 * 512 * 8) == hipSuccess ? 0 : 1;
}
#endif

template <class Cfg, bool SPLIT = false, bool F16 = false, bool FP8 = false>
__global__ __launch_bounds__(Cfg::NT) void lstm_fwd_step_kernel(GemmOperands p, LstmFwdParams e, int tiles_m, int tiles_n) {
  lstm_fwd_step_body<Cfg, SPLIT, F16, FP8>(p, e, tiles_m, tiles_n, blockIdx.x);
}

// Two independent steps of the same geometry in one launch (the first tiles_m*tiles_n workgroups run step a, the
// rest step b): layer 0 at time t+1 and layer 1 at time t of a two-layer stack with M ~ batch rows - those steps are
// latency-bound (12 us for 2 GFLOP), so the pair costs about what one of them does and the stack's chain of
// dependent launches is T+1 long instead of 2T (evc_lstm_stack2_fwd).
template <class Cfg, bool F16 = false, bool FP8 = false>
__global__ __launch_bounds__(Cfg::NT) void lstm_fwd_pair_kernel(GemmOperands pa, LstmFwdParams ea, GemmOperands pb, LstmFwdParams eb,
                                                                int tiles_m, int tiles_n) {
  const int n = tiles_m * tiles_n;
  const bool first = blockIdx.x < n;                   // workgroup-uniform: scalar selects of the two argument sets
  const GemmOperands p = first ? pa : pb;
  const LstmFwdParams e = first ? ea : eb;
  lstm_fwd_step_body<Cfg, false, F16, FP8>(p, e, tiles_m, tiles_n, first ? blockIdx.x : blockIdx.x - n);
}

typedef TileCfg<128, 4, 32, 2, 2> CfgLstmBig;    // 128 rows x 32 units x 4 gates
typedef TileCfg<64, 4, 16, 4, 1> CfgLstmSmall;   // 64 rows x 16 units x 4 gates (M ~ 256 steps)
// v2 tiles: BM rows x 64 units x 4 gates (256 accumulator columns).  The row count of a step varies with the
// batch (row plans drop the padding rows), so the tile height is chosen per launch to cut the active rows
// into a multiple of 256 workgroups: 5120 rows -> 320, ~3600 -> 224, ...
typedef TileCfg2<320, 4, 64, 2, 4, 4, false> CfgLstmV2a;
typedef TileCfg2<288, 4, 64, 2, 4, 4, false> CfgLstmV2_288;
typedef TileCfg2<256, 4, 64, 2, 4, 5, true> CfgLstmV2b;
typedef TileCfg2<224, 4, 64, 2, 4, 5, true> CfgLstmV2_224;
// The tall forward tiles on 64-wide K stages (gemm_core_v3.h): two stages of 60-64 KB instead of five of 30-32 KB - whole cache
// lines per LDS-DMA piece and one barrier per 64 K columns beat the deeper ring (same-box A/B: 79.0 -> 73.9 us per step)
typedef TileCfg3<256, 4, 64, 2, 4, 2> CfgLstmV3_256;
typedef TileCfg3<224, 4, 64, 2, 4, 2> CfgLstmV3_224;
// 240 rows = 7 row fragments on the producer waves + 8 on their SIMD partners (gemm_core_v3.h, uneven split): 3 585-3 840 live rows are 16 row tiles
// = 256 workgroups of 240 rows instead of 15 x 16 = 240 workgroups of 256 rows (round 4)
typedef TileCfg3<240, 4, 64, 2, 4, 2, 7> CfgLstmV3_240;
typedef TileCfg3<224, 4, 64, 2, 4, 2, 6> CfgLstmV3_224u;      // 6 + 8 instead of 7 + 7: the producer waves issue the LDS-DMA, their partners take the extra row fragment
typedef TileCfg3<192, 4, 64, 2, 4, 2> CfgLstmV3_192;
typedef TileCfg3<160, 4, 64, 2, 4, 3> CfgLstmV3_160;
typedef TileCfg2<192, 4, 64, 2, 4, 5, true> CfgLstmV2_192;
typedef TileCfg2<160, 4, 64, 2, 4, 5, true> CfgLstmV2_160;
typedef TileCfg2<128, 4, 64, 2, 4, 5, true> CfgLstmV2_128;
typedef TileCfg2<64, 4, 64, 2, 4, 5, true> CfgLstmV2_64;
typedef TileCfg2<64, 4, 16, 4, 1, 5, true> CfgLstmV2Small;
typedef TileCfg3<64, 4, 16, 4, 1, 4> CfgLstmV3Small;         // the same tile on 64-wide K stages (64 KB of LDS: still two workgroups per CU)   // 64 rows x 16 units x 4 gates on the ring loop, 4 waves, 40 KB: M ~ batch steps

template <class Cfg, bool SPLIT = false, bool F16 = false, bool FP8 = false>
static inline void launch_lstm_fwd(GemmOperands p, const LstmFwdParams& e, int k1, int k2, hipStream_t st) {
  p.nk1 = k1 / kdiv<Cfg>(); p.nk2 = k2 / kdiv<Cfg>();
#ifdef EVC_STAMPS
  p.stamp_slot = e.t & 7;
#endif
  const int tm = ceil_div(e.M, Cfg::BM), tn = ceil_div(e.H, Cfg::BU);
  launch_cfg<Cfg>(lstm_fwd_step_kernel<Cfg, SPLIT, F16, FP8>, tm * tn, st, p, e, tm, tn);
}

// forward tile for a step over `rows` rows: index into {320, 288, 256, 224, 192, 160 (v2), 128 (v1), 64 (v1), 128 (v2), 64 (v2)}
static inline int pick_fwd_tile(int rows, int H) {
  constexpr int NC = 11;
  static const int bm[NC] = {320, 288, 256, 224, 192, 160, 128, 64, 128, 64, 240};
  static const int bn[NC] = {256, 256, 256, 256, 256, 256, 128, 64, 256, 256, 256};
  static const int bu[NC] = {64, 64, 64, 64, 64, 64, 32, 16, 64, 64, 64};
  static const double cf[NC] = {1.0, 1.0, 1.0, 1.02, 1.04, 1.08, 1.3, 2.6, 1.15, 1.5, 1.01};   // smaller tiles: less efficient per flop
  static const bool no240 = getenv("EVC_FWD_NO_240") != nullptr;      // A/B: the tile set of round 3
  int best = 0;
  double bc = 1e300;
  for (int i = 0; i < NC; ++i) {
    if (i == 10 && no240) continue;
    const double c = tile_cost((long)ceil_div(rows, bm[i]) * ceil_div(H, bu[i]), bm[i], bn[i], 1, cf[i]);
    if (c < bc) { bc = c; best = i; }
  }
  const int f = forced_tile();        // debug: 1 -> 256, 2 -> v1 128, 3 -> v1 64, 4 -> 320, 5 -> 288, 6 -> 224, 7 -> 192, 8 -> 160, 9 -> v2 128, 10 -> v2 64, 11 -> 240
  if (f) { static const int map[12] = {0, 2, 6, 7, 0, 1, 3, 4, 5, 8, 9, 10}; best = map[f < 12 ? f : 0]; }
  return best;
}

static int lstm_layer_fwd_impl(const evc_bf16* x, const evc_bf16* wT, const float* bias, const int32_t* len,
                               int T, int M, int Kin, int H, int hoist, float* zx_ws,
                               evc_bf16* hbuf, float* c_state, float* h_state, int64_t ld_state,
                               void* gates, evc_bf16* c_all, evc_bf16* hbuf_bf16,
                               const int32_t* row_map, const int32_t* rows_per_step, void* stream, int f16 = 0, int64_t ldx = 0, int h_wide = 0);

extern "C" int evc_lstm_layer_fwd(const evc_bf16* x, const evc_bf16* wT, const float* bias, const int32_t* len,
                                  int T, int M, int Kin, int H, int hoist, float* zx_ws,
                                  evc_bf16* hbuf, float* c_state, float* h_state, int64_t ld_state,
                                  void* gates, evc_bf16* c_all, const int32_t* row_map, const int32_t* rows_per_step,
                                  void* stream) {
  return lstm_layer_fwd_impl(x, wT, bias, len, T, M, Kin, H, hoist, zx_ws, hbuf, c_state, h_state, ld_state, gates, c_all,
                             nullptr, row_map, rows_per_step, stream);
}

extern "C" int evc_lstm_layer_fwd_f16(const evc_f16* x, int64_t ldx, const evc_f16* wT, const float* bias, const int32_t* len,
                                      int T, int M, int Kin, int H, evc_f16* hbuf, int h_wide, evc_bf16* hbuf_bf16,
                                      float* c_state, float* h_state, int64_t ld_state, void* gates, evc_bf16* c_all,
                                      const int32_t* row_map, const int32_t* rows_per_step, void* stream) {
  EVC_REQUIRE(hbuf_bf16, EVC_ERR_BAD_ARG, "evc_lstm_layer_fwd_f16: hbuf_bf16 (the bf16 copy of h for the backward pass) is required");
  EVC_REQUIRE(((uintptr_t)hbuf_bf16 % 8) == 0, EVC_ERR_BAD_ALIGN, "evc_lstm_layer_fwd_f16: hbuf_bf16 must be 8-byte aligned");
  EVC_REQUIRE(ldx >= Kin && ldx % 8 == 0 && (h_wide == 0 || h_wide == 1), EVC_ERR_BAD_ARG, "evc_lstm_layer_fwd_f16: ldx=%ld (>= Kin=%d, %%8), h_wide=%d",
              (long)ldx, Kin, h_wide);
  return lstm_layer_fwd_impl((const evc_bf16*)x, (const evc_bf16*)wT, bias, len, T, M, Kin, H, 0, nullptr, (evc_bf16*)hbuf, c_state, h_state,
                             ld_state, gates, c_all, hbuf_bf16, row_map, rows_per_step, stream, 1, ldx, h_wide);
}

static int lstm_layer_fwd_impl(const evc_bf16* x, const evc_bf16* wT, const float* bias, const int32_t* len,
                               int T, int M, int Kin, int H, int hoist, float* zx_ws,
                               evc_bf16* hbuf, float* c_state, float* h_state, int64_t ld_state,
                               void* gates, evc_bf16* c_all, evc_bf16* hbuf_bf16,
                               const int32_t* row_map, const int32_t* rows_per_step, void* stream, int f16, int64_t ldx, int h_wide) {
  // f16: x, wT, hbuf hold IEEE f16 (16-bit containers), hbuf_bf16 receives the bf16 copy of every h_t; ldx = row stride of x
  // (0: Kin); h_wide: hbuf rows are [h | h/64] (2H) and the kernel's h-part is [Wh | Wh_lo*64] (2H): the recurrent weights
  // K-extended by their low-order halves.  (The split-bf16 form of a layer is evc_lstm_layer_fwd_hp below.)
  if (ldx == 0) ldx = Kin;
  const long ldh = h_wide ? 2L * H : H;            // row stride of hbuf = K of the recurrent part
  EVC_REQUIRE(T > 0 && M > 0 && H > 0 && Kin > 0, EVC_ERR_BAD_SHAPE, "evc_lstm_layer_fwd: bad shape");
  EVC_REQUIRE(Kin % 64 == 0 && H % 64 == 0, EVC_ERR_BAD_SHAPE,
              "evc_lstm_layer_fwd: Kin=%d and H=%d must be multiples of 64", Kin, H);
  EVC_REQUIRE(ring_operand_ok(M, ldx > ldh ? ldx : ldh) && ring_operand_ok(4L * H, (long)Kin + ldh), EVC_ERR_BAD_SHAPE,
              "evc_lstm_layer_fwd: a time slab or the kernel spans 4 GiB or more (M=%d Kin=%d H=%d)", M, Kin, H);
  EVC_REQUIRE(fwd_tail_ok(M, H, ld_state), EVC_ERR_BAD_SHAPE, "LSTM forward: a gate-record / state slab spans 4 GiB or more (M=%d H=%d ld_state=%ld)", M, H, (long)ld_state);
  EVC_REQUIRE(!hoist || (zx_ws && !h_wide && ldx == Kin), EVC_ERR_BAD_ARG, "evc_lstm_layer_fwd: hoist needs zx_ws (and plain operands)");
  EVC_REQUIRE(ld_state % 4 == 0 && ((uintptr_t)c_state % 16) == 0 && ((uintptr_t)h_state % 16) == 0 && ((uintptr_t)bias % 16) == 0 &&
              ((uintptr_t)hbuf % 8) == 0, EVC_ERR_BAD_ALIGN, "evc_lstm_layer_fwd: state/bias/hbuf must allow 16-byte vector access");
  EVC_REQUIRE((gates == nullptr) == (c_all == nullptr), EVC_ERR_BAD_ARG, "evc_lstm_layer_fwd: gates and c_all go together");
  EVC_REQUIRE(!gates || (((uintptr_t)gates % 16) == 0 && ((uintptr_t)c_all % 8) == 0), EVC_ERR_BAD_ALIGN,
              "evc_lstm_layer_fwd: gates must be 16-byte, c_all 8-byte aligned");
  if (rows_per_step)
    for (int t = 0; t < T; ++t)
      EVC_REQUIRE(rows_per_step[t] >= 0 && rows_per_step[t] <= M && (t == 0 || rows_per_step[t] <= rows_per_step[t - 1]), EVC_ERR_BAD_ARG,
                  "evc_lstm_layer_fwd: rows_per_step[%d]=%d must be non-increasing and within [0, M=%d]", t, rows_per_step[t], M);
  hipStream_t st = (hipStream_t)stream;
  const long ldw = Kin + ldh;
  // h_{-1} = 0 (the state buffers need no clearing: step 0 treats c_old as 0 and writes the zero
  // state of the zero-length rows it covers itself; rows beyond rows_per_step[0] are the caller's)
  EVC_CHECK_HIP(hipMemsetAsync(hbuf, 0, (size_t)M * ldh * sizeof(bf16_t), st));
  if (f16) EVC_CHECK_HIP(hipMemsetAsync(hbuf_bf16, 0, (size_t)M * H * sizeof(bf16_t), st));
  if (hoist) {
    int rc = evc_gemm_nt(x, Kin, wT, ldw, zx_ws, 4L * H, T * M, 4 * H, Kin, nullptr, 0, 0, stream);
    if (rc) return rc;
  }
  for (int t = 0; t < T; ++t) {
    const int Mt = rows_per_step ? rows_per_step[t] : M;     // active rows are the prefix [0, Mt) (row plan)
    if (Mt == 0) break;
    GemmOperands p;
    p.M = Mt; p.Nu = H; p.group_stride = H; p.ldb = ldw; p.nk1 = p.nk2 = 0;
    p.A1lo = p.A2lo = p.Blo = nullptr;
    const bf16_t* hprev = hbuf + (long)t * M * ldh;
    int k1, k2;
    if (hoist) {
      p.A1 = hprev; p.lda1 = H; k1 = (t == 0) ? 0 : H; p.A2 = hprev; p.lda2 = H; k2 = 0;
      p.B = wT + Kin;
    } else {
      p.A1 = x + (long)t * M * ldx; p.lda1 = ldx; k1 = Kin;
      p.A2 = hprev; p.lda2 = ldh; k2 = (t == 0) ? 0 : (int)ldh;
      p.B = wT;
    }
    LstmFwdParams e;
    e.zx = hoist ? zx_ws + (long)t * M * 4 * H : nullptr; e.ldzx = 4L * H;
    e.bias = bias; e.len = len; e.t = t;
    e.c_state = c_state; e.h_state = h_state; e.ld_state = ld_state;
    e.hout = hbuf + (long)(t + 1) * M * ldh; e.h_wide = h_wide;
    e.hout_lo = f16 ? hbuf_bf16 + (long)(t + 1) * M * H : nullptr;
    e.gates = gates ? (uint2*)gates + (long)t * M * H : nullptr;
    e.c_hist = c_all ? c_all + (long)(t + 1) * M * H : nullptr;      // slab t+1 = c after step t
    e.row_map = row_map;
    e.M = Mt; e.H = H;
    static const bool uneven224 = getenv("EVC_FWD_EVEN_224") == nullptr;      // the 224-row tile as 6 + 8 row fragments (A/B switch: 7 + 7; 58.3 -> 58.0 us per launch)
    if (f16) {        // IEEE f16 operands, one MFMA product per depth: the tiles of the bf16 step
      switch (pick_fwd_tile(Mt, H)) {
        case 0: launch_lstm_fwd<CfgLstmV2a, false, true>(p, e, k1, k2, st); break;
        case 1: launch_lstm_fwd<CfgLstmV2_288, false, true>(p, e, k1, k2, st); break;
        case 2: launch_lstm_fwd<CfgLstmV3_256, false, true>(p, e, k1, k2, st); break;
        case 3: if (uneven224) launch_lstm_fwd<CfgLstmV3_224u, false, true>(p, e, k1, k2, st); else launch_lstm_fwd<CfgLstmV3_224, false, true>(p, e, k1, k2, st); break;
        case 10: launch_lstm_fwd<CfgLstmV3_240, false, true>(p, e, k1, k2, st); break;
        case 4: launch_lstm_fwd<CfgLstmV3_192, false, true>(p, e, k1, k2, st); break;
        case 5: launch_lstm_fwd<CfgLstmV3_160, false, true>(p, e, k1, k2, st); break;
        case 6: launch_lstm_fwd<CfgLstmBig, false, true>(p, e, k1, k2, st); break;
        case 8: launch_lstm_fwd<CfgLstmV2_128, false, true>(p, e, k1, k2, st); break;
        case 9: launch_lstm_fwd<CfgLstmV2_64, false, true>(p, e, k1, k2, st); break;
        default: launch_lstm_fwd<CfgLstmSmall, false, true>(p, e, k1, k2, st); break;
      }
      continue;
    }
    static const bool fwd_v2 = getenv("EVC_FWD_V2_LOOP") != nullptr;      // A/B: the 32-wide K stages for the 160-256-row tiles
    switch (pick_fwd_tile(Mt, H)) {
      case 0: launch_lstm_fwd<CfgLstmV2a>(p, e, k1, k2, st); break;
      case 1: launch_lstm_fwd<CfgLstmV2_288>(p, e, k1, k2, st); break;
      case 2: if (fwd_v2) launch_lstm_fwd<CfgLstmV2b>(p, e, k1, k2, st); else launch_lstm_fwd<CfgLstmV3_256>(p, e, k1, k2, st); break;
      case 3: if (fwd_v2) launch_lstm_fwd<CfgLstmV2_224>(p, e, k1, k2, st); else if (uneven224) launch_lstm_fwd<CfgLstmV3_224u>(p, e, k1, k2, st); else launch_lstm_fwd<CfgLstmV3_224>(p, e, k1, k2, st); break;
      case 10: launch_lstm_fwd<CfgLstmV3_240>(p, e, k1, k2, st); break;
      case 4: if (fwd_v2) launch_lstm_fwd<CfgLstmV2_192>(p, e, k1, k2, st); else launch_lstm_fwd<CfgLstmV3_192>(p, e, k1, k2, st); break;
      case 5: if (fwd_v2) launch_lstm_fwd<CfgLstmV2_160>(p, e, k1, k2, st); else launch_lstm_fwd<CfgLstmV3_160>(p, e, k1, k2, st); break;
      case 6: launch_lstm_fwd<CfgLstmBig>(p, e, k1, k2, st); break;
      case 8: launch_lstm_fwd<CfgLstmV2_128>(p, e, k1, k2, st); break;
      case 9: launch_lstm_fwd<CfgLstmV2_64>(p, e, k1, k2, st); break;
      default: launch_lstm_fwd<CfgLstmSmall>(p, e, k1, k2, st); break;
    }
  }
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// forward tile among the 64-wide ring tiles only (the e4m3 tail lives in gemm_core_v3.h): 0..3 = 256 / 224 / 192 / 160 rows
static inline int pick_fwd_tile_v3(int rows, int H) {
  static const int bm[5] = {256, 224, 192, 160, 240};
  static const double cf[5] = {1.0, 1.02, 1.04, 1.08, 1.01};
  static const bool no240 = getenv("EVC_FWD_NO_240") != nullptr;
  int best = 0;
  double bc = 1e300;
  for (int i = 0; i < (no240 ? 4 : 5); ++i) {
    const double c = tile_cost((long)ceil_div(rows, bm[i]) * ceil_div(H, 64), bm[i], 256, 1, cf[i]);
    if (c < bc) { bc = c; best = i; }
  }
  const int f = forced_tile();        // debug: 1 -> 256, 6 -> 224, 7 -> 192, 8 -> 160, 11 -> 240
  if (f == 1) best = 0; else if (f == 6) best = 1; else if (f == 7) best = 2; else if (f == 8) best = 3; else if (f == 11) best = 4;
  return best;
}

// "High" precision L1 layer with the weights' low-order halves contracted in fp8: per step
//   z = [x16 | h16] . [W16x | W16h]^T  (IEEE f16, v_mfma_f32_16x16x32_f16)  +  2^-(7 + w8_scale_exp) [x8 | h8] . [W8x | W8h]^T  (OCP e4m3,
//   v_mfma_scale_f32_16x16x128_f8f6f4: per K element twice the MFMA rate)
// with W8 = e4m3((W - f16(W)) 2^w8_scale_exp) (evc_cast_f32_to_fp8_lo), x8 = e4m3(x 2^7) and h8 = e4m3(h 2^7): the weights are exact to
// ~2^-15 relative instead of f16's 2^-11, for half the MFMA time of K-extending them by f16 low-order halves.  x rows: kx16 halfwords at
// the row start (any K-extension of the input the caller likes, against the first kx16 columns of wT16) and kx8 e4m3 bytes at byte
// offset x8_off of the same row (row stride ldx halfwords); hbuf rows [T+1][M]: [f16(h_t) (H halfwords) | e4m3(h_t 2^7) (H bytes)] (3H
// bytes: what the next layer takes as its x rows with kx16 = H, x8_off = 2H, kx8 = H); wT16 [4H][kx16 + H] f16, wT8 [4H][kx8 + H] bytes.
extern "C" int evc_lstm_layer_fwd_f16_fp8lo(const evc_f16* x, int64_t ldx, int kx16, int64_t x8_off, int kx8, const evc_f16* wT16,
                                            const uint8_t* wT8, int w8_scale_exp, const float* bias, const int32_t* len,
                                            int T, int M, int H, evc_f16* hbuf, evc_bf16* hbuf_bf16, float* c_state, float* h_state,
                                            int64_t ld_state, void* gates, evc_bf16* c_all, const int32_t* row_map,
                                            const int32_t* rows_per_step, void* stream) {
  EVC_REQUIRE(T > 0 && M > 0 && H > 0 && kx16 > 0 && kx8 > 0, EVC_ERR_BAD_SHAPE, "evc_lstm_layer_fwd_f16_fp8lo: bad shape");
  EVC_REQUIRE(kx16 % 64 == 0 && H % 128 == 0 && kx8 % 128 == 0 && kx8 >= 384, EVC_ERR_BAD_SHAPE,
              "evc_lstm_layer_fwd_f16_fp8lo: kx16=%d (%%64), H=%d (%%128), kx8=%d (%%128, >= 384: the ring must be full of e4m3 stages at t = 0)", kx16, H, kx8);
  EVC_REQUIRE(x && wT16 && wT8 && hbuf && hbuf_bf16 && bias && len, EVC_ERR_BAD_ARG, "evc_lstm_layer_fwd_f16_fp8lo: NULL operand");
  EVC_REQUIRE(ldx % 8 == 0 && x8_off % 16 == 0 && ldx >= kx16 && ldx * 2 >= x8_off + kx8 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)wT16 % 16) == 0 &&
              ((uintptr_t)wT8 % 16) == 0 && ((uintptr_t)hbuf % 16) == 0 && ((uintptr_t)hbuf_bf16 % 8) == 0, EVC_ERR_BAD_ALIGN,
              "evc_lstm_layer_fwd_f16_fp8lo: ldx=%ld (%%8), x8_off=%ld (%%16), 16-byte aligned operands", (long)ldx, (long)x8_off);
  EVC_REQUIRE(w8_scale_exp >= 0 && w8_scale_exp <= 60, EVC_ERR_BAD_ARG, "evc_lstm_layer_fwd_f16_fp8lo: w8_scale_exp=%d", w8_scale_exp);
  const long ldh = 3L * H / 2;                       // halfwords per hbuf row
  EVC_REQUIRE(ring_operand_ok(M, ldx > ldh ? ldx : ldh) && ring_operand_ok(4L * H, (long)kx16 + H), EVC_ERR_BAD_SHAPE,
              "evc_lstm_layer_fwd_f16_fp8lo: a time slab or the kernel spans 4 GiB or more");
  EVC_REQUIRE(fwd_tail_ok(M, H, ld_state), EVC_ERR_BAD_SHAPE, "LSTM forward: a gate-record / state slab spans 4 GiB or more (M=%d H=%d ld_state=%ld)", M, H, (long)ld_state);
  EVC_REQUIRE(ld_state % 4 == 0 && ((uintptr_t)c_state % 16) == 0 && ((uintptr_t)h_state % 16) == 0 && ((uintptr_t)bias % 16) == 0, EVC_ERR_BAD_ALIGN,
              "evc_lstm_layer_fwd_f16_fp8lo: state/bias must allow 16-byte vector access");
  EVC_REQUIRE((gates == nullptr) == (c_all == nullptr), EVC_ERR_BAD_ARG, "evc_lstm_layer_fwd_f16_fp8lo: gates and c_all go together");
  EVC_REQUIRE(!gates || (((uintptr_t)gates % 16) == 0 && ((uintptr_t)c_all % 8) == 0), EVC_ERR_BAD_ALIGN,
              "evc_lstm_layer_fwd_f16_fp8lo: gates must be 16-byte, c_all 8-byte aligned");
  if (rows_per_step)
    for (int t = 0; t < T; ++t)
      EVC_REQUIRE(rows_per_step[t] >= 0 && rows_per_step[t] <= M && (t == 0 || rows_per_step[t] <= rows_per_step[t - 1]), EVC_ERR_BAD_ARG,
                  "evc_lstm_layer_fwd_f16_fp8lo: rows_per_step[%d]=%d must be non-increasing and within [0, M=%d]", t, rows_per_step[t], M);
  hipStream_t st = (hipStream_t)stream;
  const bf16_t* xb = (const bf16_t*)x;
  bf16_t* hb = (bf16_t*)hbuf;
  EVC_CHECK_HIP(hipMemsetAsync(hb, 0, (size_t)M * ldh * sizeof(bf16_t), st));            // h_{-1} = 0 (both parts of the rows)
  EVC_CHECK_HIP(hipMemsetAsync(hbuf_bf16, 0, (size_t)M * H * sizeof(bf16_t), st));
  for (int t = 0; t < T; ++t) {
    const int Mt = rows_per_step ? rows_per_step[t] : M;
    if (Mt == 0) break;
    GemmOperands p;
    p.M = Mt; p.Nu = H; p.group_stride = H; p.nk1 = p.nk2 = 0;
    p.A1lo = p.A2lo = p.Blo = nullptr;
    const bf16_t* xt = xb + (long)t * M * ldx;
    const bf16_t* hprev = hb + (long)t * M * ldh;
    p.A1 = xt; p.lda1 = ldx;
    p.A2 = hprev; p.lda2 = ldh;
    p.B = (const bf16_t*)wT16; p.ldb = (long)kx16 + H;
    p.A3 = (const uint8_t*)xt + x8_off; p.lda3 = ldx * 2; p.nk3 = kx8 / 128;
    p.A4 = (const uint8_t*)(hprev + H); p.lda4 = ldh * 2; p.nk4 = t == 0 ? 0 : H / 128;
    p.B8 = wT8; p.ldb8 = (long)kx8 + H;
    p.scale8_exp = -(7 + w8_scale_exp);
    const int k1 = kx16, k2 = t == 0 ? 0 : H;
    LstmFwdParams e;
    e.zx = nullptr; e.ldzx = 0;
    e.bias = bias; e.len = len; e.t = t;
    e.c_state = c_state; e.h_state = h_state; e.ld_state = ld_state;
    e.hout = hb + (long)(t + 1) * M * ldh; e.h_wide = 2;
    e.hout_lo = hbuf_bf16 + (long)(t + 1) * M * H;
    e.gates = gates ? (uint2*)gates + (long)t * M * H : nullptr;
    e.c_hist = c_all ? c_all + (long)(t + 1) * M * H : nullptr;
    e.row_map = row_map;
    e.M = Mt; e.H = H;
    switch (pick_fwd_tile_v3(Mt, H)) {
      case 0: launch_lstm_fwd<CfgLstmV3_256, false, true, true>(p, e, k1, k2, st); break;
      case 1: launch_lstm_fwd<CfgLstmV3_224, false, true, true>(p, e, k1, k2, st); break;
      case 2: launch_lstm_fwd<CfgLstmV3_192, false, true, true>(p, e, k1, k2, st); break;
      case 4: launch_lstm_fwd<CfgLstmV3_240, false, true, true>(p, e, k1, k2, st); break;
      default: launch_lstm_fwd<CfgLstmV3_160, false, true, true>(p, e, k1, k2, st); break;
    }
  }
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// "High" precision layer for the M ~ batch stacks (the L2 level): split-bf16 operands, f32-operand accuracy, as K-extensions of
// the plain loops (see evc_gemm_nt_split).  x-projection of all T steps hoisted into one split product; step t contracts
// [lo(h) | hi(h)] . [Wh_hi | Wh_lo]^T + hi(h) . Wh_hi^T (K = 3H) and writes h_t three times: hbuf (plain bf16 = the hi half, what
// the backward products read) and the wide image hbuf_lohi for the next step / the next layer's x-projection.
extern "C" int evc_lstm_layer_fwd_hp(const evc_bf16* x_lohi, const evc_bf16* wx_hilo, int64_t ldwx, const evc_bf16* wh_hilo, int64_t ldwh,
                                     const float* bias, const int32_t* len, int T, int M, int Kin, int H, float* zx_ws,
                                     evc_bf16* hbuf, evc_bf16* hbuf_lohi, float* c_state, float* h_state, int64_t ld_state,
                                     void* gates, evc_bf16* c_all, void* stream) {
  EVC_REQUIRE(T > 0 && M > 0 && H > 0 && Kin > 0 && Kin % 64 == 0 && H % 64 == 0, EVC_ERR_BAD_SHAPE,
              "evc_lstm_layer_fwd_hp: bad shape T=%d M=%d Kin=%d H=%d (Kin, H multiples of 64)", T, M, Kin, H);
  EVC_REQUIRE(x_lohi && wx_hilo && wh_hilo && zx_ws && hbuf && hbuf_lohi, EVC_ERR_BAD_ARG, "evc_lstm_layer_fwd_hp: NULL operand");
  EVC_REQUIRE(ldwx >= 2L * Kin && ldwh >= 2L * H && ldwx % 8 == 0 && ldwh % 8 == 0 && ((uintptr_t)wh_hilo % 16) == 0, EVC_ERR_BAD_ALIGN,
              "evc_lstm_layer_fwd_hp: weight images are [4H][2Kin] / [4H][2H] (ldwx=%ld ldwh=%ld)", (long)ldwx, (long)ldwh);
  EVC_REQUIRE(ring_operand_ok(M, 2L * H) && ring_operand_ok(4L * H, ldwh), EVC_ERR_BAD_SHAPE, "evc_lstm_layer_fwd_hp: operand spans 4 GiB or more");
  EVC_REQUIRE(fwd_tail_ok(M, H, ld_state), EVC_ERR_BAD_SHAPE, "LSTM forward: a gate-record / state slab spans 4 GiB or more (M=%d H=%d ld_state=%ld)", M, H, (long)ld_state);
  EVC_REQUIRE(ld_state % 4 == 0 && ((uintptr_t)c_state % 16) == 0 && ((uintptr_t)h_state % 16) == 0 && ((uintptr_t)bias % 16) == 0 &&
              ((uintptr_t)hbuf % 8) == 0 && ((uintptr_t)hbuf_lohi % 16) == 0, EVC_ERR_BAD_ALIGN,
              "evc_lstm_layer_fwd_hp: state/bias/hbuf must allow 16-byte vector access");
  EVC_REQUIRE((gates == nullptr) == (c_all == nullptr), EVC_ERR_BAD_ARG, "evc_lstm_layer_fwd_hp: gates and c_all go together");
  hipStream_t st = (hipStream_t)stream;
  EVC_CHECK_HIP(hipMemsetAsync(hbuf, 0, (size_t)M * H * sizeof(bf16_t), st));               // h_{-1} = 0
  EVC_CHECK_HIP(hipMemsetAsync(hbuf_lohi, 0, (size_t)M * 2 * H * sizeof(bf16_t), st));
  int rc = evc_gemm_nt_split(x_lohi, 2L * Kin, wx_hilo, ldwx, zx_ws, 4L * H, T * M, 4 * H, Kin, nullptr, stream);
  if (rc) return rc;
  for (int t = 0; t < T; ++t) {
    GemmOperands p;
    p.M = M; p.Nu = H; p.group_stride = H; p.ldb = ldwh; p.nk1 = p.nk2 = 0;
    p.A1lo = p.A2lo = p.Blo = nullptr;
    const bf16_t* hw = hbuf_lohi + (long)t * M * 2 * H;
    p.A1 = hw; p.lda1 = 2L * H; p.A2 = hw + H; p.lda2 = 2L * H;
    p.B = wh_hilo; p.B2 = wh_hilo;
    const int k1 = (t == 0) ? 0 : 2 * H, k2 = (t == 0) ? 0 : H;
    LstmFwdParams e;
    e.zx = zx_ws + (long)t * M * 4 * H; e.ldzx = 4L * H;
    e.bias = bias; e.len = len; e.t = t;
    e.c_state = c_state; e.h_state = h_state; e.ld_state = ld_state;
    e.hout = hbuf + (long)(t + 1) * M * H;
    e.hout_lo = hbuf_lohi + (long)(t + 1) * M * 2 * H;
    e.gates = gates ? (uint2*)gates + (long)t * M * H : nullptr;
    e.c_hist = c_all ? c_all + (long)(t + 1) * M * H : nullptr;
    e.row_map = nullptr;
    e.M = M; e.H = H;
    if (M >= 1024) launch_lstm_fwd<CfgLstmV3_256, true>(p, e, k1, k2, st);
    else launch_lstm_fwd<CfgLstmV3Small, true>(p, e, k1, k2, st);
  }
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// One layer's view of a stack for evc_lstm_stack2_fwd
struct FwdLayer {
  const bf16_t* x; int Kin;            // [T][M][Kin] input (unused when the x-projection is hoisted)
  const bf16_t* wT; const float* bias;
  const float* zx;                     // hoisted x-projection [T][M][4H] or NULL
  bf16_t* hbuf; float* c_state; float* h_state;
  void* gates; bf16_t* c_all;
};

static inline void fwd_step_args(const FwdLayer& L, const int32_t* len, int t, int M, int H, int64_t ld_state,
                                 GemmOperands& p, LstmFwdParams& e, int& k1, int& k2) {
  p.M = M; p.Nu = H; p.group_stride = H; p.ldb = L.Kin + H; p.nk1 = p.nk2 = 0;
  p.A1lo = p.A2lo = p.Blo = nullptr;
  const bf16_t* hprev = L.hbuf + (long)t * M * H;
  if (L.zx) {
    p.A1 = hprev; p.lda1 = H; k1 = (t == 0) ? 0 : H; p.A2 = hprev; p.lda2 = H; k2 = 0;
    p.B = L.wT + L.Kin;
  } else {
    p.A1 = L.x + (long)t * M * L.Kin; p.lda1 = L.Kin; k1 = L.Kin;
    p.A2 = hprev; p.lda2 = H; k2 = (t == 0) ? 0 : H;
    p.B = L.wT;
  }
  e.zx = L.zx ? L.zx + (long)t * M * 4 * H : nullptr; e.ldzx = 4L * H;
  e.bias = L.bias; e.len = len; e.t = t;
  e.c_state = L.c_state; e.h_state = L.h_state; e.ld_state = ld_state;
  e.hout = L.hbuf + (long)(t + 1) * M * H;
  e.hout_lo = nullptr;
  e.gates = L.gates ? (uint2*)L.gates + (long)t * M * H : nullptr;
  e.c_hist = L.c_all ? L.c_all + (long)(t + 1) * M * H : nullptr;
  e.row_map = nullptr;
  e.M = M; e.H = H;
}

template <class Cfg, bool F16 = false, bool FP8 = false>
static inline void launch_lstm_fwd_pair(GemmOperands pa, const LstmFwdParams& ea, int k1a, int k2a,
                                        GemmOperands pb, const LstmFwdParams& eb, int k1b, int k2b, hipStream_t st) {
  pa.nk1 = k1a / kdiv<Cfg>(); pa.nk2 = k2a / kdiv<Cfg>();
  pb.nk1 = k1b / kdiv<Cfg>(); pb.nk2 = k2b / kdiv<Cfg>();
  const int tm = ceil_div(ea.M, Cfg::BM), tn = ceil_div(ea.H, Cfg::BU);
  launch_cfg<Cfg>(lstm_fwd_pair_kernel<Cfg, F16, FP8>, 2 * tm * tn, st, pa, ea, pb, eb, tm, tn);
}

extern "C" int evc_lstm_stack2_fwd(const evc_bf16* x, const evc_bf16* wT0, const float* bias0, const evc_bf16* wT1, const float* bias1,
                                   const int32_t* len, int T, int M, int Kin, int H, float* zx_ws,
                                   evc_bf16* hbuf0, evc_bf16* hbuf1, float* c_state0, float* h_state0, float* c_state1,
                                   float* h_state1, int64_t ld_state, void* gates0, evc_bf16* c_all0, void* gates1,
                                   evc_bf16* c_all1, void* stream) {
  EVC_REQUIRE(T > 0 && M > 0 && H > 0 && Kin > 0 && H % 64 == 0 && Kin % 64 == 0, EVC_ERR_BAD_SHAPE,
              "evc_lstm_stack2_fwd: bad shape T=%d M=%d Kin=%d H=%d (Kin, H multiples of 64)", T, M, Kin, H);
  EVC_REQUIRE(ring_operand_ok(M, Kin > H ? Kin : H) && ring_operand_ok(4L * H, (long)Kin + H) && ring_operand_ok(4L * H, 2L * H), EVC_ERR_BAD_SHAPE,
              "evc_lstm_stack2_fwd: a time slab or a kernel spans 4 GiB or more (M=%d Kin=%d H=%d)", M, Kin, H);
  EVC_REQUIRE(fwd_tail_ok(M, H, ld_state), EVC_ERR_BAD_SHAPE, "LSTM forward: a gate-record / state slab spans 4 GiB or more (M=%d H=%d ld_state=%ld)", M, H, (long)ld_state);
  EVC_REQUIRE(zx_ws && hbuf0 && hbuf1, EVC_ERR_BAD_ARG, "evc_lstm_stack2_fwd: zx_ws / hbuf0 / hbuf1 must not be NULL");
  EVC_REQUIRE(ld_state % 4 == 0 && ((uintptr_t)c_state0 % 16) == 0 && ((uintptr_t)h_state0 % 16) == 0 && ((uintptr_t)c_state1 % 16) == 0 &&
              ((uintptr_t)h_state1 % 16) == 0 && ((uintptr_t)bias0 % 16) == 0 && ((uintptr_t)bias1 % 16) == 0 &&
              ((uintptr_t)hbuf0 % 8) == 0 && ((uintptr_t)hbuf1 % 8) == 0, EVC_ERR_BAD_ALIGN,
              "evc_lstm_stack2_fwd: state/bias/hbuf must allow 16-byte vector access");
  EVC_REQUIRE((gates0 == nullptr) == (c_all0 == nullptr) && (gates1 == nullptr) == (c_all1 == nullptr) &&
              (gates0 == nullptr) == (gates1 == nullptr), EVC_ERR_BAD_ARG, "evc_lstm_stack2_fwd: gates and c_all go together, for both layers");
  EVC_REQUIRE(!gates0 || (((uintptr_t)gates0 % 16) == 0 && ((uintptr_t)gates1 % 16) == 0 && ((uintptr_t)c_all0 % 8) == 0 &&
                          ((uintptr_t)c_all1 % 8) == 0), EVC_ERR_BAD_ALIGN, "evc_lstm_stack2_fwd: gates must be 16-byte, c_all 8-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  EVC_CHECK_HIP(hipMemsetAsync(hbuf0, 0, (size_t)M * H * sizeof(bf16_t), st));       // h_{-1} = 0, both layers
  EVC_CHECK_HIP(hipMemsetAsync(hbuf1, 0, (size_t)M * H * sizeof(bf16_t), st));
  // layer 0: x-projection of all T steps as one GEMM (M ~ batch: a per-step product would be a sliver)
  int rc = evc_gemm_nt(x, Kin, wT0, (int64_t)Kin + H, zx_ws, 4L * H, T * M, 4 * H, Kin, nullptr, 0, 0, stream);
  if (rc) return rc;
  const FwdLayer L0{x, Kin, wT0, bias0, zx_ws, hbuf0, c_state0, h_state0, gates0, c_all0};
  // layer 1 reads layer 0's output slab t+1 as its x_t; fused [x_t | h_{t-1}] contraction (nothing to hoist: x_t
  // exists only one launch earlier)
  const FwdLayer L1{hbuf0 + (long)M * H, H, wT1, bias1, nullptr, hbuf1, c_state1, h_state1, gates1, c_all1};
  const int tile = pick_fwd_tile(M, H);     // 6: v1 128 rows x 32 units, 7 (M ~ 256): v1 64 x 16; others: one step per launch
  for (int s = 0; s <= T; ++s) {            // launch s: layer 0 step s next to layer 1 step s-1
    GemmOperands pa, pb;
    LstmFwdParams ea, eb;
    int k1a = 0, k2a = 0, k1b = 0, k2b = 0;
    const bool has_a = s < T, has_b = s >= 1;
    if (has_a) fwd_step_args(L0, len, s, M, H, ld_state, pa, ea, k1a, k2a);
    if (has_b) fwd_step_args(L1, len, s - 1, M, H, ld_state, pb, eb, k1b, k2b);
    if (has_a && has_b && (tile == 6 || tile == 7)) {
      if (tile == 6) launch_lstm_fwd_pair<CfgLstmBig>(pa, ea, k1a, k2a, pb, eb, k1b, k2b, st);
      else if (getenv("EVC_PAIR_V1")) launch_lstm_fwd_pair<CfgLstmSmall>(pa, ea, k1a, k2a, pb, eb, k1b, k2b, st);
      else if (getenv("EVC_PAIR_V2")) launch_lstm_fwd_pair<CfgLstmV2Small>(pa, ea, k1a, k2a, pb, eb, k1b, k2b, st);
      else launch_lstm_fwd_pair<CfgLstmV3Small>(pa, ea, k1a, k2a, pb, eb, k1b, k2b, st);
      continue;
    }
    for (int r = 0; r < 2; ++r) {
      if (!(r == 0 ? has_a : has_b)) continue;
      const GemmOperands& p = r == 0 ? pa : pb;
      const LstmFwdParams& e = r == 0 ? ea : eb;
      const int k1 = r == 0 ? k1a : k1b, k2 = r == 0 ? k2a : k2b;
      switch (tile) {
        case 0: launch_lstm_fwd<CfgLstmV2a>(p, e, k1, k2, st); break;
        case 1: launch_lstm_fwd<CfgLstmV2_288>(p, e, k1, k2, st); break;
        case 2: launch_lstm_fwd<CfgLstmV2b>(p, e, k1, k2, st); break;
        case 3: launch_lstm_fwd<CfgLstmV2_224>(p, e, k1, k2, st); break;
        case 4: launch_lstm_fwd<CfgLstmV2_192>(p, e, k1, k2, st); break;
        case 5: launch_lstm_fwd<CfgLstmV2_160>(p, e, k1, k2, st); break;
        case 6: launch_lstm_fwd<CfgLstmBig>(p, e, k1, k2, st); break;
        case 8: launch_lstm_fwd<CfgLstmV2_128>(p, e, k1, k2, st); break;
        case 9: launch_lstm_fwd<CfgLstmV2_64>(p, e, k1, k2, st); break;
        default: launch_lstm_fwd<CfgLstmSmall>(p, e, k1, k2, st); break;
      }
    }
  }
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// evc_lstm_stack2_fwd on IEEE f16 operands, with the UPPER layer's weights K-extended by their low-order halves - the "high"
// precision form of the L2 level (M = videos).  The error budget (scripts/precision_budget.py) says what this level needs: f16
// (2^-12) is enough for every activation and for layer 0's weights; the one term it does not cover is the ROUNDING OF THE UPPER
// LAYER'S WEIGHTS, the same error at every one of the 20 steps into a cell state that integrates it (6e-4 on the states).  So
// layer 1 contracts [h0_t | h0_t/64 | h1_{t-1} | h1_{t-1}/64] . [Wx | (Wx - f16(Wx))*64 | Wh | (Wh - f16(Wh))*64]^T (K = 4H instead
// of 2H; split-bf16 would be 6H in three passes), layer 0 runs plain f16 with its x-projection hoisted into one f16 product.
// h rows are WIDE, [f16(h) | f16(h)/64] (2H), written by the step epilogue together with the bf16 copy the backward pass reads.
// Same wavefront as evc_lstm_stack2_fwd: launch s = layer 0 step s next to layer 1 step s-1.
extern "C" int evc_lstm_stack2_fwd_f16(const evc_f16* x, int x_segments, const evc_f16* wT0, int h0_ext, const float* bias0,
                                       const evc_f16* wT1_wlo, const float* bias1,
                                       const int32_t* len, int T, int M, int Kin, int H, float* zx_ws,
                                       evc_f16* h0_wide, evc_f16* h1_wide, evc_bf16* hbuf0, evc_bf16* hbuf1,
                                       float* c_state0, float* h_state0, float* c_state1, float* h_state1, int64_t ld_state,
                                       void* gates0, evc_bf16* c_all0, void* gates1, evc_bf16* c_all1, void* stream) {
  EVC_REQUIRE(T > 0 && M > 0 && H > 0 && Kin > 0 && H % 64 == 0 && Kin % 64 == 0, EVC_ERR_BAD_SHAPE,
              "evc_lstm_stack2_fwd_f16: bad shape T=%d M=%d Kin=%d H=%d (Kin, H multiples of 64)", T, M, Kin, H);
  EVC_REQUIRE(x && wT0 && wT1_wlo && zx_ws && h0_wide && h1_wide && hbuf0 && hbuf1, EVC_ERR_BAD_ARG, "evc_lstm_stack2_fwd_f16: NULL operand");
  EVC_REQUIRE(x_segments >= 1 && x_segments <= 3 && (h0_ext == 0 || h0_ext == 1), EVC_ERR_BAD_ARG, "evc_lstm_stack2_fwd_f16: x_segments=%d h0_ext=%d",
              x_segments, h0_ext);
  const long Kx = (long)x_segments * Kin;              // K of the hoisted x-projection (K-extended input: evc_cast_f32_to_f16_segs)
  const long ldw0 = Kx + (h0_ext ? 2L : 1L) * H;       // row of layer 0's kernel image (evc_cast_f32_to_f16_wide)
  EVC_REQUIRE(ring_operand_ok(M, 2L * H) && ring_operand_ok(4L * H, 4L * H) && ring_operand_ok(4L * H, 3L * Kin + 2L * H), EVC_ERR_BAD_SHAPE,
              "evc_lstm_stack2_fwd_f16: an operand spans 4 GiB or more");
  EVC_REQUIRE(fwd_tail_ok(M, H, ld_state), EVC_ERR_BAD_SHAPE, "LSTM forward: a gate-record / state slab spans 4 GiB or more (M=%d H=%d ld_state=%ld)", M, H, (long)ld_state);
  EVC_REQUIRE(ld_state % 4 == 0 && ((uintptr_t)c_state0 % 16) == 0 && ((uintptr_t)h_state0 % 16) == 0 && ((uintptr_t)c_state1 % 16) == 0 &&
              ((uintptr_t)h_state1 % 16) == 0 && ((uintptr_t)bias0 % 16) == 0 && ((uintptr_t)bias1 % 16) == 0 && ((uintptr_t)h0_wide % 16) == 0 &&
              ((uintptr_t)h1_wide % 16) == 0 && ((uintptr_t)hbuf0 % 8) == 0 && ((uintptr_t)hbuf1 % 8) == 0, EVC_ERR_BAD_ALIGN,
              "evc_lstm_stack2_fwd_f16: state/bias/h buffers must allow 16-byte vector access");
  EVC_REQUIRE((gates0 == nullptr) == (c_all0 == nullptr) && (gates1 == nullptr) == (c_all1 == nullptr) && (gates0 == nullptr) == (gates1 == nullptr),
              EVC_ERR_BAD_ARG, "evc_lstm_stack2_fwd_f16: gates and c_all go together, for both layers");
  hipStream_t st = (hipStream_t)stream;
  EVC_CHECK_HIP(hipMemsetAsync(h0_wide, 0, (size_t)M * 2 * H * sizeof(f16_t), st));        // h_{-1} = 0, both layers, both images
  EVC_CHECK_HIP(hipMemsetAsync(h1_wide, 0, (size_t)M * 2 * H * sizeof(f16_t), st));
  EVC_CHECK_HIP(hipMemsetAsync(hbuf0, 0, (size_t)M * H * sizeof(bf16_t), st));
  EVC_CHECK_HIP(hipMemsetAsync(hbuf1, 0, (size_t)M * H * sizeof(bf16_t), st));
  int rc = gemm_nt_f16(x, Kx, wT0, ldw0, zx_ws, 4L * H, T * M, 4 * H, (int)Kx, stream);
  if (rc) return rc;
  for (int s = 0; s <= T; ++s) {            // launch s: layer 0 step s next to layer 1 step s-1
    GemmOperands pa, pb;
    LstmFwdParams ea, eb;
    int k1a = 0, k2a = 0, k1b = 0, k2b = 0;
    const bool has_a = s < T, has_b = s >= 1;
    if (has_a) {                             // layer 0, step s: zx + h0_{s-1} . Wh0^T (K = H of the wide rows; h0_ext: all 2H against [Wh | Wh_lo*64])
      const int t = s;
      pa.M = M; pa.Nu = H; pa.group_stride = H; pa.ldb = ldw0; pa.nk1 = pa.nk2 = 0;
      pa.A1lo = pa.A2lo = pa.Blo = nullptr;
      const bf16_t* hprev = (const bf16_t*)h0_wide + (long)t * M * 2 * H;
      pa.A1 = hprev; pa.lda1 = 2L * H; k1a = (t == 0) ? 0 : (h0_ext ? 2 * H : H); pa.A2 = hprev; pa.lda2 = 2L * H; k2a = 0;
      pa.B = (const bf16_t*)wT0 + Kx;
      ea.zx = zx_ws + (long)t * M * 4 * H; ea.ldzx = 4L * H;
      ea.bias = bias0; ea.len = len; ea.t = t;
      ea.c_state = c_state0; ea.h_state = h_state0; ea.ld_state = ld_state;
      ea.hout = (bf16_t*)h0_wide + (long)(t + 1) * M * 2 * H; ea.h_wide = 1;
      ea.hout_lo = hbuf0 + (long)(t + 1) * M * H;
      ea.gates = gates0 ? (uint2*)gates0 + (long)t * M * H : nullptr;
      ea.c_hist = c_all0 ? c_all0 + (long)(t + 1) * M * H : nullptr;
      ea.row_map = nullptr; ea.M = M; ea.H = H;
    }
    if (has_b) {                             // layer 1, step s-1: [h0_t | h0_t/64 | h1_{t-1} | h1_{t-1}/64] . [Wx | Wx_lo*64 | Wh | Wh_lo*64]^T
      const int t = s - 1;
      pb.M = M; pb.Nu = H; pb.group_stride = H; pb.ldb = 4L * H; pb.nk1 = pb.nk2 = 0;
      pb.A1lo = pb.A2lo = pb.Blo = nullptr;
      pb.A1 = (const bf16_t*)h0_wide + (long)(t + 1) * M * 2 * H; pb.lda1 = 2L * H; k1b = 2 * H;
      pb.A2 = (const bf16_t*)h1_wide + (long)t * M * 2 * H; pb.lda2 = 2L * H; k2b = (t == 0) ? 0 : 2 * H;
      pb.B = (const bf16_t*)wT1_wlo;
      eb.zx = nullptr; eb.ldzx = 0;
      eb.bias = bias1; eb.len = len; eb.t = t;
      eb.c_state = c_state1; eb.h_state = h_state1; eb.ld_state = ld_state;
      eb.hout = (bf16_t*)h1_wide + (long)(t + 1) * M * 2 * H; eb.h_wide = 1;
      eb.hout_lo = hbuf1 + (long)(t + 1) * M * H;
      eb.gates = gates1 ? (uint2*)gates1 + (long)t * M * H : nullptr;
      eb.c_hist = c_all1 ? c_all1 + (long)(t + 1) * M * H : nullptr;
      eb.row_map = nullptr; eb.M = M; eb.H = H;
    }
    if (has_a && has_b) launch_lstm_fwd_pair<CfgLstmV3Small, true>(pa, ea, k1a, k2a, pb, eb, k1b, k2b, st);
    else if (has_a) launch_lstm_fwd<CfgLstmV3Small, false, true>(pa, ea, k1a, k2a, st);
    else launch_lstm_fwd<CfgLstmV3Small, false, true>(pb, eb, k1b, k2b, st);
  }
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// evc_lstm_stack2_fwd_f16 with the low-order halves of the recurrent weights of layer 0 and of all weights of layer 1 contracted as e4m3
// operands behind the f16 stages of the same (pair) launches (LOOP_FP8_TAIL) instead of f16 K-extensions: layer 1 walks 32 f16 + 16 e4m3
// stages instead of 64 f16 ones (H = 1024) - these steps are bound by their chain of dependent stages.  x [T][M][x_segments Kin] f16
// (K-extended input of the hoisted product, as before); wT0 [4H][x_segments Kin + H] f16 = [Wx segments | f16(Wh)], wT0_8 [4H][H] bytes =
// e4m3((Wh - f16(Wh)) 2^w8_scale_exp); wT1 [4H][2H] f16, wT1_8 [4H][2H] bytes; h0_rows / h1_rows [(T+1)][M] rows of 3H bytes = [f16(h) |
// e4m3(h 2^7)].  H % 128 == 0, H >= 512.
extern "C" int evc_lstm_stack2_fwd_f16_fp8lo(const evc_f16* x, int x_segments, const evc_f16* wT0, const uint8_t* wT0_8, const float* bias0,
                                             const evc_f16* wT1, const uint8_t* wT1_8, int w8_scale_exp, const float* bias1,
                                             const int32_t* len, int T, int M, int Kin, int H, float* zx_ws,
                                             evc_f16* h0_rows, evc_f16* h1_rows, evc_bf16* hbuf0, evc_bf16* hbuf1,
                                             float* c_state0, float* h_state0, float* c_state1, float* h_state1, int64_t ld_state,
                                             void* gates0, evc_bf16* c_all0, void* gates1, evc_bf16* c_all1, void* stream) {
  EVC_REQUIRE(T > 0 && M > 0 && H >= 512 && Kin > 0 && H % 128 == 0 && Kin % 64 == 0, EVC_ERR_BAD_SHAPE,
              "evc_lstm_stack2_fwd_f16_fp8lo: bad shape T=%d M=%d Kin=%d (%%64) H=%d (%%128, >= 512)", T, M, Kin, H);
  EVC_REQUIRE(x && wT0 && wT0_8 && wT1 && wT1_8 && zx_ws && h0_rows && h1_rows && hbuf0 && hbuf1, EVC_ERR_BAD_ARG, "evc_lstm_stack2_fwd_f16_fp8lo: NULL operand");
  EVC_REQUIRE(x_segments >= 1 && x_segments <= 3 && w8_scale_exp >= 0 && w8_scale_exp <= 60, EVC_ERR_BAD_ARG,
              "evc_lstm_stack2_fwd_f16_fp8lo: x_segments=%d w8_scale_exp=%d", x_segments, w8_scale_exp);
  const long Kx = (long)x_segments * Kin;
  const long ldw0 = Kx + H, ldh = 3L * H / 2;
  EVC_REQUIRE(ring_operand_ok(M, ldh) && ring_operand_ok(4L * H, ldw0) && ring_operand_ok(4L * H, 2L * H), EVC_ERR_BAD_SHAPE,
              "evc_lstm_stack2_fwd_f16_fp8lo: an operand spans 4 GiB or more");
  EVC_REQUIRE(fwd_tail_ok(M, H, ld_state), EVC_ERR_BAD_SHAPE, "LSTM forward: a gate-record / state slab spans 4 GiB or more (M=%d H=%d ld_state=%ld)", M, H, (long)ld_state);
  EVC_REQUIRE(ld_state % 4 == 0 && ((uintptr_t)c_state0 % 16) == 0 && ((uintptr_t)h_state0 % 16) == 0 && ((uintptr_t)c_state1 % 16) == 0 &&
              ((uintptr_t)h_state1 % 16) == 0 && ((uintptr_t)bias0 % 16) == 0 && ((uintptr_t)bias1 % 16) == 0 && ((uintptr_t)h0_rows % 16) == 0 &&
              ((uintptr_t)h1_rows % 16) == 0 && ((uintptr_t)hbuf0 % 8) == 0 && ((uintptr_t)hbuf1 % 8) == 0 && ((uintptr_t)wT0_8 % 16) == 0 &&
              ((uintptr_t)wT1_8 % 16) == 0 && ((uintptr_t)wT0 % 16) == 0 && ((uintptr_t)wT1 % 16) == 0, EVC_ERR_BAD_ALIGN,
              "evc_lstm_stack2_fwd_f16_fp8lo: state/bias/h buffers and weight images must allow 16-byte vector access");
  EVC_REQUIRE((gates0 == nullptr) == (c_all0 == nullptr) && (gates1 == nullptr) == (c_all1 == nullptr) && (gates0 == nullptr) == (gates1 == nullptr),
              EVC_ERR_BAD_ARG, "evc_lstm_stack2_fwd_f16_fp8lo: gates and c_all go together, for both layers");
  hipStream_t st = (hipStream_t)stream;
  bf16_t* h0r = (bf16_t*)h0_rows;
  bf16_t* h1r = (bf16_t*)h1_rows;
  EVC_CHECK_HIP(hipMemsetAsync(h0r, 0, (size_t)M * ldh * sizeof(bf16_t), st));        // h_{-1} = 0, both layers, every image
  EVC_CHECK_HIP(hipMemsetAsync(h1r, 0, (size_t)M * ldh * sizeof(bf16_t), st));
  EVC_CHECK_HIP(hipMemsetAsync(hbuf0, 0, (size_t)M * H * sizeof(bf16_t), st));
  EVC_CHECK_HIP(hipMemsetAsync(hbuf1, 0, (size_t)M * H * sizeof(bf16_t), st));
  int rc = gemm_nt_f16(x, Kx, wT0, ldw0, zx_ws, 4L * H, T * M, 4 * H, (int)Kx, stream);
  if (rc) return rc;
  for (int s = 0; s <= T; ++s) {            // launch s: layer 0 step s next to layer 1 step s-1
    GemmOperands pa, pb;
    LstmFwdParams ea, eb;
    int k1a = 0, k1b = 0, k2b = 0;
    const bool has_a = s < T, has_b = s >= 1;
    if (has_a) {                             // layer 0, step s: zx + h0_{s-1} . Wh0^T (f16) + 2^-(7+e) e4m3(h0_{s-1}) . e4m3(lo(Wh0))^T
      const int t = s;
      pa.M = M; pa.Nu = H; pa.group_stride = H; pa.ldb = ldw0; pa.nk1 = pa.nk2 = 0;
      pa.A1lo = pa.A2lo = pa.Blo = nullptr;
      const bf16_t* hprev = h0r + (long)t * M * ldh;
      pa.A1 = hprev; pa.lda1 = ldh; k1a = (t == 0) ? 0 : H; pa.A2 = hprev; pa.lda2 = ldh;
      pa.B = (const bf16_t*)wT0 + Kx;
      pa.A3 = (const uint8_t*)(hprev + H); pa.lda3 = ldh * 2; pa.nk3 = (t == 0) ? 0 : H / 128;
      pa.A4 = pa.A3; pa.lda4 = pa.lda3; pa.nk4 = 0;
      pa.B8 = wT0_8; pa.ldb8 = H; pa.scale8_exp = -(7 + w8_scale_exp);
      ea.zx = zx_ws + (long)t * M * 4 * H; ea.ldzx = 4L * H;
      ea.bias = bias0; ea.len = len; ea.t = t;
      ea.c_state = c_state0; ea.h_state = h_state0; ea.ld_state = ld_state;
      ea.hout = h0r + (long)(t + 1) * M * ldh; ea.h_wide = 2;
      ea.hout_lo = hbuf0 + (long)(t + 1) * M * H;
      ea.gates = gates0 ? (uint2*)gates0 + (long)t * M * H : nullptr;
      ea.c_hist = c_all0 ? c_all0 + (long)(t + 1) * M * H : nullptr;
      ea.row_map = nullptr; ea.M = M; ea.H = H;
    }
    if (has_b) {                             // layer 1, step s-1: [h0_t | h1_{t-1}] . [Wx | Wh]^T (f16) + the same rows' e4m3 parts against e4m3(lo([Wx | Wh]))
      const int t = s - 1;
      pb.M = M; pb.Nu = H; pb.group_stride = H; pb.ldb = 2L * H; pb.nk1 = pb.nk2 = 0;
      pb.A1lo = pb.A2lo = pb.Blo = nullptr;
      const bf16_t* xin = h0r + (long)(t + 1) * M * ldh;
      const bf16_t* hprev = h1r + (long)t * M * ldh;
      pb.A1 = xin; pb.lda1 = ldh; k1b = H;
      pb.A2 = hprev; pb.lda2 = ldh; k2b = (t == 0) ? 0 : H;
      pb.B = (const bf16_t*)wT1;
      pb.A3 = (const uint8_t*)(xin + H); pb.lda3 = ldh * 2; pb.nk3 = H / 128;
      pb.A4 = (const uint8_t*)(hprev + H); pb.lda4 = ldh * 2; pb.nk4 = (t == 0) ? 0 : H / 128;
      pb.B8 = wT1_8; pb.ldb8 = 2L * H; pb.scale8_exp = -(7 + w8_scale_exp);
      eb.zx = nullptr; eb.ldzx = 0;
      eb.bias = bias1; eb.len = len; eb.t = t;
      eb.c_state = c_state1; eb.h_state = h_state1; eb.ld_state = ld_state;
      eb.hout = h1r + (long)(t + 1) * M * ldh; eb.h_wide = 2;
      eb.hout_lo = hbuf1 + (long)(t + 1) * M * H;
      eb.gates = gates1 ? (uint2*)gates1 + (long)t * M * H : nullptr;
      eb.c_hist = c_all1 ? c_all1 + (long)(t + 1) * M * H : nullptr;
      eb.row_map = nullptr; eb.M = M; eb.H = H;
    }
    if (has_a && has_b) launch_lstm_fwd_pair<CfgLstmV3Small, true, true>(pa, ea, k1a, 0, pb, eb, k1b, k2b, st);
    else if (has_a) launch_lstm_fwd<CfgLstmV3Small, false, true, true>(pa, ea, k1a, 0, st);
    else launch_lstm_fwd<CfgLstmV3Small, false, true, true>(pb, eb, k1b, k2b, st);
  }
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// ===========================================================================
// LSTM backward step t: dh = dz_{t+1} . Wh^T (+ final-state / upper-layer grads),
// then the gate derivative -> dz_t, dc carried in dc_ws.
// ===========================================================================
struct LstmBwdParams {
  const int* len; int t;
  const uint2* gates;       // slab t   [M][H] bf16 {i, j, f, o}
  const bf16_t* c_new; const bf16_t* c_old;   // slabs t+1 / t of the bf16 cell-state history (c_old == NULL at t == 0)
  const float* dS_c; const float* dS_h; long ld_dS;
  const bf16_t* dh_above;   // slab t [M][H] bf16 (dX of the layer above) or NULL
  float* dc_ws;             // [M][H] f32 (dc_bf16: the same buffer holding [M][H] bf16): the carried cell-state gradient
  int dc_bf16;              // 1: dc crosses the launch boundary as bf16 (EVC_BWD_DC_BF16=1: -15 of the step's 113 MB; A/B switch)
  uint2* dz4;               // slab t [M][H] gate-interleaved: 4 bf16 (dz_i, dz_j, dz_f, dz_o) per (row, unit)
  const int* row_map;       // slot -> row of dS_c / dS_h (row plan) or NULL
  float* db;                // [4H] bias gradient (TF gate order), accumulated with atomics over rows and steps, or NULL
  int m_active;             // rows [m_active, M) are inactive at this step: tiles entirely beyond it only zero dz
  int M, H;
  int fused_above;          // 1: the accumulator also holds the gradient from the layer above (second K segment, wavefront):
                            // at a row's last step the final-state gradient is ADDED to it instead of replacing it
};

static inline int bwd_dc_bf16() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("EVC_BWD_DC_BF16"); v = (e && e[0] == '1') ? 1 : 0; }
  return v;
}

// Gate derivative of one (row, 4 consecutive units): dh[4] = what flowed back through the recurrent
// product; writes dz (4 x 8 bytes), carries dc in dc_ws.
// What the gate derivative of one (row, 4 consecutive units) reads: loaded in one phase for all of a lane's
// fragments (the stores of a fragment and the loads of the next hit the same arrays - dc_ws is updated in place -
// so in program order every fragment would wait for the previous one's stores to be acknowledged).
struct LstmBwdIn {
  int ln;                   // sequence length of the row (-1: row outside the launch)
  float4 dcv;               // dc arriving at this step (or the final-state gradient at t = len-1)
  float4 dhs;               // final-state dh at t = len-1
  uint2 dha;                // 4 bf16: dX of the layer above
  uint4 g01, g23;           // gate records of the 4 units
  uint2 cnq, coq;           // bf16 c after / before this step
};

__device__ __forceinline__ void lstm_bwd_load(const LstmBwdParams& e, const int m, const int u, const bool in_range, LstmBwdIn& q) {
  q.ln = in_range ? e.len[m] : -1;
  q.dcv = q.dhs = make_float4(0.f, 0.f, 0.f, 0.f);
  q.dha = q.cnq = q.coq = make_uint2(0u, 0u);
  q.g01 = q.g23 = make_uint4(0u, 0u, 0u, 0u);
  if (e.t >= q.ln) return;                        // inactive (or outside): nothing is read
  const long hu = (long)m * e.H + u;
  if (e.t == q.ln - 1) {
    const long su = (long)(e.row_map ? e.row_map[m] : m) * e.ld_dS + u;
    q.dhs = *(const float4*)(e.dS_h + su);        // nothing flows back from later (inactive) steps
    q.dcv = *(const float4*)(e.dS_c + su);
  } else {
    if (e.dc_bf16) {
      const uint2 d = *(const uint2*)((const bf16_t*)e.dc_ws + hu);
      q.dcv = make_float4(__uint_as_float(d.x << 16), __uint_as_float(d.x & 0xffff0000u), __uint_as_float(d.y << 16), __uint_as_float(d.y & 0xffff0000u));
    } else {
      q.dcv = *(const float4*)(e.dc_ws + hu);
    }
  }
  if (e.dh_above) q.dha = *(const uint2*)(e.dh_above + hu);
  const uint4* gp = (const uint4*)(e.gates + hu);
  q.g01 = gp[0]; q.g23 = gp[1];
  q.cnq = *(const uint2*)(e.c_new + hu);
  if (e.c_old) q.coq = *(const uint2*)(e.c_old + hu);
}

// dh_in[4] = what flowed back through the recurrent product; writes dz (4 x 8 bytes), carries dc in dc_ws.
// dzv[unit][gate] receives the (unrounded) f32 gate gradients - zeros for an inactive row - for the bias gradient.
__device__ __forceinline__ void lstm_bwd_finish(const LstmBwdParams& e, const int m, const int u, const float (&dh_in)[4],
                                                const LstmBwdIn& q, float (&dzv)[4][4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int g = 0; g < 4; ++g) dzv[r][g] = 0.f;
  if (q.ln < 0) return;
  const long hu = (long)m * e.H + u;
  uint4* dzp = (uint4*)(e.dz4 + hu);            // 4 units x 8 bytes = 2 x 16 bytes
  if (e.t >= q.ln) {  // inactive: state passes through, no gate gradient
    dzp[0] = make_uint4(0u, 0u, 0u, 0u);
    dzp[1] = make_uint4(0u, 0u, 0u, 0u);
    return;
  }
  float dh[4] = {dh_in[0], dh_in[1], dh_in[2], dh_in[3]};
  if (e.t == q.ln - 1) {     // nothing flows back through the recurrent product from the (inactive) later steps: dz_{t+1} of this row is 0
    if (e.fused_above) { dh[0] += q.dhs.x; dh[1] += q.dhs.y; dh[2] += q.dhs.z; dh[3] += q.dhs.w; }
    else { dh[0] = q.dhs.x; dh[1] = q.dhs.y; dh[2] = q.dhs.z; dh[3] = q.dhs.w; }
  }
  if (e.dh_above) {
    dh[0] += __uint_as_float(q.dha.x << 16); dh[1] += __uint_as_float(q.dha.x & 0xffff0000u);
    dh[2] += __uint_as_float(q.dha.y << 16); dh[3] += __uint_as_float(q.dha.y & 0xffff0000u);
  }
  const float dci[4] = {q.dcv.x, q.dcv.y, q.dcv.z, q.dcv.w};
  const uint2 recs[4] = {make_uint2(q.g01.x, q.g01.y), make_uint2(q.g01.z, q.g01.w), make_uint2(q.g23.x, q.g23.y), make_uint2(q.g23.z, q.g23.w)};
  const float cna[4] = {__uint_as_float(q.cnq.x << 16), __uint_as_float(q.cnq.x & 0xffff0000u),
                        __uint_as_float(q.cnq.y << 16), __uint_as_float(q.cnq.y & 0xffff0000u)};
  const float coa[4] = {__uint_as_float(q.coq.x << 16), __uint_as_float(q.coq.x & 0xffff0000u),
                        __uint_as_float(q.coq.y << 16), __uint_as_float(q.coq.y & 0xffff0000u)};
  float dcn[4];
  uint2 dzr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint2 rec = recs[r];
    const float gi = __uint_as_float(rec.x << 16), gj = __uint_as_float(rec.x & 0xffff0000u);
    const float gf = __uint_as_float(rec.y << 16), go = __uint_as_float(rec.y & 0xffff0000u);
    const float tcv = tanhf_(cna[r]);
    const float cp = coa[r];
    const float dc = dci[r] + dh[r] * go * (1.f - tcv * tcv);
    dcn[r] = dc * gf;
    dzv[r][0] = dc * gj * gi * (1.f - gi); dzv[r][1] = dc * gi * (1.f - gj * gj);
    dzv[r][2] = dc * cp * gf * (1.f - gf); dzv[r][3] = dh[r] * tcv * go * (1.f - go);
    dzr[r] = make_uint2(pack_bf16x2(dzv[r][0], dzv[r][1]), pack_bf16x2(dzv[r][2], dzv[r][3]));
  }
  if (e.dc_bf16) *(uint2*)((bf16_t*)e.dc_ws + hu) = make_uint2(pack_bf16x2(dcn[0], dcn[1]), pack_bf16x2(dcn[2], dcn[3]));
  else *(float4*)(e.dc_ws + hu) = make_float4(dcn[0], dcn[1], dcn[2], dcn[3]);
  dzp[0] = make_uint4(dzr[0].x, dzr[0].y, dzr[1].x, dzr[1].y);
  dzp[1] = make_uint4(dzr[2].x, dzr[2].y, dzr[3].x, dzr[3].y);
}

// tiles entirely beyond the active prefix (row plan): dz = 0, no GEMM
template <int BM, int BU, int NT>
__device__ __forceinline__ void lstm_bwd_zero_tile(const LstmBwdParams& e, int m0, int u0) {
  const int cols = min(BU, e.H - u0) / 2;                  // 16-byte pieces (2 units) per row
  for (int i = threadIdx.x; i < BM * cols; i += NT) {
    const int m = m0 + i / cols, u = u0 + (i % cols) * 2;
    if (m < e.M) *(uint4*)(e.dz4 + (long)m * e.H + u) = make_uint4(0u, 0u, 0u, 0u);
  }
}

// Row-major gate-derivative tail for the ring tiles (BM x 128 units, 512 threads): the accumulators (dh) go through LDS
// and the tail then walks the tile row by row - one wave = one row of 128 units, lane = 2 consecutive units - so every
// global access of the tail is a contiguous run over the whole wave (gate records 1 KB, dz 1 KB, dc 512 B, cell history
// 256 B per row), GROUP rows in flight per thread.  Straight from the accumulator layout (lane = 4 units of one row, 16
// rows per instruction) the same bytes moved in 32-64-byte pieces and the tail took 32 of the step's 70 us.
template <class Cfg>
__device__ __forceinline__ void lstm_bwd_tail_rowmajor(f32x4 (&acc)[Cfg::MI][1][Cfg::NI], const LstmBwdParams& e, int m0, int u0, char* lds) {
  static_assert(Cfg::BU == 128 && Cfg::NT == 512, "row-major tail: 128-unit tiles, 8 waves");
  constexpr int RS = Cfg::BU * 4 + 16;                        // dh rows in LDS, padded
  static_assert(Cfg::BM * RS <= Cfg::LDS_BYTES && 8 * 128 * 4 * 4 <= Cfg::LDS_BYTES, "dh tile (then the bias-gradient partials) must fit the ring");
  // (wave as a SCALAR: a row's length and its row_map entry are then scalar loads - counted by lgkmcnt.  As vector loads they
  // were followed by `s_waitcnt vmcnt(0)` for the branch on the length, which also waited for every data load of the rows
  // before: the GROUP rows "in flight" were loaded one after the other.)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  {
    const int wr = wave / Cfg::WC, wc = wave % Cfg::WC;
    const int l = lane & 15, g = lane >> 4;
    __syncthreads();                                           // every wave has read its last ring slot
#pragma unroll
    for (int mi = 0; mi < Cfg::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < Cfg::NI; ++ni)
        *(f32x4*)(lds + (wr * Cfg::WM + mi * 16 + l) * RS + (wc * Cfg::WU + ni * 16 + g * 4) * 4) = acc[mi][0][ni];
    __syncthreads();
  }
  const int u = u0 + lane * 2;                                // this lane's two units
  const bool u_in = u < e.H;                                  // H % 2 == 0
  float bs[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  constexpr int GROUP = 4;                                    // rows per thread in flight (2: 67.0 us per step, 4: 65.1)
  static_assert(Cfg::BM % (8 * GROUP) == 0, "tile rows must divide into 8 waves x GROUP");
  for (int p0 = 0; p0 < Cfg::BM / 8; p0 += GROUP) {
    int ln[GROUP];
    float2 dcv[GROUP], dhs[GROUP], dhv[GROUP];
    uint4 grec[GROUP];
    uint32_t cn[GROUP], co[GROUP], dha[GROUP];
#pragma unroll
    for (int i = 0; i < GROUP; ++i) {                          // load phase
      const int rl = (p0 + i) * 8 + wave;
      const int m = m0 + rl;                                   // wave-uniform
      const int lnm = m < e.M ? ((const __attribute__((address_space(4))) int*)e.len)[m] : -1;   // scalar load (constant address space)
      ln[i] = u_in ? lnm : -1;
      dcv[i] = dhs[i] = make_float2(0.f, 0.f);
      grec[i] = make_uint4(0u, 0u, 0u, 0u);
      cn[i] = co[i] = dha[i] = 0u;
      dhv[i] = *(const float2*)(lds + rl * RS + lane * 8);
      if (e.t < ln[i]) {
        const long hu = (long)m * e.H + u;
        if (e.t == ln[i] - 1) {
          const long su = (long)(e.row_map ? ((const __attribute__((address_space(4))) int*)e.row_map)[m] : m) * e.ld_dS + u;
          dhs[i] = *(const float2*)(e.dS_h + su);
          dcv[i] = *(const float2*)(e.dS_c + su);
        } else {
          if (e.dc_bf16) {
            const uint32_t d = *(const uint32_t*)((const bf16_t*)e.dc_ws + hu);
            dcv[i] = make_float2(__uint_as_float(d << 16), __uint_as_float(d & 0xffff0000u));
          } else {
            dcv[i] = *(const float2*)(e.dc_ws + hu);
          }
        }
        if (e.dh_above) dha[i] = *(const uint32_t*)(e.dh_above + hu);
        grec[i] = *(const uint4*)(e.gates + hu);
        cn[i] = *(const uint32_t*)(e.c_new + hu);
        if (e.c_old) co[i] = *(const uint32_t*)(e.c_old + hu);
      }
    }
    // compute phase, then store phase: with the stores of row i between the computations of rows i and i+1 hipcc put
    // `s_waitcnt vmcnt(0)` in front of every row (it cannot count across the per-row branches), i.e. every row waited for the
    // store acknowledgements of the row before
    float2 dcn[GROUP];
    uint4 dzr[GROUP];
    int what[GROUP];                                           // 0: nothing, 1: zero dz (inactive row), 2: dc + dz
#pragma unroll
    for (int i = 0; i < GROUP; ++i) {
      what[i] = ln[i] < 0 ? 0 : (e.t >= ln[i] ? 1 : 2);
      dcn[i] = make_float2(0.f, 0.f);
      dzr[i] = make_uint4(0u, 0u, 0u, 0u);
      if (what[i] != 2) continue;
      float dh[2] = {dhv[i].x, dhv[i].y};
      if (e.t == ln[i] - 1) {     // nothing flows back through the recurrent product from the (inactive) later steps
        if (e.fused_above) { dh[0] += dhs[i].x; dh[1] += dhs[i].y; }
        else { dh[0] = dhs[i].x; dh[1] = dhs[i].y; }
      }
      if (e.dh_above) { dh[0] += __uint_as_float(dha[i] << 16); dh[1] += __uint_as_float(dha[i] & 0xffff0000u); }
      const float dci[2] = {dcv[i].x, dcv[i].y};
      const uint2 recs[2] = {make_uint2(grec[i].x, grec[i].y), make_uint2(grec[i].z, grec[i].w)};
      const float cna[2] = {__uint_as_float(cn[i] << 16), __uint_as_float(cn[i] & 0xffff0000u)};
      const float coa[2] = {__uint_as_float(co[i] << 16), __uint_as_float(co[i] & 0xffff0000u)};
      float dcv2[2];
      uint2 dz2[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const float gi = __uint_as_float(recs[r].x << 16), gj = __uint_as_float(recs[r].x & 0xffff0000u);
        const float gf = __uint_as_float(recs[r].y << 16), go = __uint_as_float(recs[r].y & 0xffff0000u);
        const float tcv = tanhf_(cna[r]);
        const float dc = dci[r] + dh[r] * go * (1.f - tcv * tcv);
        dcv2[r] = dc * gf;
        const float z0 = dc * gj * gi * (1.f - gi), z1 = dc * gi * (1.f - gj * gj);
        const float z2 = dc * coa[r] * gf * (1.f - gf), z3 = dh[r] * tcv * go * (1.f - go);
        bs[r][0] += z0; bs[r][1] += z1; bs[r][2] += z2; bs[r][3] += z3;
        dz2[r] = make_uint2(pack_bf16x2(z0, z1), pack_bf16x2(z2, z3));
      }
      dcn[i] = make_float2(dcv2[0], dcv2[1]);
      dzr[i] = make_uint4(dz2[0].x, dz2[0].y, dz2[1].x, dz2[1].y);
    }
    // every load of the group has been consumed above; saying so (vmcnt(0), encoded 0x0F70) lets the stores below issue back
    // to back - across the per-row branches hipcc otherwise keeps some load destinations "pending" and waits before each row
    __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
    for (int i = 0; i < GROUP; ++i) {
      if (what[i] == 0) continue;
      const long hu = (long)(m0 + (p0 + i) * 8 + wave) * e.H + u;
      if (what[i] == 2) {
        if (e.dc_bf16) *(uint32_t*)((bf16_t*)e.dc_ws + hu) = pack_bf16x2(dcn[i].x, dcn[i].y);
        else *(float2*)(e.dc_ws + hu) = dcn[i];
      }
      *(uint4*)(e.dz4 + hu) = dzr[i];                          // zeros for an inactive row: state passes through, no gate gradient
    }
  }
  if (e.db) {      // bias gradient: the 8 waves hold partial sums of the same 128 units x 4 gates: through LDS, then one atomic per sum
    float* red = (float*)lds;                                  // [8 waves][128 units][4 gates], over the dh tile
    __syncthreads();                                           // every wave has read its last dh row
    *(float4*)(red + ((wave * 128) + lane * 2) * 4) = make_float4(bs[0][0], bs[0][1], bs[0][2], bs[0][3]);
    *(float4*)(red + ((wave * 128) + lane * 2 + 1) * 4) = make_float4(bs[1][0], bs[1][1], bs[1][2], bs[1][3]);
    __syncthreads();
    const int uu = threadIdx.x >> 2, gg = threadIdx.x & 3;     // 512 threads = 128 units x 4 gates
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) v += red[(w * 128 + uu) * 4 + gg];
    if (u0 + uu < e.H) atomicAdd(e.db + (long)gg * e.H + u0 + uu, v);
  }
}

// BATCH_LOADS: issue the epilogue loads of all MI fragments of a unit group before the first store (one workgroup per CU:
// the only way to overlap them); false: fragment by fragment (fewer registers: the pair kernel runs two workgroups per CU
// and hides the round trips behind the other workgroup's main loop)
template <class Cfg, bool BATCH_LOADS = true>
__device__ __forceinline__ void lstm_bwd_step_body(const GemmOperands& p, const LstmBwdParams& e, int bid, int tiles_m, int tiles_n) {
  static_assert(Cfg::G == 1, "bwd step is a plain GEMM over the H units");
  const int nwg = tiles_m * tiles_n;
  const int id = xcd_remap(bid, nwg);
  int tm, tn;
  tile_of(id, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * Cfg::BM, u0 = tn * Cfg::BU;
  if (m0 >= e.m_active) {
    lstm_bwd_zero_tile<Cfg::BM, Cfg::BU, Cfg::NT>(e, m0, u0);
    return;
  }
  f32x4 acc[Cfg::MI][1][Cfg::NI];
#ifdef EVC_ABLATE_BWD_MAIN     // debug build: epilogue only
#pragma unroll
  for (int mi = 0; mi < Cfg::MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < Cfg::NI; ++ni) acc[mi][0][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#else
  run_mainloop<Cfg, 1, true, true, EVC_BWD_LOOP_MODE>(p, m0, u0, acc);     // transposed accumulators: lane = one row, 4 consecutive units
#endif
#if !defined(EVC_ABLATE_BWD_EPI) && !defined(EVC_BWD_TAIL_FRAGMENTS)
  if constexpr (is_v2<Cfg>::value && Cfg::BU == 128 && Cfg::NT == 512 && Cfg::BM % 32 == 0 && BATCH_LOADS) {
    lstm_bwd_tail_rowmajor<Cfg>(acc, e, m0, u0, lds_dyn);
    return;
  }
#endif
  TileCoordsT<Cfg> tc;
#pragma unroll
  for (int ni = 0; ni < Cfg::NI; ++ni) {
    const int u = u0 + tc.unit0 + ni * 16;
    if (u >= e.H) continue;
    LstmBwdIn in[BATCH_LOADS ? Cfg::MI : 1];            // load phase: every fragment of this unit group
    if constexpr (BATCH_LOADS) {
#pragma unroll
      for (int mi = 0; mi < Cfg::MI; ++mi) {
        const int m = m0 + tc.row0 + mi * 16;
        lstm_bwd_load(e, m, u, m < e.M, in[mi]);
      }
    }
    float bs[4][4];                                    // this lane's column sums over its rows: [unit][gate]
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int g = 0; g < 4; ++g) bs[r][g] = 0.f;
#pragma unroll
    for (int mi = 0; mi < Cfg::MI; ++mi) {
      const int m = m0 + tc.row0 + mi * 16;
      const float dh[4] = {acc[mi][0][ni][0], acc[mi][0][ni][1], acc[mi][0][ni][2], acc[mi][0][ni][3]};
#ifdef EVC_ABLATE_BWD_EPI     // debug build: main loop only (keep the accumulators alive, store nothing)
      asm volatile("" :: "v"(dh[0]), "v"(dh[1]), "v"(dh[2]), "v"(dh[3]));
#else
      float dzv[4][4];
      if constexpr (!BATCH_LOADS) lstm_bwd_load(e, m, u, m < e.M, in[0]);
      lstm_bwd_finish(e, m, u, dh, in[BATCH_LOADS ? mi : 0], dzv);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int g = 0; g < 4; ++g) bs[r][g] += dzv[r][g];
#endif
    }
    if (e.db) {                                        // bias gradient: the 16 lanes l&15 hold 16 rows of the same 4 units
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v = bs[r][g];
          v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
          if ((threadIdx.x & 15) == 0) atomicAdd(e.db + (long)g * e.H + u + r, v);
        }
    }
  }
}

template <class Cfg>
__global__ __launch_bounds__(Cfg::NT) void lstm_bwd_step_kernel(GemmOperands p, LstmBwdParams e, int tiles_m, int tiles_n) {
  lstm_bwd_step_body<Cfg>(p, e, blockIdx.x, tiles_m, tiles_n);
}

// BPTT wavefront of a two-layer stack: layer 0's step t+1 and layer 1's step t are independent, so one launch runs
// both (workgroup-uniform choice between two argument sets).  Layer 0's role contracts [dz0_{t+2} | dz1_{t+1}] with
// [Wh0 ; Wx1] (K = 8H: the gradient arriving from the layer above is the second K segment instead of a hoisted
// dX product whose bf16 result is re-read by every step), layer 1's role is the plain step.  Role a (the longer K)
// owns the first `na` workgroups.  Twice the tiles of a single step per launch: with 128x128 tiles (80 KB of LDS)
// two workgroups share a CU and one's gate-derivative epilogue runs under the other's main loop.
template <class Cfg>
__global__ __launch_bounds__(Cfg::NT, 4) void lstm_bwd_pair_kernel(GemmOperands pa, LstmBwdParams ea, int tma, GemmOperands pb,
                                                                LstmBwdParams eb, int tmb, int tiles_n) {
  const int na = tma * tiles_n;
  const bool first = (int)blockIdx.x < na;           // workgroup-uniform: scalar selects, one copy of the code
  const GemmOperands p = first ? pa : pb;
  const LstmBwdParams e = first ? ea : eb;
  lstm_bwd_step_body<Cfg, false>(p, e, first ? blockIdx.x : blockIdx.x - na, first ? tma : tmb, tiles_n);
}

// "Skinny" BPTT step for M ~ batch (the L2 stacks: 256 rows, K = 4H = 4096): with a 32x32 tile per
// workgroup the LDS-staged loops above are latency-bound (64 dependent load->barrier->MFMA rounds, ~30 us
// for 2 GFLOP).  Here the K range is split over the KW waves of the workgroup and every wave loads its MFMA
// fragments STRAIGHT from global memory into registers (for v_mfma_f32_16x16x32_bf16 lane l supplies row
// l&15, k = 8*(l>>4)..+7 = one aligned 16-byte load): no LDS staging, no barrier in the loop, DEPTH K steps
// of loads in flight per wave.  The KW partial 32x32 tiles meet in LDS once, then 256 threads run the tail.
template <int KW, int DEPTH>
__global__ __launch_bounds__(64 * KW) void lstm_bwd_step_skinny_kernel(GemmOperands p, LstmBwdParams e, int tiles_m, int tiles_n) {
  constexpr int NT = 64 * KW;
  __shared__ float part[KW][32][36];                 // [wave][row][unit] (+4 pad: conflict-free float4 rows)
  const int nwg = tiles_m * tiles_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  const int tm = id % tiles_m, tn = id / tiles_m;    // consecutive ids (one XCD) share the B panel of a unit tile
  const int m0 = tm * 32, u0 = tn * 32;
  if (m0 >= e.m_active) {
    lstm_bwd_zero_tile<32, 32, NT>(e, m0, u0);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int nk = p.nk1;                              // 32-wide K steps
  const int per = (nk + KW - 1) / KW;
  const int k0 = min(wave * per, max(nk - 1, 0)), k1 = min(nk, wave * per + per);   // k0 clamped: idle waves still load in bounds
  const bf16_t* ap[2];
  const bf16_t* bp[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = min(m0 + i * 16 + fr, p.M - 1), u = min(u0 + i * 16 + fr, p.Nu - 1);
    ap[i] = p.A1 + (long)m * p.lda1 + fq * 8 + (long)k0 * 32;
    bp[i] = p.B + (long)u * p.ldb + fq * 8 + (long)k0 * 32;
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[DEPTH][2], fb[DEPTH][2];
  const int n = k1 - k0;                             // this wave's K steps (wave-uniform, may be <= 0)
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    const int kk = min(d, max(n - 1, 0));            // clamped: surplus loads re-read a valid step
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      fa[d][i] = *(const bf16x8*)(ap[i] + (long)kk * 32);
      fb[d][i] = *(const bf16x8*)(bp[i] + (long)kk * 32);
    }
  }
  for (int k = 0; k < n; k += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (k + d < n) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[d][i], fb[d][j], acc[i][j], 0, 0, 0);
      }
#ifndef EVC_ABLATE_SKINNY_LOADS
      const int kn = min(k + d + DEPTH, max(n - 1, 0));
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[d][i] = *(const bf16x8*)(ap[i] + (long)kn * 32);
        fb[d][i] = *(const bf16x8*)(bp[i] + (long)kn * 32);
      }
#endif
    }
  }
  // acc[i][j][r]: row = i*16 + fq*4 + r (A row), unit = j*16 + fr (B row)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[wave][i * 16 + fq * 4 + r][j * 16 + fr] = acc[i][j][r];
  __syncthreads();
  const int row = (threadIdx.x >> 3) & 31, ug = (threadIdx.x & 7) * 4;
  float dzv[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int g = 0; g < 4; ++g) dzv[r][g] = 0.f;
  if (threadIdx.x < 256) {
    float4 s = *(const float4*)&part[0][row][ug];
#pragma unroll
    for (int w = 1; w < KW; ++w) {
      const float4 v = *(const float4*)&part[w][row][ug];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int m = m0 + row, u = u0 + ug;
    if (m < e.M && u < e.H) {
      const float dh[4] = {s.x, s.y, s.z, s.w};
      LstmBwdIn in;
      lstm_bwd_load(e, m, u, true, in);
      lstm_bwd_finish(e, m, u, dh, in, dzv);
    }
  }
  if (e.db) {            // bias gradient (e.db is a kernel argument: uniform branch): column sums of the tile's 32 rows
    __syncthreads();     // every partial has been read
    float* cs = &part[0][0][0];                        // [32 rows][128 = 32 units x 4 gates]
    if (threadIdx.x < 256) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int g = 0; g < 4; ++g) cs[row * 128 + (ug + r) * 4 + g] = dzv[r][g];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
      float v = 0.f;
#pragma unroll 8
      for (int r = 0; r < 32; ++r) v += cs[r * 128 + threadIdx.x];
      const int u = u0 + (threadIdx.x >> 2), g = threadIdx.x & 3;
      if (u < e.H) atomicAdd(e.db + (long)g * e.H + u, v);
    }
  }
}

// Skinny BPTT step, second form: the same K split over the waves, but every wave stages its K slice through a
// PRIVATE ring of LDS-DMA stages (64-wide K steps, 128-byte rows: each 1 KiB DMA instruction moves 8 full cache
// lines, where a direct fragment load touches 16 half-used ones) and waits only on its own vmcnt - no barrier in
// the loop, 3 stages in flight per wave.  LDS: KW x STAGES x 8 KiB rings + the partial tiles.
template <int KW, int STAGES>
__global__ __launch_bounds__(64 * KW) void lstm_bwd_step_skinny_lds_kernel(GemmOperands p, LstmBwdParams e, int tiles_m, int tiles_n) {
  constexpr int NT = 64 * KW;
  constexpr int STAGE = 8192, RING = STAGES * STAGE;                  // A 32 rows x 128 B | B 32 rows x 128 B
  float (*part)[32][36] = (float (*)[32][36])lds_dyn;                 // [wave][row][unit] (+4 pad): ALIASES the rings (18 KB of KW x RING >= 64 KB),
                                                                      // written behind a barrier once every wave has left its loop
  const int nwg = tiles_m * tiles_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  const int tm = id % tiles_m, tn = id / tiles_m;
  const int m0 = tm * 32, u0 = tn * 32;
  if (m0 >= e.m_active) {
    lstm_bwd_zero_tile<32, 32, NT>(e, m0, u0);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* ring = lds_dyn + wave * RING;
  const int nk = p.nk1;                                               // 64-wide K steps
  const int per = (nk + KW - 1) / KW;
  const int k0 = min(wave * per, nk), k1 = min(nk, wave * per + per);
  const int n = k1 - k0;                                              // this wave's K steps (may be 0)
  // staging sources: chunk c = lane + i*64 -> row c>>3, physical 16-B chunk c&7 holds logical chunk (c&7)^(row&7)
  const int lc8 = ((lane & 7) ^ ((lane >> 3) & 7)) * 8;
  const bf16_t* asrc[4];
  const bf16_t* bsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (lane >> 3) + i * 8;
    const int m = min(m0 + r, p.M - 1), u = min(u0 + r, p.Nu - 1);
    asrc[i] = p.A1 + (long)m * p.lda1 + (long)k0 * 64 + lc8;
    bsrc[i] = p.B + (long)u * p.ldb + (long)k0 * 64 + lc8;
  }
  auto stage = [&](int j) {                                           // K step j of this wave -> ring slot j % STAGES
    char* sb = ring + (j % STAGES) * STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + (long)j * 64),
                                       (__attribute__((address_space(3))) void*)(sb + i * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[i] + (long)j * 64),
                                       (__attribute__((address_space(3))) void*)(sb + 4096 + i * 1024), 16, 0, 0);
  };
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int j = 0; j < STAGES - 1; ++j)
    if (j < n) stage(j);
  for (int k = 0; k < n; ++k) {
    // stage k has landed when at most the younger stages' DMAs (8 each) are outstanding
    const int younger = min(n - 1 - k, STAGES - 2);
    if (younger >= 2) wait_vmcnt<16>();
    else if (younger == 1) wait_vmcnt<8>();
    else wait_vmcnt<0>();
    const char* sb = ring + (k % STAGES) * STAGE;
    bf16x8 a[2][2], b[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = i * 16 + frow;
        const int off = row * 128 + (((kk * 4 + fq) ^ (row & 7)) << 4);
        a[kk][i] = *(const bf16x8*)(sb + off);
        b[kk][i] = *(const bf16x8*)(sb + 4096 + off);
      }
    if (k + STAGES - 1 < n) stage(k + STAGES - 1);   // refills the slot read in the previous iteration (its reads have returned)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
  }
  __syncthreads();               // every wave's last fragment reads have returned: the rings are free
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[wave][i * 16 + fq * 4 + r][j * 16 + frow] = acc[i][j][r];
  __syncthreads();
  const int row = (threadIdx.x >> 3) & 31, ug = (threadIdx.x & 7) * 4;
  float dzv[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int g = 0; g < 4; ++g) dzv[r][g] = 0.f;
  {
    float4 s = *(const float4*)&part[0][row][ug];
#pragma unroll
    for (int w = 1; w < KW; ++w) {
      const float4 v = *(const float4*)&part[w][row][ug];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int m = m0 + row, u = u0 + ug;
    if (m < e.M && u < e.H) {
      const float dh[4] = {s.x, s.y, s.z, s.w};
      LstmBwdIn in;
      lstm_bwd_load(e, m, u, true, in);
      lstm_bwd_finish(e, m, u, dh, in, dzv);
    }
  }
  if (e.db) {            // bias gradient: column sums of the tile's 32 rows
    __syncthreads();
    float* cs = &part[0][0][0];                        // [32 rows][128 = 32 units x 4 gates]
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int g = 0; g < 4; ++g) cs[row * 128 + (ug + r) * 4 + g] = dzv[r][g];
    __syncthreads();
    if (threadIdx.x < 128) {
      float v = 0.f;
#pragma unroll 8
      for (int r = 0; r < 32; ++r) v += cs[r * 128 + threadIdx.x];
      const int u = u0 + (threadIdx.x >> 2), g = threadIdx.x & 3;
      if (u < e.H) atomicAdd(e.db + (long)g * e.H + u, v);
    }
  }
}

template <class Cfg>
static inline void launch_lstm_bwd(GemmOperands p, const LstmBwdParams& e, int k1, hipStream_t st) {
  p.nk1 = k1 / kdiv<Cfg>();
  const int tm = ceil_div(e.M, Cfg::BM), tn = ceil_div(e.H, Cfg::BU);
  launch_cfg<Cfg>(lstm_bwd_step_kernel<Cfg>, tm * tn, st, p, e, tm, tn);
}

typedef TileCfg2<128, 1, 128, 2, 4, 5, true> CfgBwdV2_128;   // BPTT step tiles: BM rows x 128 units, 8 waves (2x4)
typedef TileCfg3<128, 1, 128, 2, 4, 4> CfgBwdV3_128;         // the same tile on 64-wide K stages (whole cache lines per LDS-DMA piece)
// shallower rings for the same tile (A/B, EVC_BWD_STAGES=3 | 2): 96 / 66 KB of LDS instead of 128 - room for a 64 KB workgroup of another stream on the CU
typedef TileCfg3<128, 1, 128, 2, 4, 3> CfgBwdV3_128s3;
typedef TileCfg3<128, 1, 128, 2, 4, 5> CfgBwdV3_128s5;        // (and a deeper one: the whole 160 KB)
struct CfgBwdV3_128s2 : TileCfg3<128, 1, 128, 2, 4, 2> { static constexpr int LDS_BYTES = 128 * (128 * 4 + 16); };   // (the row-major tail's dh tile: 66 KB)
template <> struct is_v2<CfgBwdV3_128s2> { static constexpr bool value = true; };
template <> struct is_v3<CfgBwdV3_128s2> { static constexpr bool value = true; };
typedef TileCfg3<64, 1, 64, 2, 4, 4> CfgBwdV3_64;            // ~1000 live rows (the student's L1 levels): 16 x 16 = 256 tiles of 64 x 64, 64 KB of LDS
typedef TileCfg2<160, 1, 128, 2, 4, 5, true> CfgBwdV2_160;
typedef TileCfg2<192, 1, 128, 2, 4, 5, true> CfgBwdV2_192;
// (128x64 and 64x128 tiles at two workgroups per CU were measured: 84-86 us vs 69 us for 128x128 at ~3800 rows -
// the extra L2->LDS traffic of the smaller tiles costs more than overlapping the epilogues gains)

extern "C" int evc_lstm_layer_bwd(const evc_bf16* w_il, const int32_t* len, int T, int M, int Kin, int H,
                                  const void* gates, const evc_bf16* c_all, const float* dS_c, const float* dS_h, int64_t ld_dS,
                                  const evc_bf16* dh_above, float* dc_ws, evc_bf16* dz4, float* db,
                                  const int32_t* row_map, const int32_t* rows_per_step, const evc_bf16* dz_above,
                                  const evc_bf16* w_above, void* stream) {
  EVC_REQUIRE(T > 0 && M > 0 && H > 0 && Kin > 0 && H % 64 == 0, EVC_ERR_BAD_SHAPE, "evc_lstm_layer_bwd: bad shape");
  EVC_REQUIRE(ring_operand_ok(M, 4L * H) && ring_operand_ok(H, 4L * H), EVC_ERR_BAD_SHAPE,
              "evc_lstm_layer_bwd: a dz time slab spans 4 GiB or more (M=%d H=%d)", M, H);
  EVC_REQUIRE((dz_above != nullptr) == (w_above != nullptr) && !(dz_above && dh_above), EVC_ERR_BAD_ARG,
              "evc_lstm_layer_bwd: dz_above and w_above come together, and instead of dh_above");
  EVC_REQUIRE(!dz_above || (((uintptr_t)dz_above % 16) == 0 && ((uintptr_t)w_above % 16) == 0 && H % 128 == 0), EVC_ERR_BAD_ALIGN,
              "evc_lstm_layer_bwd: dz_above / w_above must be 16-byte aligned, H %% 128 == 0");
  EVC_REQUIRE(gates && c_all && ((uintptr_t)gates % 16) == 0 && ((uintptr_t)c_all % 8) == 0 && ((uintptr_t)dz4 % 16) == 0,
              EVC_ERR_BAD_ALIGN, "evc_lstm_layer_bwd: gates/c_all/dz4 alignment");
  EVC_REQUIRE(ld_dS % 4 == 0 && ((uintptr_t)dS_c % 16) == 0 && ((uintptr_t)dS_h % 16) == 0 && ((uintptr_t)dc_ws % 16) == 0 &&
              (!dh_above || ((uintptr_t)dh_above % 8) == 0), EVC_ERR_BAD_ALIGN, "evc_lstm_layer_bwd: f32 operands must allow 16-byte vector access");
  if (rows_per_step)
    for (int t = 0; t < T; ++t)
      EVC_REQUIRE(rows_per_step[t] >= 0 && rows_per_step[t] <= M && (t == 0 || rows_per_step[t] <= rows_per_step[t - 1]), EVC_ERR_BAD_ARG,
                  "evc_lstm_layer_bwd: rows_per_step[%d]=%d must be non-increasing and within [0, M=%d]", t, rows_per_step[t], M);
  hipStream_t st = (hipStream_t)stream;
  for (int t = T - 1; t >= 0; --t) {
    const int Mt = rows_per_step ? rows_per_step[t] : M;    // active rows = prefix [0, Mt); the grid still covers all M
    // rows: tiles beyond Mt only zero their dz rows (the weight-gradient products contract over every row)
    // Tile choice: 256 CUs work through ceil(tiles/256) tiles each.  v2 tiles (BM x 128, LDS-DMA ring) for the
    // large steps; v1 64x64 / 32x32 (several workgroups per CU, epilogues overlap main loops) for the small ones.
    // (index 5 = the skinny kernel, chosen by rule below; 6 = 64 x 64 ring tiles on 64-wide K stages)
    static const int cand[6] = {0, 1, 2, 3, 4, 6};
    static const int bm[7] = {192, 160, 128, 64, 32, 0, 64}, bn[7] = {128, 128, 128, 64, 32, 0, 64};
    // measured: ~1000 rows x 1024 run 32 us on the v1 32x32 tiles, 38 us on v1 64x64, 23 us on the 64x64 ring tiles (256 tiles: one round)
    static const double cf[7] = {1.0, 1.0, 1.02, 2.0, 1.9, 0.0, 1.36};
    int pick = 3;
    double bc = 1e300;
    const int ma = Mt > 0 ? Mt : 1;
    for (int ci = 0; ci < 6; ++ci) {
      const int i = cand[ci];
      const double c = tile_cost((long)ceil_div(ma, bm[i]) * ceil_div(H, bn[i]), bm[i], bn[i], 1, cf[i]);
      if (c < bc) { bc = c; pick = i; }
    }
    if ((long)ceil_div(ma, 32) * ceil_div(H, 32) <= 512) pick = 5;   // M ~ batch: K split over the waves, fragments straight from global
    if (forced_tile()) pick = forced_tile() - 1;          // debug: 1 -> 192, 2 -> 160, 3 -> 128, 4 -> v1 64, 5 -> v1 32, 6 -> skinny, 7 -> ring 64x64
    if (dz_above && pick > 2) pick = 2;                   // the two-matrix K walk (B2) exists in the ring loop only
    GemmOperands p;
    p.M = M; p.Nu = H; p.group_stride = 0; p.nk1 = p.nk2 = 0;
    p.A1lo = p.A2lo = p.Blo = nullptr;
    p.A1 = dz4 + (long)(t + 1 < T ? t + 1 : t) * M * 4 * H; p.lda1 = 4L * H;   // gate-interleaved K index u*4+g
    p.A2 = p.A1; p.lda2 = p.lda1;
    p.B = w_il + (long)Kin * 4 * H; p.ldb = 4L * H;   // rows Kin..Kin+H of the kernel = Wh [H][4H], same K order
    const int k1 = (t == T - 1) ? 0 : 4 * H;
    if (dz_above) {      // [dz_{t+1} | dz_above_t] . [Wh ; Wx_above]^T: the upper layer's dX is contracted here (K = 8H)
      p.A2 = dz_above + (long)t * M * 4 * H;
      p.nk2 = 4 * H / 32;
      p.B2 = w_above;
    }
    LstmBwdParams e;
    e.len = len; e.t = t;
    e.gates = (const uint2*)gates + (long)t * M * H;
    e.c_new = c_all + (long)(t + 1) * M * H;
    e.c_old = t > 0 ? c_all + (long)t * M * H : nullptr;
    e.dS_c = dS_c; e.dS_h = dS_h; e.ld_dS = ld_dS;
    e.dh_above = dh_above ? dh_above + (long)t * M * H : nullptr;
    e.dc_ws = dc_ws; e.dz4 = (uint2*)dz4 + (long)t * M * H;
    e.dc_bf16 = bwd_dc_bf16();
    e.row_map = row_map; e.db = db; e.m_active = Mt;
    e.M = M; e.H = H; e.fused_above = dz_above ? 1 : 0;
    static const int bwd_stages = getenv("EVC_BWD_STAGES") ? atoi(getenv("EVC_BWD_STAGES")) : 4;     // A/B: ring depth of the 128 x 128 BPTT tile
    switch (pick) {
      case 0: launch_lstm_bwd<CfgBwdV2_192>(p, e, k1, st); break;
      case 1: launch_lstm_bwd<CfgBwdV2_160>(p, e, k1, st); break;
      case 2:
        if (getenv("EVC_BWD_V2_LOOP") || dz_above) launch_lstm_bwd<CfgBwdV2_128>(p, e, k1, st);   // (two-matrix K walk: nk2 is set in 32-wide steps above)
        else if (bwd_stages == 3) launch_lstm_bwd<CfgBwdV3_128s3>(p, e, k1, st);
        else if (bwd_stages == 5) launch_lstm_bwd<CfgBwdV3_128s5>(p, e, k1, st);
        else if (bwd_stages == 2) launch_lstm_bwd<CfgBwdV3_128s2>(p, e, k1, st);
        else launch_lstm_bwd<CfgBwdV3_128>(p, e, k1, st);
        break;
      case 4: launch_lstm_bwd<CfgPlainTiny>(p, e, k1, st); break;
      case 6: launch_lstm_bwd<CfgBwdV3_64>(p, e, k1, st); break;
      case 5: {
        const int tm = ceil_div(M, 32), tn = ceil_div(H, 32);
        if (getenv("EVC_SKINNY_DIRECT")) {               // first form: fragments straight from global memory
          p.nk1 = k1 / 32;
          hipLaunchKernelGGL((lstm_bwd_step_skinny_kernel<8, 4>), dim3(tm * tn), dim3(512), 0, st, p, e, tm, tn);
        } else {
          // ring depth per wave / waves per workgroup (LDS = waves x depth x 8 KiB; the partial tiles alias the rings).  Two stages = 64 KiB:
          // ALONE the step is a little slower than with four (one stage in flight per wave instead of three), but in the training step
          // these launches run next to the other towers' / the optimizer's workgroups, and a 64 KiB workgroup finds room on a CU that a
          // 146 KiB one has to wait for: 10.37 -> 10.15-10.23 ms per step (same box, alternating runs; three stages: no change)
          static const int stg = getenv("EVC_SKINNY_STAGES") ? atoi(getenv("EVC_SKINNY_STAGES")) : 2;
          p.nk1 = k1 / 64;
#define EVC_SKINNY_LAUNCH(KW_, STG_)                                                                                              \
  do {                                                                                                                          \
    allow_big_lds((const void*)lstm_bwd_step_skinny_lds_kernel<KW_, STG_>, KW_ * STG_ * 8192);                                  \
    hipLaunchKernelGGL((lstm_bwd_step_skinny_lds_kernel<KW_, STG_>), dim3(tm * tn), dim3(64 * KW_), KW_ * STG_ * 8192, st, p, e, tm, tn); \
  } while (0)
          if (stg == 3) EVC_SKINNY_LAUNCH(4, 3);           // (four waves: the tail's thread -> (row, unit) map is written for 256 threads)
          else if (stg == 4) EVC_SKINNY_LAUNCH(4, 4);
          else EVC_SKINNY_LAUNCH(4, 2);
#undef EVC_SKINNY_LAUNCH
        }
        break;
      }
      default: launch_lstm_bwd<CfgPlainSmall>(p, e, k1, st); break;
    }
  }
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// ===========================================================================
// Two-layer stack, BPTT in wavefront order (see lstm_bwd_pair_kernel): T + 1 dependent launches instead of 2T + the
// hoisted dX product of the upper layer.  Layer 0 has input width Kin0, layer 1 input width H; both kernels in the
// backward layout [in+H][4H] (4H axis gate-interleaved).  dS [M][4H] f32 = d(final state) as [c0 | h0 | c1 | h1].
// ===========================================================================
extern "C" int evc_lstm_stack2_bwd(const evc_bf16* w_il0, const evc_bf16* w_il1, const int32_t* len, int T, int M, int Kin0, int H,
                                   const void* gates0, const evc_bf16* c_all0, const void* gates1, const evc_bf16* c_all1,
                                   const float* dS, int64_t ld_dS, float* dc_ws0, float* dc_ws1, evc_bf16* dz0, evc_bf16* dz1,
                                   float* db0, float* db1, const int32_t* row_map, const int32_t* rows_per_step, void* stream) {
  EVC_REQUIRE(T > 0 && M > 0 && H > 0 && Kin0 > 0 && H % 128 == 0, EVC_ERR_BAD_SHAPE, "evc_lstm_stack2_bwd: bad shape (H %% 128)");
  EVC_REQUIRE(ring_operand_ok(M, 4L * H) && ring_operand_ok(H, 4L * H), EVC_ERR_BAD_SHAPE, "evc_lstm_stack2_bwd: a dz time slab spans 4 GiB or more");
  EVC_REQUIRE(gates0 && gates1 && c_all0 && c_all1 && dz0 && dz1 && dc_ws0 && dc_ws1 && dS, EVC_ERR_BAD_ARG, "evc_lstm_stack2_bwd: null operand");
  EVC_REQUIRE(ld_dS % 4 == 0 && ((uintptr_t)dS % 16) == 0 && ((uintptr_t)dc_ws0 % 16) == 0 && ((uintptr_t)dc_ws1 % 16) == 0 &&
              ((uintptr_t)dz0 % 16) == 0 && ((uintptr_t)dz1 % 16) == 0, EVC_ERR_BAD_ALIGN, "evc_lstm_stack2_bwd: 16-byte alignment");
  if (rows_per_step)
    for (int t = 0; t < T; ++t)
      EVC_REQUIRE(rows_per_step[t] >= 0 && rows_per_step[t] <= M && (t == 0 || rows_per_step[t] <= rows_per_step[t - 1]), EVC_ERR_BAD_ARG,
                  "evc_lstm_stack2_bwd: rows_per_step[%d]=%d must be non-increasing and within [0, M=%d]", t, rows_per_step[t], M);
  typedef CfgBwdV2_128 Cfg;
  hipStream_t st = (hipStream_t)stream;
  const long slab = (long)M * H;
  const int tn = ceil_div(H, Cfg::BU), tm = ceil_div(M, Cfg::BM);
  auto base = [&](GemmOperands& p) {
    p.M = M; p.Nu = H; p.group_stride = 0; p.nk1 = p.nk2 = 0;
    p.A1lo = p.A2lo = p.Blo = nullptr; p.B2 = nullptr;
    p.lda1 = p.lda2 = 4L * H; p.ldb = 4L * H;
  };
  auto tail = [&](LstmBwdParams& e, int layer, int t) {
    e.len = len; e.t = t;
    e.gates = (const uint2*)(layer ? gates1 : gates0) + (long)t * slab;
    const evc_bf16* ca = layer ? c_all1 : c_all0;
    e.c_new = ca + (long)(t + 1) * slab;
    e.c_old = t > 0 ? ca + (long)t * slab : nullptr;
    e.dS_c = dS + (long)(2 * layer) * H; e.dS_h = dS + (long)(2 * layer + 1) * H; e.ld_dS = ld_dS;
    e.dh_above = nullptr;
    e.dc_ws = layer ? dc_ws1 : dc_ws0;
    e.dc_bf16 = bwd_dc_bf16();
    e.dz4 = (uint2*)(layer ? dz1 : dz0) + (long)t * slab;
    e.row_map = row_map; e.db = layer ? db1 : db0;
    e.m_active = rows_per_step ? rows_per_step[t] : M;
    e.M = M; e.H = H;
    e.fused_above = layer == 0;
  };
  for (int i = 0; i <= T; ++i) {
    const int t1 = T - 1 - i, t0 = T - i;            // layer 1 runs step t1, layer 0 step t0 = t1 + 1
    GemmOperands pa, pb;
    LstmBwdParams ea, eb;
    const bool has_a = t0 <= T - 1, has_b = t1 >= 0;
    if (has_a) {                                      // layer 0, step t0: [dz0_{t0+1} | dz1_{t0}] . [Wh0 ; Wx1]^T
      base(pa);
      pa.A1 = dz0 + (long)(t0 + 1 < T ? t0 + 1 : t0) * slab * 4;
      pa.nk1 = (t0 == T - 1) ? 0 : 4 * H / 32;
      pa.A2 = dz1 + (long)t0 * slab * 4;
      pa.nk2 = 4 * H / 32;
      pa.B = w_il0 + (long)Kin0 * 4 * H;              // Wh0: rows Kin0 .. Kin0+H-1 of layer 0's kernel
      pa.B2 = w_il1;                                  // Wx1: rows 0 .. H-1 of layer 1's kernel
      tail(ea, 0, t0);
    }
    if (has_b) {                                      // layer 1, step t1: dz1_{t1+1} . Wh1^T
      base(pb);
      pb.A1 = dz1 + (long)(t1 + 1 < T ? t1 + 1 : t1) * slab * 4;
      pb.nk1 = (t1 == T - 1) ? 0 : 4 * H / 32;
      pb.A2 = pb.A1;
      pb.B = w_il1 + (long)H * 4 * H;                 // Wh1
      tail(eb, 1, t1);
    }
    if (has_a && has_b) launch_cfg<Cfg>(lstm_bwd_pair_kernel<Cfg>, 2 * tm * tn, st, pa, ea, tm, pb, eb, tm, tn);
    else if (has_a) launch_cfg<Cfg>(lstm_bwd_step_kernel<Cfg>, tm * tn, st, pa, ea, tm, tn);
    else launch_cfg<Cfg>(lstm_bwd_step_kernel<Cfg>, tm * tn, st, pb, eb, tm, tn);
  }
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}
