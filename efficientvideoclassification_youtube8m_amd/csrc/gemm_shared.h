// Shared by the GEMM-shaped translation units (evc_gemm.hip: NT products; evc_gemm_tn.hip: TN products + the fused MoE update;
// evc_lstm_fwd.hip / evc_lstm_bwd.hip: the fused LSTM step kernels): loop options, the main-loop dispatch, the LDS-transposed
// tile store.  (One file until round 5; split so that the four compile in parallel.)
#pragma once
#include "gemm_launch.h"

// Loop options per kernel (gemm_core_v2.h; same-box A/B measurements in DESIGN.md 4.2)
#ifndef EVC_FWD_STORE_POLICY
#define EVC_FWD_STORE_POLICY 0      // cache policy of the forward step's epilogue stores (evc_common.h store16<>): 0 plain, 1 sc1 (write-through), 2 nt
#endif
#ifndef EVC_FWD_TAPE_POLICY
#define EVC_FWD_TAPE_POLICY 0       // ... of its write-once tape stores alone (gate records + cell history: 36 of the 68 MB a teacher L1 launch writes)
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
  // fused squared norm of the stored f32 tile (round 6, evc_gemm_nt_sqnorm): sq_out[0] += sum over the tile of (c + sq_l2 * sq_p)^2, sq_p laid out as C
  const float* sq_p = nullptr; float sq_l2 = 0.f; float* sq_out = nullptr;
};

// Epilogue of the ring-tile (v2) kernels for a plain overwrite of C: every wave transposes its WM x WU sub-tile through
// its own slice of the (idle) LDS ring and stores whole rows of the sub-tile, 16 bytes per lane.  From the accumulator
// layout itself a store instruction touches 16-64 different lines with 2-32 bytes each, and the stores of a bf16 output
// were issue-bound: 0.41 -> 0.33 ms on 16384 x 8192 x 1152 (DESIGN.md 4.6 has the same measurement on the DBoF kernel).
// acc: TRANSPOSED accumulators (lane 16g + l: row mi*16 + l, columns ni*16 + 4g .. 4g+3).  ES = bytes per output element.
template <class Cfg, int ES, bool ATOMIC = false, bool RMW = false>     // RMW: C += tile by plain 16-byte read-modify-write (f32)
__device__ __forceinline__ void store_tile_via_lds(f32x4 (&acc)[Cfg::MI][1][Cfg::NI], char* lds, void* C, long ldc, int M, int N,
                                                   int m0, int u0, const float* bias, int row_il_H = 0,
                                                   const float* sq_p = nullptr, float sq_l2 = 0.f, float* sq_out = nullptr) {
  float sq_acc = 0.f, sq_acc_p = 0.f;     // (plain f32 stores only: the tile's contribution to |C + sq_l2 * P|^2 and |P|^2, one atomic each per wave)
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
            if constexpr (ES == 4) {
              if (sq_out) {
                float4 gq = *(const float4*)&q;
                if (sq_p) {
                  const float4 pv = *(const float4*)((const char*)sq_p + (orow * ldc + colw) * 4 + (lane % CPR) * 16);
                  gq = make_float4(gq.x + sq_l2 * pv.x, gq.y + sq_l2 * pv.y, gq.z + sq_l2 * pv.z, gq.w + sq_l2 * pv.w);
                  sq_acc_p += pv.x * pv.x + pv.y * pv.y + pv.z * pv.z + pv.w * pv.w;
                }
                sq_acc += gq.x * gq.x + gq.y * gq.y + gq.z * gq.z + gq.w * gq.w;
              }
            }
          }
        }
      }
    }
  }
  if constexpr (ES == 4 && !ATOMIC && !RMW) {
    if (sq_out) {                 // (kernel-uniform)
      // one pair of PLAIN stores per wave into its own slot of the partials workspace (same-address atomics, ~12 ns apiece, cost the cfg-5 step
      // 0.25 ms with 57 k of them); the caller's finishing launch adds the slots in index order: {|C + l2 P|^2, |P|^2}, the same bits every run
      const float t = wave_sum(sq_acc), tp = wave_sum(sq_acc_p);
      if (lane == 0) {
        float* slot = sq_out + ((long)blockIdx.x * (Cfg::NT / 64) + wave) * 2;
        slot[0] = t;
        slot[1] = tp;
      }
    }
  }
}

// plain (one column group) tiles of the NT / TN products
typedef TileCfg<128, 1, 128, 2, 2> CfgPlainBig;   // 128x128, 4 waves, 4x4 MFMA tiles per wave
typedef TileCfg<64, 1, 64, 2, 2> CfgPlainSmall;   // 64x64 for skinny problems
typedef TileCfg<32, 1, 32, 2, 2> CfgPlainTiny;    // 32x32: M ~ batch recurrent steps (256 workgroups at M=256, H=1024)
typedef TileCfg2<256, 1, 256, 2, 4, 5, true> CfgPlainV2;   // 256x256, 8 waves (2x4), 128x64 per wave, 5-deep ring (160 KiB)
typedef TileCfg2<224, 1, 256, 2, 4, 5, true> CfgPlainV2_224;   // same, 224 rows: picked when it cuts M into fewer rounds of 256 workgroups
typedef TileCfg2<320, 1, 256, 2, 4, 4, false> CfgPlainV2_320;  // 320 rows (4-deep ring, single fragment set): 5120 rows = 16 x 16 tiles, ONE round of 256 workgroups instead of 320 tiles
typedef TileCfg2<128, 1, 128, 2, 4, 5, true> CfgTn128;     // 128x128 v2 tile (80 KB ring: two workgroups per CU)
typedef TileCfg2<256, 1, 64, 2, 4, 5, true> CfgTallV2;     // 256x64: M <= 256 (batch-row) products against a long weight matrix
// (256x128 tiles + split-K 2, to halve the re-reads of the [256][K] row operand: 80 vs 61 us at N = 14148 - not the bound)

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) { return pack_bf16x2_hw(lo, hi); }

// f16 NT product of evc_gemm.hip, also the hoisted x-projection of the f16 L2 level (evc_lstm_fwd.hip)
int gemm_nt_f16(const evc_f16* A, int64_t lda, const evc_f16* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K, void* stream);
