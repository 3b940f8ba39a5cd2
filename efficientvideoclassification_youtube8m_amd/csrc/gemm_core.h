// bf16 MFMA GEMM main loop for gfx950 (CDNA4), "NT" form:
//   acc[m][n] = sum_k A[m][k] * B[n][k]      (both operands K-contiguous)
//
// Design (see DESIGN.md "GEMM core"):
//  * 64-wide wavefronts, v_mfma_f32_16x16x32_bf16; a workgroup of WR x WC waves
//    owns a BM x (G*BU) tile: G column *groups* of BU columns each.  A plain
//    GEMM has G = 1.  The LSTM step has G = 4 (gate blocks i,j,f,o of the TF
//    kernel): group g, unit u is row g*group_stride + u of B, so one lane ends
//    up holding all four gate pre-activations of (row, unit) and the cell
//    update runs in registers with no cross-lane traffic.
//  * K is walked in 64-element tiles; A may come from two row-major sources
//    back to back ([x_t | h_{t-1}] for the fused LSTM step) - nk1 tiles from
//    A1 then nk2 tiles from A2; B's k index runs on.
//  * Tiles are staged into LDS with 16-byte LDS-DMA (global_load_lds_dwordx4),
//    double buffered.  The LDS image is linear per wave-instruction (the
//    hardware requires it), rows are 128 B (64 bf16); the bank-conflict
//    XOR-swizzle (16-B chunk index ^= row & 7) is applied on the per-lane
//    SOURCE address and again on the ds_read_b128 fragment reads.
//  * Rows beyond M / units beyond Nu are clamped on load (valid memory, results
//    discarded by the epilogue), so any M, N works; K % 64 == 0 is required.
#pragma once
#include "evc_common.h"
#include <type_traits>

template <int BM_, int G_, int BU_, int WR_, int WC_>
struct TileCfg {
  static constexpr int BM = BM_, G = G_, BU = BU_, BN = G_ * BU_, WR = WR_, WC = WC_;
  static constexpr int WM = BM / WR, WU = BU / WC, MI = WM / 16, NI = WU / 16;
  static constexpr int NT = 64 * WR * WC;
  static constexpr int BK = 64;
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
  static constexpr int ACH = BM * 8 / NT, BCH = BN * 8 / NT;  // 16-B chunks per thread per tile
  static_assert(WM % 16 == 0 && WU % 16 == 0, "wave tile must be a multiple of 16x16");
  static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "staging must divide evenly");
};

struct GemmOperands {
  const bf16_t* A1; long lda1; int nk1;  // nk1 64-wide k-tiles from A1 ...
  const bf16_t* A2; long lda2; int nk2;  // ... then nk2 from A2
  const bf16_t* B;  long ldb;            // B row index = g*group_stride + unit
  long group_stride;                     // rows of B between column groups
  int M;                                 // valid rows of A
  int Nu;                                // valid units per group
  // split-bf16 ("high" precision): low-order halves with the same strides; x = hi + lo to ~16 bits
  const bf16_t* A1lo; const bf16_t* A2lo; const bf16_t* Blo;
  // v2 loop only: B rows for the A2 segment start at B2 (k index restarts at 0 there) instead of continuing behind the
  // A1 segment's columns of B - two weight matrices with the same row stride contracted back to back (the BPTT
  // wavefront: [dz0_{t+1} | dz1_t] . [Wh0 ; Wx1]^T).  nullptr: B's k index runs on.
  const bf16_t* B2 = nullptr;
  // v3 loop with LOOP_FP8_TAIL only: behind the nk1 + nk2 16-bit stages come nk3 + nk4 stages of 128 OCP e4m3 BYTES per row -
  // A3 rows (row stride lda3 BYTES), then A4 rows, against B8 rows (row stride ldb8 bytes, row index as for B; B8's k index
  // runs on from the A3 segment into the A4 segment) - contracted by v_mfma_scale_f32_16x16x128_f8f6f4 (per K element twice
  // the rate of the 16-bit stages) with every product scaled by 2^scale8_exp: the "high" mode's low-order weight halves.
  const uint8_t* A3 = nullptr; long lda3 = 0; int nk3 = 0;
  const uint8_t* A4 = nullptr; long lda4 = 0; int nk4 = 0;
  const uint8_t* B8 = nullptr; long ldb8 = 0;
  int scale8_exp = 0;
  // dynamic range of the A8 image (round 6): 64 partial |x| maxima (evc_absmax_partials) and the exponent the image was asked for; the kernel adds
  // fp8_range_drop(amax_ws, a8_hi_exp) to scale8_exp - the shift the writer of the image applied (evc_cast_f32_to_f16_fp8x_dyn).  nullptr: fixed scale.
  const float* amax_ws = nullptr; int a8_hi_exp = 0;
  // LOOP_ROW_SCALE (round 6): behind the A1 segment acc = acc * row_scale[row] + col_add[g * group_stride + unit] (+ g2_add for column group 2);
  // b8_gap: bytes of B8's rows skipped between the A3 and the A4 segment's columns (a weight image that holds a block this launch does not contract)
  const float* row_scale = nullptr; const float* col_add = nullptr; float g2_add = 0.f; long b8_gap = 0;
#ifdef EVC_STAMPS
  int stamp_slot = 0;                    // diagnostic build: which slot of evc_stamps this launch writes (gemm_core_v3.h)
#endif
};

// XCD-aware bijective remap of the linear workgroup id: consecutive remapped
// ids run on the same XCD (ids b and b+8 share an XCD's L2), so tiles that
// share an A/B panel hit in one L2 (guide 5.5 T1).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// Tile coordinates of a (remapped) linear id: walks GROUP_M rows x all columns, column-major
// inside the group, so any 32-64 consecutive ids - the workgroups resident on one XCD at a
// time - form a compact ~8x4..8x8 patch that shares A row-panels AND B column-panels in that
// XCD's L2 (12-16 panel fetches per K step instead of 33-65).  Bijective for any grid.
__device__ __forceinline__ void tile_of(int id, int tiles_m, int tiles_n, int& tm, int& tn, const int GROUP_M = 8) {
  const int per_group = GROUP_M * tiles_n;
  const int g = id / per_group, r = id - g * per_group;
  const int first_m = g * GROUP_M;
  const int rows = min(tiles_m - first_m, GROUP_M);
  tm = first_m + r % rows;
  tn = r / rows;
}

// Rows of the tile patch one XCD works on at a time when a launch has `nwg` tiles (per K split): the nwg/8
// consecutive ids of an XCD should form a gm x gn patch with gm + gn small (gm A panels + gn B panels are
// fetched into that XCD's L2 per K step).  8 is right for >= 256 tiles (8x4 patches); a split-K launch with 64
// tiles per split has only 8 tiles per XCD and split: 2x4 (6 panels) instead of 8x1 (9 panels).
__device__ __forceinline__ int patch_rows(int nwg, int tiles_n) {
  const int per = max(1, nwg >> 3);
  int gm = 1;
  while (gm * gm * 4 <= per) gm <<= 1;          // largest power of two with gm^2 <= per
  while (gm < per && per / gm > tiles_n) gm <<= 1;
  return min(gm, 8);
}

// SWAP = true issues the MFMA with the B fragment as its first operand: the accumulator tile is then
// TRANSPOSED - lane l holds column (row of A) l&15 and 4 consecutive rows (units of B)
// (l>>4)*4+reg - so an epilogue that walks units fastest gets 16-byte vector accesses.
// (Split-bf16 products are K-extensions of this same loop since round 3 - evc_gemm_nt_split - on the ring loops; the variant that
// staged hi and lo halves side by side here is gone.)
// INIT = false: the caller has pre-loaded the accumulators (e.g. with a bias) - the loop only adds to them.
// F16 = true: the operands are IEEE f16 (one v_mfma_f32_16x16x32_f16 per depth; not with SPLIT).
template <class Cfg, bool SWAP = false, bool INIT = true, bool F16 = false>
__device__ __forceinline__ void gemm_mainloop(const GemmOperands& p, const int m0, const int u0, char* lds,
                                              f32x4 (&acc)[Cfg::MI][Cfg::G][Cfg::NI]) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / Cfg::WC, wc = wave % Cfg::WC;

  if (INIT) {
#pragma unroll
    for (int mi = 0; mi < Cfg::MI; ++mi)
#pragma unroll
      for (int g = 0; g < Cfg::G; ++g)
#pragma unroll
        for (int ni = 0; ni < Cfg::NI; ++ni) acc[mi][g][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int nk = p.nk1 + p.nk2;
  if (nk == 0) return;

  // ---- per-thread staging sources ----
  // chunk c = tid + i*NT -> tile row r = c>>3, physical 16-B chunk c&7, logical
  // chunk (c&7)^(r&7).  Offsets for the CURRENT A segment live in registers and
  // are rebuilt once when the k-walk crosses from A1 to A2 (compile-time
  // indexed arrays only: runtime-selected arrays would be demoted to scratch).
  int a_row[Cfg::ACH];
  long a_off[Cfg::ACH], b_off[Cfg::BCH];
  const int lc8 = ((tid & 7) ^ ((tid >> 3) & 7)) * 8;   // NT/8 is a multiple of 8, so the same for every i
#pragma unroll
  for (int i = 0; i < Cfg::ACH; ++i) {
    int gr = m0 + ((tid + i * Cfg::NT) >> 3);
    a_row[i] = gr < p.M ? gr : p.M - 1;
  }
#pragma unroll
  for (int i = 0; i < Cfg::BCH; ++i) {
    const int r = (tid + i * Cfg::NT) >> 3;
    const int g = r / Cfg::BU, u = r % Cfg::BU;
    int gu = u0 + u;
    gu = gu < p.Nu ? gu : p.Nu - 1;
    b_off[i] = ((long)g * p.group_stride + gu) * p.ldb + lc8;
  }
  const bf16_t* a_base;
  const bf16_t* b_base = p.B;
  {
    const bool s1 = p.nk1 > 0;
    a_base = s1 ? p.A1 : p.A2;
    const long lda = s1 ? p.lda1 : p.lda2;
#pragma unroll
    for (int i = 0; i < Cfg::ACH; ++i) a_off[i] = (long)a_row[i] * lda + lc8;
  }
  constexpr int STAGE_ALL = Cfg::STAGE_BYTES;

  auto stage = [&](int buf) {
    char* sbase = lds + buf * STAGE_ALL;
#pragma unroll
    for (int i = 0; i < Cfg::ACH; ++i) {
      char* dst = sbase + (wave * 64 + i * Cfg::NT) * 16;  // wave-uniform; HW adds lane*16
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_base + a_off[i]),
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < Cfg::BCH; ++i) {
      char* dst = sbase + Cfg::A_BYTES + (wave * 64 + i * Cfg::NT) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_base + b_off[i]),
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
    a_base += 64;
    b_base += 64;
  };

  // fragment read offsets (bytes within a stage); row & 7 == lane & 7 because
  // every 16-row tile starts on a multiple of 16.
  const int frow = lane & 15, fq = lane >> 4;
  int a_rd[Cfg::MI], b_rd[Cfg::G][Cfg::NI];
#pragma unroll
  for (int mi = 0; mi < Cfg::MI; ++mi) a_rd[mi] = (wr * Cfg::WM + mi * 16 + frow) * 128;
#pragma unroll
  for (int g = 0; g < Cfg::G; ++g)
#pragma unroll
    for (int ni = 0; ni < Cfg::NI; ++ni)
      b_rd[g][ni] = Cfg::A_BYTES + (g * Cfg::BU + wc * Cfg::WU + ni * 16 + frow) * 128;
  const int sw = frow & 7;

  stage(0);
  __syncthreads();  // hipcc drains vmcnt(0) (LDS-DMA) before the barrier

  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      if (kt + 1 == p.nk1) {  // crossing from A1 to A2 (wave-uniform, at most once)
        a_base = p.A2;
#pragma unroll
        for (int i = 0; i < Cfg::ACH; ++i) a_off[i] = (long)a_row[i] * p.lda2 + lc8;
      }
      stage(cur ^ 1);
    }
    const char* sb = lds + cur * STAGE_ALL;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ch = ((ks * 4 + fq) ^ sw) * 16;
      bf16x8 af[Cfg::MI], bfr[Cfg::G][Cfg::NI];
#pragma unroll
      for (int mi = 0; mi < Cfg::MI; ++mi) af[mi] = *(const bf16x8*)(sb + a_rd[mi] + ch);
#pragma unroll
      for (int g = 0; g < Cfg::G; ++g)
#pragma unroll
        for (int ni = 0; ni < Cfg::NI; ++ni) bfr[g][ni] = *(const bf16x8*)(sb + b_rd[g][ni] + ch);
#pragma unroll
      for (int mi = 0; mi < Cfg::MI; ++mi)
#pragma unroll
        for (int g = 0; g < Cfg::G; ++g)
#pragma unroll
          for (int ni = 0; ni < Cfg::NI; ++ni)
            acc[mi][g][ni] = SWAP ? mfma16<F16>(bfr[g][ni], af[mi], acc[mi][g][ni]) : mfma16<F16>(af[mi], bfr[g][ni], acc[mi][g][ni]);
    }
    __syncthreads();
    cur ^= 1;
  }
}

// Coordinates of accumulator element (mi, ni, reg) of this lane inside the tile:
//   row  = wr*WM + mi*16 + (lane>>4)*4 + reg        (C/D map of 16x16x32: row=(lane>>4)*4+reg)
//   unit = wc*WU + ni*16 + (lane&15)                (col = lane&15)
template <class Cfg>
struct TileCoords {
  int row0, unit0;  // add mi*16 + reg / ni*16
  __device__ __forceinline__ TileCoords() {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / Cfg::WC, wc = wave % Cfg::WC;
    row0 = wr * Cfg::WM + (lane >> 4) * 4;
    unit0 = wc * Cfg::WU + (lane & 15);
  }
};
// Same for a SWAP (transposed) accumulator: element (mi, ni, reg) is
//   row  = wr*WM + mi*16 + (lane&15),   unit = wc*WU + ni*16 + (lane>>4)*4 + reg
// first tile row of wave row wr = wr * stride: WM, or ROW1 where a tile splits its rows unevenly between two wave rows (gemm_core_v3.h)
template <class Cfg, class = void> struct wave_row_stride { static constexpr int value = Cfg::WM; };
template <class Cfg> struct wave_row_stride<Cfg, std::void_t<decltype(Cfg::ROW1)>> { static constexpr int value = Cfg::ROW1; };
// row fragments that belong to wave row wr (the others of a lane's Cfg::MI accumulator rows are not part of the tile)
template <class Cfg, class = void> struct wave_row_frags { static __device__ __forceinline__ int of(int) { return Cfg::MI; } };
template <class Cfg> struct wave_row_frags<Cfg, std::void_t<decltype(Cfg::MIP)>> {
  static __device__ __forceinline__ int of(int wr) { return wr == 0 ? Cfg::MIP : Cfg::MIC; }
};
template <class Cfg>
struct TileCoordsT {
  int row0, unit0;  // add mi*16 / ni*16 + reg
  __device__ __forceinline__ TileCoordsT() {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / Cfg::WC, wc = wave % Cfg::WC;
    row0 = wr * wave_row_stride<Cfg>::value + (lane & 15);
    unit0 = wc * Cfg::WU + (lane >> 4) * 4;
  }
};
