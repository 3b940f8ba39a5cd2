// v3 main loop: the v2 ring loop with 64-wide K stages - LDS rows of 128 bytes, so that every LDS-DMA piece (1 KiB = 8 rows
// x 128 B) moves 8 WHOLE cache lines.
//
// Why (scripts/probes/ingest_probe.hip, 256 workgroups streaming an L2-resident panel the way a GEMM stages it): with 64 bytes
// per row and K step (the v2 loop: 32 bf16) a piece touches 16 lines and uses half of each, the other half is fetched again
// one K step later (32 KB of other lines have gone through the CU's L1 in between): 63-68 GB/s per CU, and the 128 x 128
// BPTT step's main loop sat exactly there (71 GB/s).  With 128 bytes per row and stage: 100-110 GB/s per CU.
//
//  * stage = [BM + BN rows][64 bf16]; 16-byte chunk c of row r is stored at chunk c ^ (r & 7) (applied to the DMA source
//    address and to the fragment reads): every ds_read_b128 lane group hits 16 distinct 16-byte slots of the 256-byte bank row
//    (the swizzle of the skinny BPTT kernel; for lane group {0-3, 12-15, 20-27}: slots 0 9 2 11 | 4 13 6 15 | 5 12 7 14 1 8 3 10).
//  * one trip of the loop = one stage = two MFMA depths (kb = 0, 1): the first half-step needs no wait and no barrier (its
//    fragments are in the stage that is already being read); the second waits for the next stage (counted vmcnt), passes the
//    only barrier of the trip and refills the slot of the stage just consumed - STAGES-1 stages stay in flight.
//  * operands, column groups, SWAP, clamping and the MODE options as in gemm_core_v2.h; K % 64 == 0 (nk1 / nk2 count 64-wide
//    stages here).
//  * LOOP_FP8_TAIL: after the 16-bit stages the ring carries on with stages of 128 e4m3 BYTES per row (GemmOperands::A3 / A4 /
//    B8: the same 128-byte LDS rows, the same LDS-DMA pieces, the same swizzle).  One v_mfma_scale_f32_16x16x128_f8f6f4 per
//    fragment pair takes BOTH 16-byte chunks of a lane's row (chunks fq and 4 + fq; the K position of an operand byte is the
//    same function of lane group and register byte for A and B - scripts/probes/fp8_mfma_probe.hip - so the pairs meet) and runs
//    1.77x as long as one 16-bit MFMA for 4x the K.  The two halves of such a trip split the tile's ROW fragments instead of
//    the stage's K range: half 1 multiplies the lower row fragments while the upper ones are read, half 2 (behind the barrier
//    and the refill) the upper ones while the next stage's lower row fragments and - column group by column group, as its last
//    MFMA has been issued - its B fragments are read: the same register budget as the 16-bit trips.
#pragma once
#include "gemm_core_v2.h"

#ifndef EVC_V3_AUX_A
#define EVC_V3_AUX_A 0      // cache policy of the ring's LDS-DMA loads (2 = nt), A / B operand: measured in round 4 (dropped), default policy kept
#endif
#ifndef EVC_V3_AUX_B
#define EVC_V3_AUX_B 0
#endif
#ifdef EVC_STAMPS      // diagnostic build (scripts/fwd_stamps.py): s_memrealtime (100 MHz) marks per workgroup, read back through evc_debug_read_stamps
static __device__ unsigned long long evc_stamps[8][512][8];       // [GemmOperands::stamp_slot][workgroup][mark]
#define EVC_STAMP(slot, k) do { if (threadIdx.x == 0 && blockIdx.x < 512) evc_stamps[(slot) & 7][blockIdx.x][k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define EVC_STAMP(slot, k) do { } while (0)
#endif
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v8i_t __attribute__((ext_vector_type(8)));

// MIP_ > 0: UNEVEN row split between the two wave rows (WR = 2, LOOP_PRODUCER loops): the producer waves (wave row 0: they also issue the
// stage's LDS-DMA) take MIP_ row fragments, the other wave row the remaining BM/16 - MIP_ - the two waves of a SIMD are one of each, so the
// SIMD's MFMA work is BM/16 fragments whatever the split, and BM moves in steps of 16 rows instead of 32 (240 = 7 + 8: DESIGN.md 4.3)
template <int BM_, int G_, int BU_, int WR_, int WC_, int STAGES_ = 4, int MIP_ = 0>
struct TileCfg3 {
  static constexpr int BM = BM_, G = G_, BU = BU_, BN = G_ * BU_, WR = WR_, WC = WC_;
  static constexpr bool UNEVEN = MIP_ > 0;
  static constexpr int MIP = UNEVEN ? MIP_ : BM / WR / 16, MIC = UNEVEN ? BM / 16 - MIP_ : MIP;    // row fragments of wave row 0 / of the other wave rows
  static constexpr int MI = MIP > MIC ? MIP : MIC, WM = UNEVEN ? MI * 16 : BM / WR, ROW1 = UNEVEN ? MIP * 16 : WM;   // wave row wr starts at tile row wr * ROW1
  static constexpr int WU = BU / WC, NI = WU / 16;
  static_assert(!UNEVEN || (WR == 2 && WC == 4 && BM % 16 == 0 && MIC > 0), "uneven split: two wave rows of four waves (wave row 0 = the producers)");
  static constexpr int NT = 64 * WR * WC;
  static constexpr int BK = 64, STAGES = STAGES_;
  static constexpr bool PIPE = true;
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr bool RAGGED = (BM / 8) % (NT / 64) != 0 || (BN / 8) % (NT / 64) != 0 || (BM / 8) % 4 != 0 || (BN / 8) % 4 != 0;
  static constexpr int DUMMY_OFF = STAGES * STAGE_BYTES;       // 1 KiB sink for the surplus lanes of a ragged last round
#ifdef EVC_V3_PREFETCH_SINK
  static constexpr bool SINK = RAGGED || DUMMY_OFF + 1024 <= 160 * 1024;     // (LOOP_PREFETCH builds: the sink also takes the prefetch dwords)
#else
  static constexpr bool SINK = RAGGED;
#endif
  static constexpr int LDS_BYTES = DUMMY_OFF + (SINK ? 1024 : 0);
  static_assert((UNEVEN || BM % (16 * WR) == 0) && WU % 16 == 0 && BM % 8 == 0 && BN % 8 == 0, "wave tile must be a multiple of 16x16");
  static_assert(STAGES >= 2 && STAGES <= 6, "ring depth 2..6");
  static_assert(LDS_BYTES <= 160 * 1024, "exceeds the 160 KiB LDS of a CU");
};

// PHASE (round 5, persistent kernels that walk several output tiles per workgroup): 0 = the whole loop; 1 = ONLY issue the first
// min(nk, STAGES) stages of tile (m0, u0) and return (acc untouched) - called for the NEXT tile once every wave has left the ring
// (a barrier), so that the ring fills under the current tile's epilogue; 2 = the loop for a tile whose first stages were issued by a
// PHASE-1 call (same p, m0, u0): nothing is issued in the prologue and the first wait is vmcnt(0) - the wave's epilogue stores were
// issued after those stages and a counted wait would count them instead.
template <class Cfg, bool SWAP = false, bool INIT = true, int MODE = EVC_LOOP_MODE_DEFAULT, int PHASE = 0>
__device__ __forceinline__ void gemm_mainloop_v3(const GemmOperands& p, const int m0, const int u0, char* lds,
                                                 f32x4 (&acc)[Cfg::MI][Cfg::G][Cfg::NI]) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / Cfg::WC, wc = wave % Cfg::WC;

  // (with LOOP_FP8_TAIL the first STAGES stages of a tile are 16-bit stages too - the launchers keep nk1 >= STAGES - so the PHASE split of the
  //  prologue is the same)
  if (INIT && PHASE != 1) {
#pragma unroll
    for (int mi = 0; mi < Cfg::MI; ++mi)
#pragma unroll
      for (int g = 0; g < Cfg::G; ++g)
#pragma unroll
        for (int ni = 0; ni < Cfg::NI; ++ni) acc[mi][g][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  constexpr bool FP8 = (MODE & LOOP_FP8_TAIL) != 0;
  const int nkf = p.nk1 + p.nk2;                       // 16-bit stages (64 elements wide)
  const int nk = nkf + (FP8 ? p.nk3 + p.nk4 : 0);      // + e4m3 stages (128 elements wide)
  if (nk == 0) return;

  // ---- staging: piece q (1 KiB) = tile rows 8q .. 8q+7, 128 bytes each; lane -> row lane>>3, physical chunk lane&7 ----
  const int lc8 = ((lane & 7) ^ (lane >> 3)) * 8;              // logical chunk (in elements) this lane fetches
  constexpr bool PRODUCERS = (MODE & LOOP_PRODUCER) != 0 && Cfg::NT == 512;
  constexpr int NPW = PRODUCERS ? 4 : Cfg::NT / 64;
  const bool producer = wave < NPW;
  constexpr int ACH = (Cfg::BM / 8 + NPW - 1) / NPW, BCH = (Cfg::BN / 8 + NPW - 1) / NPW;
  constexpr int PER = ACH + BCH;
  static_assert(Cfg::RAGGED || (ACH * NPW * 8 == Cfg::BM && BCH * NPW * 8 == Cfg::BN), "surplus pieces need the dummy sink");
  int a_row[ACH];              // (addressing as in gemm_core_v2.h: wave-uniform base + 32-bit lane byte offset)
  uint32_t b_vo[BCH];          // FP8: the B row index (two row strides: the byte offset is formed per stage, as for A)
  int a_dst[ACH], b_dst[BCH];   // wave-uniform LDS byte offsets within a stage (or the dummy sink)
#pragma unroll
  for (int i = 0; i < ACH; ++i) {
    const int q = (wave % NPW) + i * NPW;
    const bool live = q < Cfg::BM / 8;                       // wave-uniform
    const int r = live ? q * 8 + (lane >> 3) : 0;
    const int gr = m0 + r;
    a_row[i] = gr < p.M ? gr : p.M - 1;
    a_dst[i] = live ? q * 1024 : -1;
  }
#pragma unroll
  for (int i = 0; i < BCH; ++i) {
    const int q = (wave % NPW) + i * NPW;
    const bool live = q < Cfg::BN / 8;
    const int r = live ? q * 8 + (lane >> 3) : 0;
    const int g = r / Cfg::BU, u = r % Cfg::BU;
    int gu = u0 + u;
    gu = gu < p.Nu ? gu : p.Nu - 1;
    b_vo[i] = FP8 ? (uint32_t)((long)g * p.group_stride + gu) : (uint32_t)((((long)g * p.group_stride + gu) * p.ldb + lc8) * 2);
    b_dst[i] = live ? Cfg::A_BYTES + q * 1024 : -1;
  }
  const bf16_t* const b2 = p.B2 ? p.B2 - (long)p.nk1 * 64 : p.B;   // base such that b2 + ks*64 addresses the A2 segment's B columns
  int ks_issue = 0;                    // next stage to issue
  int slot_issue = 0, slot_read = 0;

  auto stage = [&]() {                 // branch-free (scalar selects only)
    const bool s1 = ks_issue < p.nk1;
    const char* ab = (const char*)(s1 ? p.A1 + (long)ks_issue * 64 : p.A2 + (long)(ks_issue - p.nk1) * 64);
    uint32_t lda_b = (uint32_t)(s1 ? p.lda1 : p.lda2) * 2u;
    const char* b_base = (const char*)((s1 ? p.B : b2) + (long)ks_issue * 64);
    uint32_t ldb_b = (uint32_t)p.ldb * 2u;
    uint32_t chunk_b = (uint32_t)(lc8 * 2);
    if constexpr (FP8) {               // stages behind the 16-bit ones: rows of 128 e4m3 bytes
      const int k8 = ks_issue - nkf;
      const bool f = k8 < 0, s3 = k8 < p.nk3;
      const char* ab8 = s3 ? (const char*)p.A3 + (long)k8 * 128 : (const char*)p.A4 + (long)(k8 - p.nk3) * 128;
      ab = f ? ab : ab8;
      lda_b = f ? lda_b : (uint32_t)(s3 ? p.lda3 : p.lda4);
      b_base = f ? b_base : (const char*)p.B8 + (long)k8 * 128 + (s3 ? 0 : p.b8_gap);
      ldb_b = f ? ldb_b : (uint32_t)p.ldb8;
#ifdef EVC_ABLATE_FP6   // TIMING ablation (wrong results): what an e2m3 tail could cost - the e-stages fetch DENSE stage-major rows of 96 bytes
      // (lanes 6, 7 of a row repeat chunk 5: 96 of the 128 bytes per row come from memory) and the MFMAs run in the FP6 formats
      ab = f ? ab : (s3 ? (const char*)p.A3 + (long)k8 * p.M * 96 : (const char*)p.A4 + (long)(k8 - p.nk3) * p.M * 96);
      lda_b = f ? lda_b : 96u;
      b_base = f ? b_base : (const char*)p.B8 + (long)k8 * (4 * p.group_stride) * 96;
      ldb_b = f ? ldb_b : 96u;
      chunk_b = f ? (uint32_t)(lc8 * 2) : (uint32_t)(min(lc8 / 8, 5) * 16);
#endif
#ifdef EVC_ABLATE_E_HOT   // TIMING ablation (wrong results): every row of an e-stage fetches row 0's bytes - L2-hot lines, the same LDS-DMA count
      lda_b = f ? lda_b : 0u;
      ldb_b = f ? ldb_b : 0u;
#endif
    }
#ifdef EVC_ABLATE_E_NOA   // TIMING ablation (wrong results): the e-stages issue their B pieces only (half the LDS-DMA instructions)
    const bool skip_a = FP8 && ks_issue >= nkf;
#else
    constexpr bool skip_a = false;
#endif
    char* sbase = lds + slot_issue * Cfg::STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      if (skip_a) break;
      char* dst = a_dst[i] >= 0 ? sbase + a_dst[i] : lds + Cfg::DUMMY_OFF;
      const uint32_t vo = __umul24((uint32_t)a_row[i], lda_b) + chunk_b;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ab + vo),
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, EVC_V3_AUX_A);
    }
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
      char* dst = b_dst[i] >= 0 ? sbase + b_dst[i] : lds + Cfg::DUMMY_OFF;
      const uint32_t vo = FP8 ? __umul24(b_vo[i], ldb_b) + chunk_b : b_vo[i];
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_base + vo),
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, (MODE & LOOP_B_NT) ? 2 : EVC_V3_AUX_B);
    }
    ++ks_issue;
    slot_issue = (slot_issue + 1 == Cfg::STAGES) ? 0 : slot_issue + 1;
  };

  // fragment read offsets within a stage: kb = 0 as computed, kb = 1 = the same offset with byte bit 6 flipped
  // (chunk (4 + fq) ^ s = (fq ^ s) ^ 4 for fq < 4)
  const int frow = lane & 15, fq = lane >> 4;
  const int fch = (fq ^ (frow & 7)) * 16;      // tile rows start on multiples of 16 -> row & 7 == frow & 7
  // one base per operand and K half; fragment (mi | g, ni) sits a COMPILE-TIME multiple of 128 bytes behind it (adding a multiple of
  // 128 commutes with flipping bit 6): immediates of the ds_read instead of one address register per fragment and half
  const int a_rd0 = (wr * Cfg::ROW1 + frow) * 128 + fch, b_rd0 = Cfg::A_BYTES + (wc * Cfg::WU + frow) * 128 + fch;
  const int a_rd1 = a_rd0 ^ 64, b_rd1 = b_rd0 ^ 64;
  auto a_off = [](const int mi) { return mi * 16 * 128; };
  auto b_off = [](const int g, const int ni) { return (g * Cfg::BU + ni * 16) * 128; };

  // (mi_tag: the row fragments of the calling wave's role - Cfg::MI unless the tile splits its rows unevenly)
  auto read_half = [&](auto mi_tag, const int kb, bf16x8 (&af)[Cfg::MI], bf16x8 (&bfr)[Cfg::G][Cfg::NI]) {   // reads half kb of ring slot slot_read
    constexpr int MIr = decltype(mi_tag)::value;
    const char* sb = lds + slot_read * Cfg::STAGE_BYTES;
    const char* pa = sb + (kb ? a_rd1 : a_rd0);
    const char* pb = sb + (kb ? b_rd1 : b_rd0);
#pragma unroll
    for (int g = 0; g < Cfg::G; ++g)
#pragma unroll
      for (int ni = 0; ni < Cfg::NI; ++ni) bfr[g][ni] = *(const bf16x8*)(pb + b_off(g, ni));
#pragma unroll
    for (int mi = 0; mi < MIr; ++mi) af[mi] = *(const bf16x8*)(pa + a_off(mi));
  };
  auto next_slot = [&]() { slot_read = (slot_read + 1 == Cfg::STAGES) ? 0 : slot_read + 1; };
  auto mfma_all = [&](auto mi_tag, const bf16x8 (&af)[Cfg::MI], const bf16x8 (&bfr)[Cfg::G][Cfg::NI]) {
#pragma unroll
    for (int mi = 0; mi < decltype(mi_tag)::value; ++mi)
#pragma unroll
      for (int g = 0; g < Cfg::G; ++g)
#pragma unroll
        for (int ni = 0; ni < Cfg::NI; ++ni)
          acc[mi][g][ni] = SWAP ? mfma16<(MODE & LOOP_F16) != 0>(bfr[g][ni], af[mi], acc[mi][g][ni])
                                : mfma16<(MODE & LOOP_F16) != 0>(af[mi], bfr[g][ni], acc[mi][g][ni]);
  };
  auto end_of_step = [&]() {          // see gemm_core_v2.h: retire the LDS reads explicitly, nothing loop-carried for hipcc
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
  };
  constexpr int AHEAD = Cfg::STAGES - 2;     // stages that may stay in flight while the next one is awaited
  auto wait_landed = [&](int outstanding_stages) {   // wave-uniform small switch
    if (outstanding_stages >= 4) wait_vmcnt<4 * PER>();
    else if (outstanding_stages == 3) wait_vmcnt<3 * PER>();
    else if (outstanding_stages == 2) wait_vmcnt<2 * PER>();
    else if (outstanding_stages == 1) wait_vmcnt<PER>();
    else wait_vmcnt<0>();
  };

  // ---- LOOP_PREFETCH: which lines this wave touches ahead of the ring.  Within an XCD's patch of tiles (tile_of: 8 row tiles x ~4 column
  // tiles) an A row panel is shared by the column tiles, a B column panel by the row tiles: the workgroup with tn % 4 == q takes quarter q of its A
  // rows (wave NPW), the one with tm % 8 == q eighth q of its B rows (wave NPW + 1); a share nobody takes just stays a demand miss.
#ifndef EVC_PREFETCH_DIST
#define EVC_PREFETCH_DIST 2
#endif
  constexpr bool PREFETCH = (MODE & LOOP_PREFETCH) != 0 && PRODUCERS && Cfg::SINK && !FP8;
  long pf_off1 = -1, pf_off2 = -1;       // byte offset of this lane's row in the A1 / A2 segment (wave NPW) or in B (wave NPW + 1); -1: no duty
  if constexpr (PREFETCH) {
    const int tm_ = m0 / Cfg::BM, tn_ = u0 / Cfg::BU;
    if (wave == NPW) {
      constexpr int QA = (Cfg::BM / 4 + 7) / 8 * 8;
      const int r = (tn_ & 3) * QA + lane;
      if (lane < QA && r < Cfg::BM) {
        int gr = m0 + r;
        gr = gr < p.M ? gr : p.M - 1;
        pf_off1 = (long)gr * p.lda1 * 2;
        pf_off2 = (long)gr * p.lda2 * 2;
      }
    } else if (wave == NPW + 1) {
      constexpr int QB = Cfg::BN / 8;
      const int r = (tm_ & 7) * QB + lane;
      if (lane < QB) {
        const int g = r / Cfg::BU, u = r % Cfg::BU;
        int gu = u0 + u;
        gu = gu < p.Nu ? gu : p.Nu - 1;
        pf_off1 = pf_off2 = ((long)g * p.group_stride + gu) * p.ldb * 2;
      }
    }
  }
  auto prefetch = [&](int ks) {          // one dword of every line of stage ks (wave-uniform ks; lanes without a duty are masked)
    if constexpr (PREFETCH) {
      if (ks < nkf && wave <= NPW + 1) {
        const bool s1 = ks < p.nk1;
        const char* base = wave == NPW ? (const char*)(s1 ? p.A1 + (long)ks * 64 : p.A2 + (long)(ks - p.nk1) * 64)
                                       : (const char*)((s1 ? p.B : b2) + (long)ks * 64);
        const long off = s1 ? pf_off1 : pf_off2;
        if (off >= 0)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off + (lane & 31) * 4),
                                           (__attribute__((address_space(3))) void*)(lds + Cfg::DUMMY_OFF), 4, 0, 0);
      }
    }
  };

  auto run = [&](auto prod_tag) {     // one copy of the loop per role (LOOP_PRODUCER), each steady-state body branch-free
  constexpr bool PROD = decltype(prod_tag)::value;
  constexpr bool PRIO = (MODE & LOOP_NO_PRIO) == 0;
  constexpr int PERX = PROD ? PER : 0;
  static_assert(!Cfg::UNEVEN || PRODUCERS, "an uneven row split needs the producer / consumer roles of LOOP_PRODUCER");
  constexpr int MIr = Cfg::UNEVEN ? (PROD ? Cfg::MIP : Cfg::MIC) : Cfg::MI;      // row fragments of this role
  using MIT = std::integral_constant<int, MIr>;
  constexpr int NMFMA = MIr * Cfg::G * Cfg::NI, NREAD = MIr + Cfg::G * Cfg::NI;
  auto stage_role = [&]() {
    if constexpr (PROD) stage();
  };
  // EVC_STAGGER_LEAD (experiment, MI355X_MICROARCH.md "Two waves per SIMD" item 9 adapted to a loop whose two halves are alike): the waves
  // that issue no LDS-DMA (4-7: the SIMD partners of the producers) open the first half-step with LEAD bare MFMAs and read their
  // fragments behind them, so that the partners' LDS read bursts do not start together at the barrier.
#ifndef EVC_STAGGER_LEAD
#define EVC_STAGGER_LEAD 0
#endif
  auto interleave = [&](auto ndma_tag, auto lead_tag) {   // MFMAs with one LDS read / LDS-DMA between small groups of them
    constexpr int ndma = decltype(ndma_tag)::value;
    constexpr int lead = (decltype(lead_tag)::value < NMFMA - (NREAD + ndma)) ? decltype(lead_tag)::value : 0;
    if constexpr (lead > 0) __builtin_amdgcn_sched_group_barrier(0x008, lead, 0);
    constexpr int per = (NMFMA - lead) / (NREAD + ndma) > 0 ? (NMFMA - lead) / (NREAD + ndma) : 1;
    if constexpr ((MODE & LOOP_DMA_FIRST) != 0) {
#pragma unroll
      for (int i = 0; i < ndma; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, per, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // VMEM read (LDS-DMA)
      }
    }
#pragma unroll
    for (int i = 0; i < NREAD; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, per, 0);     // MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // DS read
    }
    if constexpr ((MODE & LOOP_DMA_FIRST) == 0) {
#pragma unroll
      for (int i = 0; i < ndma; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, per, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - lead - per * (NREAD + ndma), 0);
  };

  // ---- prologue: every slot of the ring in flight ----
  if constexpr (PHASE != 2) {
#pragma unroll
    for (int i = 0; i < Cfg::STAGES; ++i)
      if (i < nk) stage_role();
    if constexpr (PHASE == 1) return;
    if constexpr (PROD) wait_landed(min(nk, Cfg::STAGES) - 1);
  } else {                             // issued by the PHASE-1 call: only the bookkeeping
    ks_issue = min(nk, Cfg::STAGES);
    slot_issue = ks_issue == Cfg::STAGES ? 0 : ks_issue;
    if constexpr (PROD) wait_vmcnt<0>();
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  EVC_STAMP(p.stamp_slot, 1);
  bf16x8 afA[Cfg::MI], bfA[Cfg::G][Cfg::NI], afB[Cfg::MI], bfB[Cfg::G][Cfg::NI];
  read_half(MIT{}, 0, afA, bfA);
  __builtin_amdgcn_s_waitcnt(0xC07F);   // enter the loop with no LDS read pending

  int j = 0;
  // one steady-state trip over a 16-bit stage (a later stage exists and is refilled into the slot this trip frees);
  // prefetch: read the first half of the next stage's fragments behind the barrier (false: the next stage is an e4m3 stage)
  auto trip16 = [&](auto prefetch_tag) {
    // first half: its partner fragments are in the stage being read - no wait, no barrier
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
    read_half(MIT{}, 1, afB, bfB);
    mfma_all(MIT{}, afA, bfA);
#ifndef EVC_NO_INTERLEAVE
    interleave(std::integral_constant<int, 0>{}, std::integral_constant<int, (PRODUCERS && !PROD) ? EVC_STAGGER_LEAD : 0>{});
#endif
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    end_of_step();
    // second half: the next stage must have landed; after the barrier every wave has read all of this stage -> refill its slot
    if constexpr (PROD) wait_vmcnt<AHEAD * PER>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
    stage_role();
    if constexpr (!PROD) prefetch(j + Cfg::STAGES + EVC_PREFETCH_DIST);
    next_slot();
    if constexpr (decltype(prefetch_tag)::value) read_half(MIT{}, 0, afA, bfA);
    mfma_all(MIT{}, afB, bfB);
#ifndef EVC_NO_INTERLEAVE
    if constexpr (decltype(prefetch_tag)::value) interleave(std::integral_constant<int, PERX>{}, std::integral_constant<int, 0>{});
#endif
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    end_of_step();
  };

  if constexpr (!FP8) {
    for (; j + Cfg::STAGES < nk; ++j) trip16(std::true_type{});   // steady state: stage j+STAGES exists, so every trip refills
    for (; j < nk; ++j) {                 // last STAGES stages: no refills
      read_half(MIT{}, 1, afB, bfB);
      mfma_all(MIT{}, afA, bfA);
      end_of_step();
      if (j + 1 < nk) {
        if constexpr (PROD) wait_landed(nk - (j + 2));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        next_slot();
        read_half(MIT{}, 0, afA, bfA);
      }
      mfma_all(MIT{}, afB, bfB);
      end_of_step();
    }
  } else {
    // ---- 16-bit stages (every one of them is followed by >= STAGES more stages: all trips refill), then the e4m3 stages ----
    if constexpr ((MODE & LOOP_ROW_SCALE) != 0) {
      static_assert(SWAP, "LOOP_ROW_SCALE is written for transposed accumulators (lane = one row, 4 consecutive units)");
      // between two trips every MFMA of the stages before is issued into acc and the next stage's fragments are only READ: a plain VALU pass over
      // the accumulators (MI x G x NI x 4 FMAs per lane, once per tile) is all the boundary costs
      auto rescale = [&]() {
        TileCoordsT<Cfg> tc;
        float rsv[Cfg::MI];
#pragma unroll
        for (int mi = 0; mi < Cfg::MI; ++mi) {
          const int m = m0 + tc.row0 + mi * 16;
          rsv[mi] = m < p.M ? p.row_scale[m] : 0.f;
        }
#pragma unroll
        for (int ni = 0; ni < Cfg::NI; ++ni) {
          const int u = min(u0 + tc.unit0 + ni * 16, p.Nu - 4);
#pragma unroll
          for (int g = 0; g < Cfg::G; ++g) {
            const float4 b = *(const float4*)(p.col_add + (long)g * p.group_stride + u);
            const float fb = g == 2 ? p.g2_add : 0.f;
#pragma unroll
            for (int mi = 0; mi < Cfg::MI; ++mi) {
              f32x4& a = acc[mi][g][ni];
              a = f32x4{a[0] * rsv[mi] + (b.x + fb), a[1] * rsv[mi] + (b.y + fb), a[2] * rsv[mi] + (b.z + fb), a[3] * rsv[mi] + (b.w + fb)};
            }
          }
        }
      };
      for (; j + 1 < nkf && j < p.nk1; ++j) trip16(std::true_type{});      // the A1 segment (all of it when an A2 segment follows)
      if (j == p.nk1) rescale();                                           // ... an A2 segment follows: rescale between the two
      for (; j + 1 < nkf; ++j) trip16(std::true_type{});
      trip16(std::false_type{});
      ++j;
      if (p.nk2 == 0) rescale();                                           // no A2 segment (t = 0): behind the last 16-bit stage, before the e4m3 ones
    } else {
    for (; j + 1 < nkf; ++j) trip16(std::true_type{});
    trip16(std::false_type{});
    ++j;
    }
    // Two halves per trip, split by the tile's ROW fragments (the 16-bit trips split the stage's K range): half 1 multiplies the lower
    // row fragments while the upper ones are read; behind the barrier and the refill, half 2 multiplies the upper ones while the next
    // stage's lower row fragments and - column group by column group, as its last MFMA has been issued - its B fragments are read.
    constexpr int ML = (MIr + 1) / 2, MH = MIr - ML;
    // (a wave with ONE row fragment - the 64-row tiles of the M ~ batch stacks - has no upper half: half 1 multiplies everything, half 2
    //  only re-reads; those steps are bound by the chain of dependent stages, and an e4m3 stage covers twice the K of a 16-bit one)
    v8i_t aL[ML], aH[MH > 0 ? MH : 1], b8[Cfg::G][Cfg::NI];
    // e8m0 scale bytes: the whole factor rides on the first operand (+ the A8 image's dynamic range shift, if it has one)
    const int sc_first = 127 + p.scale8_exp + (p.amax_ws ? fp8_range_drop(p.amax_ws, p.a8_hi_exp) : 0), sc_second = 127;
    auto rd8 = [&](const int base0, const int base1, const int off) -> v8i_t {   // both 16-byte chunks of this lane's row in ring slot slot_read
      const char* sb = lds + slot_read * Cfg::STAGE_BYTES;
      const v4i_t lo = *(const v4i_t*)(sb + base0 + off), hi = *(const v4i_t*)(sb + base1 + off);
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
#ifdef EVC_ABLATE_FP6
    constexpr int FMT8 = 2;             // e2m3 (timing ablation)
#else
    constexpr int FMT8 = 0;             // e4m3
#endif
    auto mfma8 = [&](const v8i_t& a, const v8i_t& b, f32x4& c) {
      c = SWAP ? __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b, a, c, FMT8, FMT8, 0, sc_first, 0, sc_second)
               : __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, FMT8, FMT8, 0, sc_first, 0, sc_second);
    };
    auto rd_lower = [&]() {
#pragma unroll
      for (int mi = 0; mi < ML; ++mi) aL[mi] = rd8(a_rd0, a_rd1, a_off(mi));
    };
    auto lower = [&]() {                  // half 1: upper row fragments arrive under the lower ones' MFMAs
#pragma unroll
      for (int mi = 0; mi < MH; ++mi) aH[mi] = rd8(a_rd0, a_rd1, a_off(ML + mi));
#pragma unroll
      for (int g = 0; g < Cfg::G; ++g)
#pragma unroll
        for (int ni = 0; ni < Cfg::NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < ML; ++mi) mfma8(aL[mi], b8[g][ni], acc[mi][g][ni]);
    };
    auto upper = [&](auto reload_tag) {   // half 2; a B fragment is re-read (next stage) once its last MFMA is out
#pragma unroll
      for (int g = 0; g < Cfg::G; ++g)
#pragma unroll
        for (int ni = 0; ni < Cfg::NI; ++ni) {
#pragma unroll
          for (int mi = 0; mi < MH; ++mi) mfma8(aH[mi], b8[g][ni], acc[ML + mi][g][ni]);
          if constexpr (decltype(reload_tag)::value) b8[g][ni] = rd8(b_rd0, b_rd1, b_off(g, ni));
        }
    };
    auto pattern = [&](auto nmfma_tag, auto nread_tag, auto ndma_tag) {   // LDS-DMAs and fragment reads spread behind the MFMAs
      constexpr int nm = decltype(nmfma_tag)::value, nr = decltype(nread_tag)::value, nd = decltype(ndma_tag)::value;
      constexpr int rper = nm > 0 ? (nr + nm - 1) / nm : 0, dper = nm > 0 ? (nd + nm - 1) / nm : 0;
#pragma unroll
      for (int i = 0; i < nm; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (dper > 0) __builtin_amdgcn_sched_group_barrier(0x020, dper, 0);
        if constexpr (rper > 0) __builtin_amdgcn_sched_group_barrier(0x100, rper, 0);
      }
    };
    using I0 = std::integral_constant<int, 0>;
    constexpr int GN = Cfg::G * Cfg::NI;
    // the first e4m3 stage's lower row fragments and B fragments (once per launch: nothing to hide these reads behind)
    rd_lower();
#pragma unroll
    for (int g = 0; g < Cfg::G; ++g)
#pragma unroll
      for (int ni = 0; ni < Cfg::NI; ++ni) b8[g][ni] = rd8(b_rd0, b_rd1, b_off(g, ni));
    __builtin_amdgcn_s_waitcnt(0xC07F);
    auto half1 = [&]() {
      lower();
#ifndef EVC_NO_INTERLEAVE
      pattern(std::integral_constant<int, ML * GN>{}, std::integral_constant<int, 2 * MH>{}, I0{});
#endif
      // hipcc otherwise SINKS these MFMAs below the barrier (their results are first read a trip later): all 32 of a trip then sit
      // behind the barrier, the reads of both halves stand alone in front of an lgkmcnt(0), and the two waves of a SIMD - in step
      // through the barrier - wait for LDS together
#pragma unroll
      for (int mi = 0; mi < ML; ++mi)
#pragma unroll
        for (int g = 0; g < Cfg::G; ++g)
#pragma unroll
          for (int ni = 0; ni < Cfg::NI; ++ni) asm volatile("" : "+v"(acc[mi][g][ni]));
      end_of_step();                      // every fragment of this stage is in registers
    };
    for (; j + Cfg::STAGES < nk; ++j) {   // steady state
      half1();
      if constexpr (PROD) wait_vmcnt<AHEAD * PER>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      stage_role();
      next_slot();
      rd_lower();
      upper(std::true_type{});
#ifndef EVC_NO_INTERLEAVE
      pattern(std::integral_constant<int, MH * GN>{}, std::integral_constant<int, 2 * (ML + GN)>{}, std::integral_constant<int, PERX>{});
#endif
      end_of_step();
    }
    for (; j + 1 < nk; ++j) {             // last STAGES stages: no refills
      half1();
      if constexpr (PROD) wait_landed(nk - (j + 2));
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      next_slot();
      rd_lower();
      upper(std::true_type{});
      end_of_step();
    }
    half1();
    upper(std::false_type{});
    end_of_step();
  }
  };   // run
  if constexpr (PRODUCERS) {
    if (producer) run(std::true_type{});
    else run(std::false_type{});
  } else {
    run(std::true_type{});
  }
}
