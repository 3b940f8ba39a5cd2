// v2 main loop for the large GEMM-shaped kernels: big workgroup tiles
// (up to 320 x 256), 8 waves, a 4-deep LDS ring of 32-wide K steps filled by
// LDS-DMA that stays in flight across barriers (counted vmcnt, raw s_barrier).
//
// Why (numbers in DESIGN.md "GEMM core v2"): a 128x128 tile needs ~64 B/clk/CU
// of L2->LDS traffic at full MFMA rate, above what an XCD's L2 delivers
// (~56 B/clk/CU), and the v1 loop drains its single prefetch (vmcnt(0)) at every
// barrier.  A 256x256 (320x256) tile needs 32 (28) B/clk/CU, and three K steps
// stay in flight while the fourth is consumed.
//
//  * acc[m][n] = sum_k A[m][k] * B[n][k], both operands K-contiguous (NT), bf16,
//    v_mfma_f32_16x16x32_bf16 (one K step = one MFMA depth).
//  * column groups (G) as in v1: the LSTM step keeps the four gate blocks of a
//    unit in one lane.
//  * LDS image per stage: rows of 64 B (32 bf16); 16-B chunk index is XOR-ed with
//    t[(row>>2)&3], t = {2,0,1,3}, which makes every ds_read_b128 lane group hit 16
//    distinct 16-B slots of the 256-B bank row (derivation in DESIGN.md); the
//    swizzle is applied to the per-lane SOURCE address of the LDS-DMA and to the
//    fragment reads (the DMA destination must stay lane-linear).
//  * Rows beyond M / units beyond Nu are clamped on load; K % 32 == 0.
#pragma once
#include "gemm_core.h"
#include <type_traits>

// Loop options (template parameter MODE of the main loops, a bit mask; measured per kernel):
//  LOOP_PRODUCER   only waves 0..3 (one per SIMD: waves i and i+4 share a SIMD) issue the LDS-DMA, twice as many pieces
//                  each; their SIMD partners run nothing but fragment reads and MFMAs (two copies of the loop, chosen once)
//  LOOP_DMA_FIRST  the scheduler is asked to place the LDS-DMA ahead of the fragment reads among the MFMAs of a K step
//  LOOP_NO_PRIO    no s_setprio 1 / 0 around every K step's MFMA cluster
//  LOOP_F16       the operands are IEEE f16: v_mfma_f32_16x16x32_f16 instead of _bf16 (staging and LDS image are the same)
//  LOOP_FP8_TAIL  (v3 loop only) the 16-bit stages are followed by stages of 128 e4m3 bytes per row on the MX-scaled MFMA
//                 (GemmOperands::A3 / A4 / B8); needs nk1 + nk2 >= 1 and nk3 + nk4 >= STAGES
constexpr int LOOP_PRODUCER = 1, LOOP_DMA_FIRST = 2, LOOP_NO_PRIO = 4, LOOP_F16 = 8, LOOP_FP8_TAIL = 16;
//  LOOP_PREFETCH  (v3 loop with LOOP_PRODUCER, experiment of round 4) two of the waves that issue no LDS-DMA touch, per trip, one dword of every 128-byte
//                 line of this workgroup's SHARE of the operand panels of a stage a few trips ahead (a 4-byte LDS-DMA into the sink: no destination
//                 register), so that the producers' refill of that stage finds its lines in the XCD's L2 instead of waiting for the Infinity Cache / HBM
constexpr int LOOP_PREFETCH = 32;
//  LOOP_B_NT      (v3 loop) the B operand's LDS-DMA loads are non-temporal: for products whose B rows are read by ONE workgroup each (the batch-row
//                 products, M <= 256: one row tile) - the weight matrix streams through the chip once and should not evict what the others re-read
constexpr int LOOP_B_NT = 64;
// v3 loop with LOOP_FP8_TAIL, transposed accumulators (SWAP): between the A1 and the A2 segment of the 16-bit stages every accumulator is
// rescaled per ROW and offset per COLUMN - acc = acc * row_scale[row] + col_add[column] (GemmOperands) - the integer-frame form of an L1 layer's
// step (round 6): the A1 segment contracts exact integers, the row scale is the frame's dequantise / l2-normalise factor.
constexpr int LOOP_ROW_SCALE = 128;
#ifndef EVC_LOOP_MODE_DEFAULT
#define EVC_LOOP_MODE_DEFAULT 0
#endif

template <int BM_, int G_, int BU_, int WR_, int WC_, int STAGES_ = 5, bool PIPE_ = true>
struct TileCfg2 {
  static constexpr int BM = BM_, G = G_, BU = BU_, BN = G_ * BU_, WR = WR_, WC = WC_;
  static constexpr int WM = BM / WR, WU = BU / WC, MI = WM / 16, NI = WU / 16;
  static constexpr int NT = 64 * WR * WC;
  static constexpr int BK = 32, STAGES = STAGES_;   // LDS ring depth: STAGES-2 K steps of LDS-DMA in flight
  static constexpr bool PIPE = PIPE_;               // double-buffer the MFMA fragments in registers
  static constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int ACH = (BM * 4 + NT - 1) / NT, BCH = (BN * 4 + NT - 1) / NT;  // LDS-DMA instructions per thread per stage
  static constexpr int PER = ACH + BCH;
  static constexpr bool RAGGED = (BM * 4) % NT != 0 || (BN * 4) % NT != 0;
  static constexpr int DUMMY_OFF = STAGES * STAGE_BYTES;       // 1 KiB sink for the surplus lanes of a ragged last round
  static constexpr int LDS_BYTES = DUMMY_OFF + (RAGGED ? 1024 : 0);
  static_assert(WM % 16 == 0 && WU % 16 == 0, "wave tile must be a multiple of 16x16");
  static_assert(BM % 16 == 0 && BN % 16 == 0 && (NT / 4) % 16 == 0, "staging rows per round must keep (row>>2)&3 fixed");
  static_assert(LDS_BYTES <= 160 * 1024, "exceeds the 160 KiB LDS of a CU");
};

__device__ __forceinline__ int swz64(int row) { return (0xD2 >> (2 * ((row >> 2) & 3))) & 3; }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {     // (the counter has 6 bits: a larger bound waits for 63, which is only stricter)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N > 63 ? 63 : N) : "memory");
}

template <class Cfg, bool SWAP = false, bool INIT = true, int MODE = EVC_LOOP_MODE_DEFAULT>
__device__ __forceinline__ void gemm_mainloop_v2(const GemmOperands& p, const int m0, const int u0, char* lds,
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

  const int nk = p.nk1 + p.nk2;   // in 32-wide K steps
  if (nk == 0) return;

  // ---- staging: chunk c = tid + i*NT -> tile row c>>2, physical chunk c&3 ----
  // Piece q (1 KiB = 16 tile rows) belongs to staging wave q % NPW (LOOP_PRODUCER: 4 staging waves, else all of them).
  const int lc8 = ((tid & 3) ^ swz64(tid >> 2)) * 8;
  constexpr bool PRODUCERS = (MODE & LOOP_PRODUCER) != 0 && Cfg::NT == 512;
  constexpr int NPW = PRODUCERS ? 4 : Cfg::NT / 64;
  const bool producer = wave < NPW;
  constexpr int ACH = (Cfg::BM / 16 + NPW - 1) / NPW, BCH = (Cfg::BN / 16 + NPW - 1) / NPW;
  constexpr int PER = ACH + BCH;
  static_assert(Cfg::RAGGED || (ACH * NPW * 16 == Cfg::BM && BCH * NPW * 16 == Cfg::BN), "surplus pieces need the dummy sink");
  // Every LDS-DMA is addressed as wave-uniform base + 32-bit per-lane BYTE offset (the launchers check that an operand spans
  // less than 4 GiB, rows < 2^24, row stride < 2^24 bytes): A's offset is one full-rate 24-bit multiply-add of the row index per
  // piece, B's is loop-invariant.  (With 64-bit pointers per piece the loop carried 6 VALU instructions per A piece, three
  // of them quarter-rate multiplies, on the waves whose issue slots the LDS-DMA already crowds.)
  int a_row[ACH];
  uint32_t b_vo[BCH];
  int a_dst[ACH], b_dst[BCH];   // wave-uniform LDS byte offsets within a stage (or the dummy sink)
#pragma unroll
  for (int i = 0; i < ACH; ++i) {
    const int c0 = ((wave % NPW) + i * NPW) * 64;           // first chunk of this wave-instruction
    const bool live = c0 < Cfg::BM * 4;                     // wave-uniform
    int r = (c0 + lane) >> 2;
    r = live ? r : 0;
    int gr = m0 + r;
    a_row[i] = gr < p.M ? gr : p.M - 1;
    a_dst[i] = live ? c0 * 16 : -1;
  }
#pragma unroll
  for (int i = 0; i < BCH; ++i) {
    const int c0 = ((wave % NPW) + i * NPW) * 64;
    const bool live = c0 < Cfg::BN * 4;
    int r = (c0 + lane) >> 2;
    r = live ? r : 0;
    const int g = r / Cfg::BU, u = r % Cfg::BU;
    int gu = u0 + u;
    gu = gu < p.Nu ? gu : p.Nu - 1;
    b_vo[i] = (uint32_t)((((long)g * p.group_stride + gu) * p.ldb + lc8) * 2);
    b_dst[i] = live ? Cfg::A_BYTES + c0 * 16 : -1;
  }
  const bf16_t* const b2 = p.B2 ? p.B2 - (long)p.nk1 * 32 : p.B;   // base such that b2 + kt*32 addresses the A2 segment's B columns
  int kt_issue = 0;   // next K step to stage
  int slot_issue = 0, slot_read = 0;   // ring slots (STAGES need not be a power of two)

  // Branch-free (scalar selects only) so the steady-state loop stays one basic block and
  // hipcc's waitcnt pass can count the loop-carried LDS reads exactly.
  auto stage = [&]() {
    const bool s1 = kt_issue < p.nk1;
    const char* ab = (const char*)(s1 ? p.A1 + (long)kt_issue * 32 : p.A2 + (long)(kt_issue - p.nk1) * 32);
    const uint32_t lda_b = (uint32_t)(s1 ? p.lda1 : p.lda2) * 2u;
    const char* b_base = (const char*)((s1 ? p.B : b2) + (long)kt_issue * 32);
    char* sbase = lds + slot_issue * Cfg::STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      char* dst = a_dst[i] >= 0 ? sbase + a_dst[i] : lds + Cfg::DUMMY_OFF;
      const uint32_t vo = __umul24((uint32_t)a_row[i], lda_b) + (uint32_t)(lc8 * 2);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ab + vo),
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
      char* dst = b_dst[i] >= 0 ? sbase + b_dst[i] : lds + Cfg::DUMMY_OFF;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_base + b_vo[i]),
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
    ++kt_issue;
    slot_issue = (slot_issue + 1 == Cfg::STAGES) ? 0 : slot_issue + 1;
  };

  // fragment read offsets within a stage
  const int frow = lane & 15, fq = lane >> 4;
  const int fch = (fq ^ swz64(frow)) * 16;     // tile rows start on multiples of 16 -> (row>>2)&3 == (frow>>2)&3
  int a_rd[Cfg::MI], b_rd[Cfg::G][Cfg::NI];
#pragma unroll
  for (int mi = 0; mi < Cfg::MI; ++mi) a_rd[mi] = (wr * Cfg::WM + mi * 16 + frow) * 64 + fch;
#pragma unroll
  for (int g = 0; g < Cfg::G; ++g)
#pragma unroll
    for (int ni = 0; ni < Cfg::NI; ++ni)
      b_rd[g][ni] = Cfg::A_BYTES + (g * Cfg::BU + wc * Cfg::WU + ni * 16 + frow) * 64 + fch;

  auto read_frags = [&](bf16x8 (&af)[Cfg::MI], bf16x8 (&bfr)[Cfg::G][Cfg::NI]) {   // reads the next ring slot
    const char* sb = lds + slot_read * Cfg::STAGE_BYTES;
    slot_read = (slot_read + 1 == Cfg::STAGES) ? 0 : slot_read + 1;
#pragma unroll
    for (int g = 0; g < Cfg::G; ++g)
#pragma unroll
      for (int ni = 0; ni < Cfg::NI; ++ni) bfr[g][ni] = *(const bf16x8*)(sb + b_rd[g][ni]);
#pragma unroll
    for (int mi = 0; mi < Cfg::MI; ++mi) af[mi] = *(const bf16x8*)(sb + a_rd[mi]);
  };
  auto mfma_all = [&](const bf16x8 (&af)[Cfg::MI], const bf16x8 (&bfr)[Cfg::G][Cfg::NI]) {
#pragma unroll
    for (int mi = 0; mi < Cfg::MI; ++mi)
#pragma unroll
      for (int g = 0; g < Cfg::G; ++g)
#pragma unroll
        for (int ni = 0; ni < Cfg::NI; ++ni)
          acc[mi][g][ni] = SWAP ? mfma16<(MODE & LOOP_F16) != 0>(bfr[g][ni], af[mi], acc[mi][g][ni])
                                : mfma16<(MODE & LOOP_F16) != 0>(af[mi], bfr[g][ni], acc[mi][g][ni]);
  };
  auto end_of_step = [&]() {
    // The next step's fragment reads were issued BEFORE these MFMAs and have landed long before
    // the 32 MFMAs retire; retiring them explicitly here (lgkmcnt(0) alone = 0xC07F) leaves hipcc's
    // waitcnt pass nothing loop-carried to be conservative about, so it does not put an
    // lgkmcnt(0) between the reads and the MFMAs of the following step.
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
  };
  // Issue order inside one step (T19): the in-order wave would otherwise issue 4 LDS-DMA + 12 LDS
  // reads in front of its 32 MFMAs while its SIMD partner does the same (all waves leave the
  // barrier together), leaving the matrix pipe idle.  Ask the scheduler for MFMA pairs with one
  // LDS read / one LDS-DMA between them.
  constexpr int NMFMA = Cfg::MI * Cfg::G * Cfg::NI, NREAD = Cfg::MI + Cfg::G * Cfg::NI;
  constexpr int AHEAD = Cfg::STAGES - 2;   // K steps of LDS-DMA left in flight at a wait (ring minus the slot
                                           // being read and the slot whose reads may still be pending)
  auto wait_landed = [&](int outstanding_stages) {   // wave-uniform small switch; only the tail leaves the first arm
    if (outstanding_stages >= AHEAD) wait_vmcnt<AHEAD * PER>();
    else if (outstanding_stages == 5) wait_vmcnt<5 * PER>();
    else if (outstanding_stages == 4) wait_vmcnt<4 * PER>();
    else if (outstanding_stages == 3) wait_vmcnt<3 * PER>();
    else if (outstanding_stages == 2) wait_vmcnt<2 * PER>();
    else if (outstanding_stages == 1) wait_vmcnt<PER>();
    else wait_vmcnt<0>();
  };
  static_assert(AHEAD >= 1 && AHEAD <= 6, "ring depth 3..8");

  // One copy of the loop per role (LOOP_PRODUCER: waves 0..3 stage, waves 4..7 do not - a wave-uniform choice made once, so
  // that each steady-state loop stays one branch-free basic block).
  auto run = [&](auto prod_tag) {
  constexpr bool PROD = decltype(prod_tag)::value;
  constexpr bool PRIO = (MODE & LOOP_NO_PRIO) == 0;
  constexpr int PERX = PROD ? PER : 0;               // LDS-DMA instructions this role issues per K step
  auto stage_role = [&]() {
    if constexpr (PROD) stage();
  };
  auto interleave_pipe = [&]() {
    constexpr int per = NMFMA / (NREAD + PERX) > 0 ? NMFMA / (NREAD + PERX) : 1;
    if constexpr ((MODE & LOOP_DMA_FIRST) != 0) {
#pragma unroll
      for (int i = 0; i < PERX; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, per, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read (LDS-DMA)
      }
    }
#pragma unroll
    for (int i = 0; i < NREAD; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, per, 0);   // MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // DS read
    }
    if constexpr ((MODE & LOOP_DMA_FIRST) == 0) {
#pragma unroll
      for (int i = 0; i < PERX; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, per, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read (LDS-DMA)
      }
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - per * (NREAD + PERX), 0);
  };
  // ---- prologue: STAGES-1 K steps in flight ----
#pragma unroll
  for (int i = 0; i < Cfg::STAGES - 1; ++i)
    if (i < nk) stage_role();

  if constexpr (Cfg::PIPE) {
    // Software-pipelined: iteration kt makes step kt+1 visible, refills the ring, starts the
    // fragment reads of kt+1 and runs the MFMAs of step kt from registers.
    // At the wait of iteration kt the steps kt+2 .. kt+AHEAD may stay in flight.
    auto full_step = [&](const bf16x8 (&afc)[Cfg::MI], const bf16x8 (&bfc)[Cfg::G][Cfg::NI],
                         bf16x8 (&afn)[Cfg::MI], bf16x8 (&bfn)[Cfg::G][Cfg::NI]) {
      if constexpr (PROD) wait_vmcnt<(AHEAD - 1) * PER>();
      __builtin_amdgcn_s_barrier();   // step kt+1 landed for every wave; every wave has consumed step kt-1's fragments
      asm volatile("" ::: "memory");
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
      stage_role();                   // refill step kt-1's slot with step kt+STAGES-1
      read_frags(afn, bfn);
      mfma_all(afc, bfc);
#ifndef EVC_NO_INTERLEAVE
      interleave_pipe();
#endif
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
      end_of_step();
    };
    auto tail_step = [&](int kt, const bf16x8 (&afc)[Cfg::MI], const bf16x8 (&bfc)[Cfg::G][Cfg::NI],
                         bf16x8 (&afn)[Cfg::MI], bf16x8 (&bfn)[Cfg::G][Cfg::NI]) {
      if (kt + 1 < nk) {
        if constexpr (PROD) wait_landed(min(nk, kt + Cfg::STAGES - 1) - (kt + 2));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + Cfg::STAGES - 1 < nk) stage_role();
        read_frags(afn, bfn);
      }
      mfma_all(afc, bfc);
      end_of_step();
    };
    if constexpr (PROD) wait_landed(min(nk, Cfg::STAGES - 1) - 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    bf16x8 afA[Cfg::MI], bfA[Cfg::G][Cfg::NI], afB[Cfg::MI], bfB[Cfg::G][Cfg::NI];
    read_frags(afA, bfA);
    __builtin_amdgcn_s_waitcnt(0xC07F);   // enter the loop with no LDS read pending (see mfma_all)
    int kt = 0;
    for (; kt + Cfg::STAGES < nk; kt += 2) {   // steady state: branch-free body, two steps per trip (static register sets)
      full_step(afA, bfA, afB, bfB);
      full_step(afB, bfB, afA, bfA);
    }
    for (; kt + 1 < nk; kt += 2) {
      tail_step(kt, afA, bfA, afB, bfB);
      tail_step(kt + 1, afB, bfB, afA, bfA);
    }
    if (kt < nk) tail_step(kt, afA, bfA, afB, bfB);
  } else {
    // Un-pipelined fragments (taller tiles that leave no registers for a second set):
    // iteration kt waits for step kt, refills the slot read in iteration kt-1, reads, multiplies.
    // Steps kt+1 .. kt+STAGES-2 may stay in flight at the wait.
    bf16x8 af[Cfg::MI], bfr[Cfg::G][Cfg::NI];
    int kt = 0;
    for (; kt + Cfg::STAGES - 1 < nk; ++kt) {
      if constexpr (PROD) wait_vmcnt<(Cfg::STAGES - 2) * PER>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
      read_frags(af, bfr);
      stage_role();
      mfma_all(af, bfr);
#ifndef EVC_NO_INTERLEAVE
      // B fragments + the first A fragment up front, then one MFMA row per further A read, LDS-DMA last
      __builtin_amdgcn_sched_group_barrier(0x100, Cfg::G * Cfg::NI + 1, 0);
#pragma unroll
      for (int i = 0; i + 1 < Cfg::MI; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, Cfg::G * Cfg::NI, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (i < PERX) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, Cfg::G * Cfg::NI, 0);
#endif
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
      end_of_step();
    }
    for (; kt < nk; ++kt) {
      if constexpr (PROD) wait_landed(nk - 1 - kt);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      read_frags(af, bfr);
      mfma_all(af, bfr);
      end_of_step();
    }
  }
  };   // run
  if constexpr (PRODUCERS) {
    if (producer) run(std::true_type{});
    else run(std::false_type{});
  } else {
    run(std::true_type{});
  }
}
