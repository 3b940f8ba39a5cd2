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
#pragma once
#include "gemm_core_v2.h"

template <int BM_, int G_, int BU_, int WR_, int WC_, int STAGES_ = 4>
struct TileCfg3 {
  static constexpr int BM = BM_, G = G_, BU = BU_, BN = G_ * BU_, WR = WR_, WC = WC_;
  static constexpr int WM = BM / WR, WU = BU / WC, MI = WM / 16, NI = WU / 16;
  static constexpr int NT = 64 * WR * WC;
  static constexpr int BK = 64, STAGES = STAGES_;
  static constexpr bool PIPE = true;
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr bool RAGGED = (BM / 8) % (NT / 64) != 0 || (BN / 8) % (NT / 64) != 0 || (BM / 8) % 4 != 0 || (BN / 8) % 4 != 0;
  static constexpr int DUMMY_OFF = STAGES * STAGE_BYTES;       // 1 KiB sink for the surplus lanes of a ragged last round
  static constexpr int LDS_BYTES = DUMMY_OFF + (RAGGED ? 1024 : 0);
  static_assert(WM % 16 == 0 && WU % 16 == 0 && BM % 8 == 0 && BN % 8 == 0, "wave tile must be a multiple of 16x16");
  static_assert(STAGES >= 2 && STAGES <= 6, "ring depth 2..6");
  static_assert(LDS_BYTES <= 160 * 1024, "exceeds the 160 KiB LDS of a CU");
};

template <class Cfg, bool SWAP = false, bool INIT = true, int MODE = EVC_LOOP_MODE_DEFAULT>
__device__ __forceinline__ void gemm_mainloop_v3(const GemmOperands& p, const int m0, const int u0, char* lds,
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
  const int nk = p.nk1 + p.nk2;   // in 64-wide stages
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
  uint32_t b_vo[BCH];
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
    b_vo[i] = (uint32_t)((((long)g * p.group_stride + gu) * p.ldb + lc8) * 2);
    b_dst[i] = live ? Cfg::A_BYTES + q * 1024 : -1;
  }
  const bf16_t* const b2 = p.B2 ? p.B2 - (long)p.nk1 * 64 : p.B;   // base such that b2 + ks*64 addresses the A2 segment's B columns
  int ks_issue = 0;                    // next stage to issue
  int slot_issue = 0, slot_read = 0;

  auto stage = [&]() {                 // branch-free (scalar selects only)
    const bool s1 = ks_issue < p.nk1;
    const char* ab = (const char*)(s1 ? p.A1 + (long)ks_issue * 64 : p.A2 + (long)(ks_issue - p.nk1) * 64);
    const uint32_t lda_b = (uint32_t)(s1 ? p.lda1 : p.lda2) * 2u;
    const char* b_base = (const char*)((s1 ? p.B : b2) + (long)ks_issue * 64);
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
    ++ks_issue;
    slot_issue = (slot_issue + 1 == Cfg::STAGES) ? 0 : slot_issue + 1;
  };

  // fragment read offsets within a stage: kb = 0 as computed, kb = 1 = the same offset with byte bit 6 flipped
  // (chunk (4 + fq) ^ s = (fq ^ s) ^ 4 for fq < 4)
  const int frow = lane & 15, fq = lane >> 4;
  const int fch = (fq ^ (frow & 7)) * 16;      // tile rows start on multiples of 16 -> row & 7 == frow & 7
  int a_rd[Cfg::MI], b_rd[Cfg::G][Cfg::NI];
#pragma unroll
  for (int mi = 0; mi < Cfg::MI; ++mi) a_rd[mi] = (wr * Cfg::WM + mi * 16 + frow) * 128 + fch;
#pragma unroll
  for (int g = 0; g < Cfg::G; ++g)
#pragma unroll
    for (int ni = 0; ni < Cfg::NI; ++ni)
      b_rd[g][ni] = Cfg::A_BYTES + (g * Cfg::BU + wc * Cfg::WU + ni * 16 + frow) * 128 + fch;

  auto read_half = [&](const int kb, bf16x8 (&af)[Cfg::MI], bf16x8 (&bfr)[Cfg::G][Cfg::NI]) {   // reads half kb of ring slot slot_read
    const char* sb = lds + slot_read * Cfg::STAGE_BYTES;
    const int x = kb << 6;
#pragma unroll
    for (int g = 0; g < Cfg::G; ++g)
#pragma unroll
      for (int ni = 0; ni < Cfg::NI; ++ni) bfr[g][ni] = *(const bf16x8*)(sb + (b_rd[g][ni] ^ x));
#pragma unroll
    for (int mi = 0; mi < Cfg::MI; ++mi) af[mi] = *(const bf16x8*)(sb + (a_rd[mi] ^ x));
  };
  auto next_slot = [&]() { slot_read = (slot_read + 1 == Cfg::STAGES) ? 0 : slot_read + 1; };
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
  auto end_of_step = [&]() {          // see gemm_core_v2.h: retire the LDS reads explicitly, nothing loop-carried for hipcc
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
  };
  constexpr int NMFMA = Cfg::MI * Cfg::G * Cfg::NI, NREAD = Cfg::MI + Cfg::G * Cfg::NI;
  constexpr int AHEAD = Cfg::STAGES - 2;     // stages that may stay in flight while the next one is awaited
  auto wait_landed = [&](int outstanding_stages) {   // wave-uniform small switch
    if (outstanding_stages >= 4) wait_vmcnt<4 * PER>();
    else if (outstanding_stages == 3) wait_vmcnt<3 * PER>();
    else if (outstanding_stages == 2) wait_vmcnt<2 * PER>();
    else if (outstanding_stages == 1) wait_vmcnt<PER>();
    else wait_vmcnt<0>();
  };

  auto run = [&](auto prod_tag) {     // one copy of the loop per role (LOOP_PRODUCER), each steady-state body branch-free
  constexpr bool PROD = decltype(prod_tag)::value;
  constexpr bool PRIO = (MODE & LOOP_NO_PRIO) == 0;
  constexpr int PERX = PROD ? PER : 0;
  auto stage_role = [&]() {
    if constexpr (PROD) stage();
  };
  auto interleave = [&](auto ndma_tag) {   // MFMAs with one LDS read / LDS-DMA between small groups of them
    constexpr int ndma = decltype(ndma_tag)::value;
    constexpr int per = NMFMA / (NREAD + ndma) > 0 ? NMFMA / (NREAD + ndma) : 1;
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
    __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - per * (NREAD + ndma), 0);
  };

  // ---- prologue: every slot of the ring in flight ----
#pragma unroll
  for (int i = 0; i < Cfg::STAGES; ++i)
    if (i < nk) stage_role();
  if constexpr (PROD) wait_landed(min(nk, Cfg::STAGES) - 1);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  bf16x8 afA[Cfg::MI], bfA[Cfg::G][Cfg::NI], afB[Cfg::MI], bfB[Cfg::G][Cfg::NI];
  read_half(0, afA, bfA);
  __builtin_amdgcn_s_waitcnt(0xC07F);   // enter the loop with no LDS read pending

  int j = 0;
  for (; j + Cfg::STAGES < nk; ++j) {   // steady state: stage j+STAGES exists, so every trip refills
    // first half: its partner fragments are in the stage being read - no wait, no barrier
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
    read_half(1, afB, bfB);
    mfma_all(afA, bfA);
#ifndef EVC_NO_INTERLEAVE
    interleave(std::integral_constant<int, 0>{});
#endif
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    end_of_step();
    // second half: stage j+1 must have landed; after the barrier every wave has read all of stage j -> refill its slot
    if constexpr (PROD) wait_vmcnt<AHEAD * PER>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
    stage_role();
    next_slot();
    read_half(0, afA, bfA);
    mfma_all(afB, bfB);
#ifndef EVC_NO_INTERLEAVE
    interleave(std::integral_constant<int, PERX>{});
#endif
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    end_of_step();
  }
  for (; j < nk; ++j) {                 // last STAGES stages: no refills
    read_half(1, afB, bfB);
    mfma_all(afA, bfA);
    end_of_step();
    if (j + 1 < nk) {
      if constexpr (PROD) wait_landed(nk - (j + 2));
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      next_slot();
      read_half(0, afA, bfA);
    }
    mfma_all(afB, bfB);
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
