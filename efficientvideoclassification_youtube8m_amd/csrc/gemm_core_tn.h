// "TN" main loop: acc[m][n] = sum_k A[k][m] * B[k][n], both operands stored with the
// CONTRACTION index as the row (A [K][lda], B [K][ldb]; m / n contiguous) - the shape of the
// weight-gradient products dW^T = dz^T . [x | h] where dz, x and h are all [rows = T*M][width].
// It removes the transposed copies an NT kernel would need.
//
// Same skeleton as gemm_core_v2.h (256x256 tile, 8 waves, K steps of 32, 5-deep LDS-DMA ring with
// counted vmcnt + raw barriers, register-double-buffered fragments).  Differences:
//  * LDS image per stage: [32 k rows][BM m] bf16 (512-byte rows) - one LDS-DMA wave-instruction
//    fills two full rows, so the global reads are 512-byte runs.
//  * MFMA fragments come from ds_read_b64_tr_b16 (hardware transpose read; semantics verified by
//    scripts/probes/tr_probe.hip): for the 16x16x32 A operand, lane l = 16g + 4q + p supplies the
//    address of (row k = 8g + 4j + q, columns mc + 4p ..) and receives column mc + (l & 15) of rows
//    8g + 4j .. +3; j = 0, 1 gives the 8 k values of the fragment.
//  * Bank conflicts: a 32-lane half reads 8 rows x 32 bytes; 512-byte rows put them all on the same
//    banks, so 16-byte chunk PAIRS are XOR-swizzled by h(row) = ((row>>3)&1)*4 + (row&3) (applied to
//    the DMA source address and to the read address): the 8 rows land in 8 distinct 32-byte slots
//    of the 256-byte bank row.
//  * M % 8 == 0, N % 8 == 0 (16-byte chunks along m / n), K % 32 == 0; chunks beyond M / N are
//    clamped in-bounds (results discarded by the epilogue).
#pragma once
#include "gemm_core_v2.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;
#ifndef EVC_TN_AUX_A
#define EVC_TN_AUX_A 0      // cache policy of the LDS-DMA loads (2 = nt).  Measured: nt on A +2.5 % in a replayed
                            // microbenchmark, -1.2 % in the training step (dz is fresh in the caches there); nt on B -5 %
#endif
#ifndef EVC_TN_AUX_B
#define EVC_TN_AUX_B 0
#endif

struct GemmOperandsT {
  const bf16_t* A; long lda;   // [K][lda], m contiguous
  const bf16_t* B; long ldb;   // [K][ldb], n contiguous
  int M, N;                    // valid columns of A / B
  int nk;                      // 32-wide K steps
  // B as two column segments (evc_gemm_tn2): product columns [0, N1) come from B, columns [N1, N) from B2 [K][ldb2];
  // N1 is a multiple of the tile width, so a workgroup's columns lie in one segment (gemm_tn_kernel picks it)
  const bf16_t* B2 = nullptr; long ldb2 = 0; int N1 = 0;
  int c_col2 = 0;              // C column where the second segment's columns start (>= N1: the segments need not be adjacent in C)
  // K as up to 16 SEGMENTS of live K steps (evc_gemm_tn2_rows, SEG loops only): the contraction index runs over time slabs of slab_rows rows of
  // which only a prefix is live (row plans: rows sorted by length) - segment i covers K steps [seg_start[i], seg_start[i] + seg_len[i]) of
  // the operands; nk then counts LIVE steps, and a workgroup's first step is step seg_off0 of segment seg0.
  int nseg = 0, seg0 = 0, seg_off0 = 0;
  unsigned short seg_start[16] = {0}, seg_len[16] = {0};
};

__device__ __forceinline__ int tn_h(int row) { return (((row >> 3) & 1) << 2) | (row & 3); }

// SWAP = true issues the MFMA with the B fragment first: the accumulator tile is transposed - lane l holds
// row (A column) m = l&15 and 4 consecutive columns n = (l>>4)*4 + reg (TileCoordsT) - so an epilogue that
// walks n fastest gets 16-byte vector accesses to row-major [M][N] arrays.
template <class Cfg, bool SWAP = false, int MODE = 0, bool SEG = false>
__device__ __forceinline__ void gemm_mainloop_tn(const GemmOperandsT& p, const int m0, const int n0, char* lds,
                                                 f32x4 (&acc)[Cfg::MI][1][Cfg::NI]) {
  static_assert(Cfg::G == 1 && Cfg::PIPE && !Cfg::RAGGED, "TN loop: plain tiles, pipelined, even staging");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / Cfg::WC, wc = wave % Cfg::WC;
  constexpr int ACPR = Cfg::BM / 8, BCPR = Cfg::BN / 8;          // 16-byte chunks per k row
  constexpr int AROWB = Cfg::BM * 2, BROWB = Cfg::BN * 2;

#pragma unroll
  for (int mi = 0; mi < Cfg::MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < Cfg::NI; ++ni) acc[mi][0][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = p.nk;
  if (nk == 0) return;

  // ---- staging: chunk c = piece*64 + lane -> k row c / CPR, physical chunk c % CPR; piece q belongs to wave q % NPW ----
  constexpr bool PRODUCERS = (MODE & LOOP_PRODUCER) != 0 && Cfg::NT == 512;   // waves 0..3 (one per SIMD) issue all the LDS-DMA
  constexpr int NPW = PRODUCERS ? 4 : Cfg::NT / 64;
  const bool producer = wave < NPW;
  constexpr int ACH = Cfg::BM / 16 / NPW, BCH = Cfg::BN / 16 / NPW, PER = ACH + BCH;
  static_assert(ACH * NPW * 16 == Cfg::BM && BCH * NPW * 16 == Cfg::BN, "pieces must divide over the staging waves");
  uint32_t a_vo[ACH], b_vo[BCH];      // per-lane byte offsets within a K step (the K walk is a scalar added per step: operands < 4 GiB)
#pragma unroll
  for (int i = 0; i < ACH; ++i) {
    const int c = ((wave % NPW) + i * NPW) * 64 + lane;
    const int row = c / ACPR, pc = c % ACPR;
    int col = m0 + ((pc ^ (tn_h(row) << 1)) << 3);
    col = col + 8 <= p.M ? col : p.M - 8;
    a_vo[i] = (uint32_t)(((long)row * p.lda + col) * 2);
  }
#pragma unroll
  for (int i = 0; i < BCH; ++i) {
    const int c = ((wave % NPW) + i * NPW) * 64 + lane;
    const int row = c / BCPR, pc = c % BCPR;
    int col = n0 + ((pc ^ (tn_h(row) << 1)) << 3);
    col = col + 8 <= p.N ? col : p.N - 8;
    b_vo[i] = (uint32_t)(((long)row * p.ldb + col) * 2);
  }
  const char* const a_base = (const char*)p.A;
  const char* const b_base = (const char*)p.B;
  const uint32_t a_step = (uint32_t)(64 * p.lda), b_step = (uint32_t)(64 * p.ldb);   // bytes per K step of 32 rows
  uint32_t a_k = 0, b_k = 0;
  int slot_issue = 0, slot_read = 0;
  // SEG: the K walk jumps over the dead rows at the end of every time slab.  The segment table (<= 16 entries, len << 16 | start) stays in
  // SCALAR registers and is indexed by a chain of scalar selects - only when a segment ends (a uniform, rarely taken branch; a handful of
  // SALU instructions per step otherwise).  (A table in one vector register read with v_readlane was tried first: it was spilled, and its
  // reload inside the loop brought the `s_waitcnt vmcnt(0)` of DESIGN.md 4.2b back.)
  uint32_t seg_e[16];
  int seg = 0, seg_left = 0;
  auto seg_entry = [&](int sidx) {
    uint32_t e = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) e = sidx == i ? seg_e[i] : e;
    return e;
  };
  if constexpr (SEG) {
#pragma unroll
    for (int i = 0; i < 16; ++i) seg_e[i] = ((uint32_t)p.seg_len[i] << 16) | (uint32_t)p.seg_start[i];
    seg = p.seg0;
    const uint32_t e = seg_entry(seg);
    seg_left = (int)(e >> 16) - p.seg_off0;
    a_k = ((e & 0xffffu) + (uint32_t)p.seg_off0) * a_step;
    b_k = ((e & 0xffffu) + (uint32_t)p.seg_off0) * b_step;
  }
  auto advance = [&]() {               // the K walk of the staging waves: one step on, or to the first step of the next segment
    if constexpr (SEG) {
      seg_left -= 1;
      if (__builtin_expect(seg_left == 0, 0)) {
        seg += 1;
        const uint32_t e = seg_entry(seg);
        a_k = (e & 0xffffu) * a_step;
        b_k = (e & 0xffffu) * b_step;
        seg_left = (int)(e >> 16);
      } else {
        a_k += a_step;
        b_k += b_step;
      }
    } else {
      a_k += a_step;
      b_k += b_step;
    }
  };

  auto stage = [&]() {
    char* sbase = lds + slot_issue * Cfg::STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < ACH; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_base + (a_vo[i] + a_k)),
                                       (__attribute__((address_space(3))) void*)(sbase + ((wave % NPW) + i * NPW) * 1024), 16, 0, EVC_TN_AUX_A);
#pragma unroll
    for (int i = 0; i < BCH; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_base + (b_vo[i] + b_k)),
                                       (__attribute__((address_space(3))) void*)(sbase + Cfg::A_BYTES + ((wave % NPW) + i * NPW) * 1024), 16, 0, EVC_TN_AUX_B);
    advance();
    slot_issue = (slot_issue + 1 == Cfg::STAGES) ? 0 : slot_issue + 1;
  };

  // ---- fragment (transpose) read addresses within a stage ----
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  int a_rd[Cfg::MI][2], b_rd[Cfg::NI][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = 8 * g + 4 * j + q, hs = tn_h(r) << 1;
#pragma unroll
    for (int mi = 0; mi < Cfg::MI; ++mi) {
      const int mc = wr * Cfg::WM + mi * 16;
      a_rd[mi][j] = r * AROWB + ((((mc >> 3) + (pp >> 1)) ^ hs) << 4) + ((pp & 1) << 3);
    }
#pragma unroll
    for (int ni = 0; ni < Cfg::NI; ++ni) {
      const int nc = wc * Cfg::WU + ni * 16;
      b_rd[ni][j] = Cfg::A_BYTES + r * BROWB + ((((nc >> 3) + (pp >> 1)) ^ hs) << 4) + ((pp & 1) << 3);
    }
  }
  // The transposing read as inline asm: through the builtin, hipcc's waitcnt pass sees an LDS read of unknown provenance
  // behind the LDS-DMA of the same K step and puts `s_waitcnt vmcnt(0)` in front of it - every K step then waited for the
  // DMA it had just issued (the whole ring's latency hiding gone; found in the ISA in round 2).  The asm has no memory
  // operand; its result is retired by the explicit lgkmcnt(0) of end_of_step() before the next step's MFMAs read it.
  auto tr = [&](const char* ptr) {
#ifdef EVC_TN_TR_BUILTIN
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)ptr);
#else
    const uint32_t a = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) const char*)ptr);
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(a));
    return v;
#endif
  };
  auto read_frags = [&](bf16x8 (&af)[Cfg::MI], bf16x8 (&bfr)[Cfg::NI]) {
    const char* sb = lds + slot_read * Cfg::STAGE_BYTES;
    slot_read = (slot_read + 1 == Cfg::STAGES) ? 0 : slot_read + 1;
#pragma unroll
    for (int ni = 0; ni < Cfg::NI; ++ni) {
      const s16x4 lo = tr(sb + b_rd[ni][0]), hi = tr(sb + b_rd[ni][1]);
      bfr[ni] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int mi = 0; mi < Cfg::MI; ++mi) {
      const s16x4 lo = tr(sb + a_rd[mi][0]), hi = tr(sb + a_rd[mi][1]);
      af[mi] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
  };
  auto mfma_all = [&](const bf16x8 (&af)[Cfg::MI], const bf16x8 (&bfr)[Cfg::NI]) {
#pragma unroll
    for (int mi = 0; mi < Cfg::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < Cfg::NI; ++ni)
        acc[mi][0][ni] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[mi][0][ni], 0, 0, 0)
                              : __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], bfr[ni], acc[mi][0][ni], 0, 0, 0);
  };
  auto end_of_step = [&]() {   // see gemm_core_v2.h: retire the LDS reads explicitly, nothing loop-carried for hipcc
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
  };
  constexpr int AHEAD = Cfg::STAGES - 2;
  auto wait_landed = [&](int outstanding_stages) {
    if (outstanding_stages >= AHEAD) wait_vmcnt<AHEAD * PER>();
    else if (outstanding_stages == 2) wait_vmcnt<2 * PER>();
    else if (outstanding_stages == 1) wait_vmcnt<PER>();
    else wait_vmcnt<0>();
  };
  static_assert(AHEAD == 3, "TN loop is written for the 5-deep ring");

  auto run = [&](auto prod_tag) {     // one copy of the loop per role (see gemm_mainloop_v2)
  constexpr bool PROD = decltype(prod_tag)::value;
  auto stage_role = [&]() {
    if constexpr (PROD) stage();
  };
#pragma unroll
  for (int i = 0; i < Cfg::STAGES - 1; ++i)
    if (i < nk) stage_role();

  // Steady-state step with the issue order written out: the transposing reads are inline asm (see tr()), which
  // sched_group_barrier cannot name, and left to itself hipcc bunches 22 of the 24 reads of every second step behind the last
  // MFMA, where nothing hides them.  One group = one MFMA with, in front of it, one LDS-DMA piece (producers, first PER groups)
  // or one fragment of the next step (two reads), spread evenly; sched_barrier(0) pins the groups.
  auto full_step = [&](const bf16x8 (&afc)[Cfg::MI], const bf16x8 (&bfc)[Cfg::NI], bf16x8 (&afn)[Cfg::MI], bf16x8 (&bfn)[Cfg::NI]) {
    if constexpr (PROD) wait_vmcnt<(AHEAD - 1) * PER>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr ((MODE & LOOP_NO_PRIO) == 0) __builtin_amdgcn_s_setprio(1);
#ifdef EVC_TN_UNORDERED
    stage_role();
    read_frags(afn, bfn);
    mfma_all(afc, bfc);
#else
    // EVC_TN_STAGGER (experiment): the waves that issue no LDS-DMA start their fragment reads SKIP MFMA groups into the step (the producers'
    // reads come behind their LDS-DMA pieces anyway), so that the SIMD partners' LDS reads do not run in step
#ifndef EVC_TN_STAGGER
#define EVC_TN_STAGGER 0
#endif
    constexpr int NM = Cfg::MI * Cfg::NI, ND = PROD ? PER : 0, NF = Cfg::MI + Cfg::NI, NIT = ND + NF;
    constexpr int SKIP = (PRODUCERS && !PROD && EVC_TN_STAGGER < NM - NIT) ? EVC_TN_STAGGER : 0;
    char* sbase = lds + slot_issue * Cfg::STAGE_BYTES;
    const char* sb = lds + slot_read * Cfg::STAGE_BYTES;
    slot_read = (slot_read + 1 == Cfg::STAGES) ? 0 : slot_read + 1;
#pragma unroll
    for (int gi = 0; gi < NM; ++gi) {
#pragma unroll
      for (int it = (gi < SKIP ? 0 : (gi - SKIP) * NIT / (NM - SKIP)); it < (gi < SKIP ? 0 : (gi - SKIP + 1) * NIT / (NM - SKIP)); ++it) {   // items of this group: LDS-DMA pieces first, then fragments
        if (it < ND) {
          if (it < ACH)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_base + (a_vo[it] + a_k)),
                                             (__attribute__((address_space(3))) void*)(sbase + ((wave % NPW) + it * NPW) * 1024), 16, 0, EVC_TN_AUX_A);
          else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_base + (b_vo[it - ACH] + b_k)),
                                             (__attribute__((address_space(3))) void*)(sbase + Cfg::A_BYTES + ((wave % NPW) + (it - ACH) * NPW) * 1024), 16, 0, EVC_TN_AUX_B);
        } else if (it - ND < Cfg::NI) {
          const int f = it - ND;
          const s16x4 lo = tr(sb + b_rd[f][0]), hi = tr(sb + b_rd[f][1]);
          bfn[f] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        } else {
          const int mi = it - ND - Cfg::NI;
          const s16x4 lo = tr(sb + a_rd[mi][0]), hi = tr(sb + a_rd[mi][1]);
          afn[mi] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      }
      const int mi = gi / Cfg::NI, ni = gi % Cfg::NI;
      acc[mi][0][ni] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfc[ni], afc[mi], acc[mi][0][ni], 0, 0, 0)
                            : __builtin_amdgcn_mfma_f32_16x16x32_bf16(afc[mi], bfc[ni], acc[mi][0][ni], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (PROD) {
      advance();
      slot_issue = (slot_issue + 1 == Cfg::STAGES) ? 0 : slot_issue + 1;
    }
#endif
    if constexpr ((MODE & LOOP_NO_PRIO) == 0) __builtin_amdgcn_s_setprio(0);
    end_of_step();
  };
  auto tail_step = [&](int kt, const bf16x8 (&afc)[Cfg::MI], const bf16x8 (&bfc)[Cfg::NI], bf16x8 (&afn)[Cfg::MI], bf16x8 (&bfn)[Cfg::NI]) {
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
  bf16x8 afA[Cfg::MI], bfA[Cfg::NI], afB[Cfg::MI], bfB[Cfg::NI];
  read_frags(afA, bfA);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  int kt = 0;
  for (; kt + Cfg::STAGES < nk; kt += 2) {
    full_step(afA, bfA, afB, bfB);
    full_step(afB, bfB, afA, bfA);
  }
  for (; kt + 1 < nk; kt += 2) {
    tail_step(kt, afA, bfA, afB, bfB);
    tail_step(kt + 1, afB, bfB, afA, bfA);
  }
  if (kt < nk) tail_step(kt, afA, bfA, afB, bfB);
  };   // run
  if constexpr (PRODUCERS) {
    if (producer) run(std::true_type{});
    else run(std::false_type{});
  } else {
    run(std::true_type{});
  }
}
