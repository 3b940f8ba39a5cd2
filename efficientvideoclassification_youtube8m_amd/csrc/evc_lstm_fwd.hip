// Fused LSTM forward step, the L2 wavefront pair launches and their entry points.
#include "gemm_shared.h"

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
                                     // operands of a step whose low-order weight halves are contracted in fp8 (evc_lstm_layer_fwd_f16_fp8lo);
                                     // 3 (round 6) = 2 + the low-order half of h itself: rows of 4H bytes [f16(h_t) | e4m3(h_t 2^7) |
                                     // e4m3((h_t - f16(h_t)) 2^18)] - contracted against [lo(W) | hi(W)] weight rows (h_lo = 1 of the same entries)
};


// F16: the operands (x_t, h_{t-1}, W) are IEEE f16 and ONE v_mfma_f32_16x16x32_f16 product is issued per depth - the cost of
// the bf16 step with 8x smaller operand rounding; h_t leaves twice, as f16 (next step's / next layer's operand) and as bf16
// (what the BPTT products contract over).
// the accumulators start from bias (+ forget_bias 1.0): its loads fly under the loop's prologue, and the tail
// has no load left that hipcc could re-issue between the fragments' stores
template <class Cfg>
__device__ __forceinline__ void lstm_fwd_acc_bias(const LstmFwdParams& e, const int u0, f32x4 (&acc)[Cfg::MI][4][Cfg::NI]) {
  {
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
}

// gate tail of one tile: sigma / tanh, c' = c f + i j, h' = tanh(c') o, masking, the stores (transposed accumulators: lane = one row, 4 units)
template <class Cfg, bool SPLIT, bool F16, bool FP8>
__device__ __forceinline__ void lstm_fwd_epilogue(const GemmOperands& p, const LstmFwdParams& e, const int m0, const int u0,
                                                  f32x4 (&acc)[Cfg::MI][4][Cfg::NI]) {
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
    constexpr int TP = EVC_FWD_TAPE_POLICY;           // the write-once tapes of the backward pass (gate records, bf16 cell history) on their own policy
#pragma unroll
    for (int mi = 0; mi < Cfg::MI; ++mi) {
      const int m = m0 + tc.row0 + mi * 16;
      if (ln[mi] < 0) continue;
      const uint32_t hu = (uint32_t)m * (uint32_t)H + (uint32_t)u;                      // element index in an [M][H] slab
      const uint32_t su4 = ((uint32_t)rm[mi] * (uint32_t)e.ld_state + (uint32_t)u) * 4u;   // byte offset in c_state / h_state
      // byte offset of this lane's 4 units in hout: FP8 rows are [f16(h) (H halfwords) | e4m3 (H bytes)] = 3H bytes, wide f16 rows 2H halfwords
      const bool rows8 = FP8 && e.h_wide >= 2;        // (an FP8 launch with h_wide 0 - evc_lstm_layer_fwd_f16_dith, whose e4m3 stages are the input's only - writes plain f16 rows)
      const bool rows8lo = FP8 && e.h_wide == 3;      // ... + the e4m3 image of h's own low-order half: rows of 4H bytes
      const uint32_t rowb = (uint32_t)m * (uint32_t)((rows8lo ? 4 : 3) * H);
      const uint32_t hw2 = rows8 ? rowb + (uint32_t)u * 2u : (F16 && e.h_wide == 1) ? ((uint32_t)m * (uint32_t)(2 * H) + (uint32_t)u) * 2u : hu * 2u;
      const uint32_t h8o = rowb + (uint32_t)(2 * H) + (uint32_t)u;      // (FP8: the row's e4m3 part)
      const u32x2_t z2 = {0u, 0u};
      if (e.t >= ln[mi]) {          // dynamic_rnn: state copied through, zero output
        store8<SP>(e.hout, hw2, z2);
        if (rows8) store4<SP>(e.hout, h8o, 0u);
        else if (F16 && e.h_wide == 1) store8<SP>(e.hout, hw2 + (uint32_t)H * 2u, z2);
        if (rows8lo) store4<SP>(e.hout, h8o + (uint32_t)H, 0u);
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
      if (e.c_hist) store8<TP>(e.c_hist, hu * 2u, u32x2_t{pack_bf16x2(cn[0], cn[1]), pack_bf16x2(cn[2], cn[3])});
      if (e.t == ln[mi] - 1)
        store16<SP>(e.h_state, su4, u32x4_t{__float_as_uint(hn[0]), __float_as_uint(hn[1]), __float_as_uint(hn[2]), __float_as_uint(hn[3])});
      const u32x2_t hb = {pack_bf16x2(hn[0], hn[1]), pack_bf16x2(hn[2], hn[3])};
      if (F16) {
        const uint32_t p01 = pack_f16x2_hw(hn[0], hn[1]), p23 = pack_f16x2_hw(hn[2], hn[3]);
        store8<SP>(e.hout, hw2, u32x2_t{p01, p23});
        if (rows8) {                // e4m3(h * 2^7): the activation operand of the weights' low-order halves (|h| < 1: no saturation)
          int w8 = __builtin_amdgcn_cvt_pk_fp8_f32(hn[0] * 128.0f, hn[1] * 128.0f, 0, false);
          w8 = __builtin_amdgcn_cvt_pk_fp8_f32(hn[2] * 128.0f, hn[3] * 128.0f, w8, true);
          store4<SP>(e.hout, h8o, (uint32_t)w8);
          if (rows8lo) {            // e4m3((h - f16(h)) 2^18): |h - f16(h)| <= 2^-12, at most 64 - against e4m3(W 2^6), the same 2^-24 as 7 + 17
            const float l0 = (hn[0] - f16_to_f32((f16_t)(p01 & 0xffffu))) * 262144.0f, l1 = (hn[1] - f16_to_f32((f16_t)(p01 >> 16))) * 262144.0f;
            const float l2 = (hn[2] - f16_to_f32((f16_t)(p23 & 0xffffu))) * 262144.0f, l3 = (hn[3] - f16_to_f32((f16_t)(p23 >> 16))) * 262144.0f;
            int v8 = __builtin_amdgcn_cvt_pk_fp8_f32(l0, l1, 0, false);
            v8 = __builtin_amdgcn_cvt_pk_fp8_f32(l2, l3, v8, true);
            store4<SP>(e.hout, h8o + (uint32_t)H, (uint32_t)v8);
          }
        } else if (e.h_wide == 1) { // f16(h)/64: the operand of the weights' low-order halves (scaled by 64)
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
        store16<TP>(e.gates, hu * 8u, u32x4_t{rec[0].x, rec[0].y, rec[1].x, rec[1].y});
        store16<TP>(e.gates, hu * 8u + 16u, u32x4_t{rec[2].x, rec[2].y, rec[3].x, rec[3].y});
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

template <class Cfg, bool SPLIT = false, bool F16 = false, bool FP8 = false, bool XINT = false>
__device__ __forceinline__ void lstm_fwd_step_body(const GemmOperands& p, const LstmFwdParams& e, int tiles_m, int tiles_n, int bid) {
  static_assert(!XINT || FP8, "the integer-frame form rides on the f16 + e4m3 loop");
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
  lstm_fwd_acc_bias<Cfg>(e, u0, acc);
  // (SPLIT: the split-bf16 products hi.hi + hi.lo + lo.hi are a K-EXTENSION of the same loop - the caller hands A = [lo | hi] rows
  //  against B = [W_hi | W_lo] rows as segment 1 and A = hi against B2 = W_hi as segment 2, evc_lstm_layer_fwd_hp - so the loop
  //  itself is the plain one; only the epilogue differs: it writes h_t's wide [lo | hi] image for the next step.)
  // (XINT, round 6: e.bias holds (255/256) colsum(f16(Wx)) here - the accumulators start from the constant term of the dequantised frames - and the
  //  loop rescales them by the frame's factor and adds the true bias behind the x-part: LOOP_ROW_SCALE)
  run_mainloop<Cfg, 4, true, false, EVC_FWD_LOOP_MODE | (F16 ? LOOP_F16 : 0) | (FP8 ? LOOP_FP8_TAIL : 0) | (XINT ? LOOP_ROW_SCALE : 0)>(p, m0, u0, acc);   // transposed accumulators: lane = one row, 4 consecutive units
  EVC_STAMP(p.stamp_slot, 2);
  lstm_fwd_epilogue<Cfg, SPLIT, F16, FP8>(p, e, m0, u0, acc);
}

// Two tiles per workgroup (round 5; bf16, the 64-wide ring tiles): layer 0's step s and layer 1's step s-1 of a two-layer L1 level are
// independent, so one launch runs both - every workgroup computes tile (tm, tn) of step a, then the same tile of step b.  Once every wave
// has left the ring, the first stages of tile b are issued (gemm_mainloop_v3 PHASE 1) and land under tile a's gate tail (which needs no
// LDS); tile b's loop starts on them (PHASE 2).  Per pair of steps that is one ring fill and one kernel boundary less.
// (round 6, the "high" L1 level: tile a = layer 0 on f16 + e4m3 stages (FP8, XINT: integer frames), tile b = the dithered upper layer on plain f16 stages
//  (FP8B = false) - each tile's loop and gate tail in its own mode, the PHASE split as in bf16)
template <class Cfg, bool F16 = false, bool FP8 = false, bool XINT = false, bool FP8B = FP8>
__global__ __launch_bounds__(Cfg::NT) void lstm_fwd_walk2_kernel(GemmOperands pa, LstmFwdParams ea, int tiles_ma, GemmOperands pb, LstmFwdParams eb,
                                                                 int tiles_mb, int tiles_n) {
  static_assert(is_v3<Cfg>::value && Cfg::G == 4, "tile walk: the 64-wide ring tiles");
  static_assert((!FP8 && !FP8B) || F16, "the e4m3 tail rides behind f16 stages");
  static_assert(!XINT || FP8, "the integer-frame form rides on the f16 + e4m3 loop");
  constexpr int MODE_A = EVC_FWD_LOOP_MODE | (F16 ? LOOP_F16 : 0) | (FP8 ? LOOP_FP8_TAIL : 0) | (XINT ? LOOP_ROW_SCALE : 0);
  constexpr int MODE_B = EVC_FWD_LOOP_MODE | (F16 ? LOOP_F16 : 0) | (FP8B ? LOOP_FP8_TAIL : 0);
  const int tiles_m = tiles_ma > tiles_mb ? tiles_ma : tiles_mb;
  const int id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  int tm, tn;
  tile_of(id, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * Cfg::BM, u0 = tn * Cfg::BU;
  const bool has_a = tm < tiles_ma, has_b = tm < tiles_mb;          // workgroup-uniform
  f32x4 acc[Cfg::MI][4][Cfg::NI];
  if (has_a) {
    lstm_fwd_acc_bias<Cfg>(ea, u0, acc);
    gemm_mainloop_v3<Cfg, true, false, MODE_A, 0>(pa, m0, u0, lds_dyn, acc);
    if (has_b) {
      __syncthreads();                                              // every wave has read its last ring slot
      gemm_mainloop_v3<Cfg, true, false, MODE_B, 1>(pb, m0, u0, lds_dyn, acc);
    }
    lstm_fwd_epilogue<Cfg, false, F16, FP8>(pa, ea, m0, u0, acc);
    if (has_b) {
      __builtin_amdgcn_sched_barrier(0);                            // (keep tile b's address set-up out of tile a's tail: register pressure)
      asm volatile("" ::: "memory");
      lstm_fwd_acc_bias<Cfg>(eb, u0, acc);
      gemm_mainloop_v3<Cfg, true, false, MODE_B, 2>(pb, m0, u0, lds_dyn, acc);
      lstm_fwd_epilogue<Cfg, false, F16, FP8B>(pb, eb, m0, u0, acc);
    }
  } else if (has_b) {
    lstm_fwd_acc_bias<Cfg>(eb, u0, acc);
    gemm_mainloop_v3<Cfg, true, false, MODE_B, 0>(pb, m0, u0, lds_dyn, acc);
    lstm_fwd_epilogue<Cfg, false, F16, FP8B>(pb, eb, m0, u0, acc);
  }
}
#ifdef EVC_STAMPS
extern "C" int evc_debug_read_stamps(unsigned long long* out) {     // out: [8][512][8]
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(evc_stamps), sizeof(unsigned long long) * 8 * 512 * 8) == hipSuccess ? 0 : 1;
}
#endif

template <class Cfg, bool SPLIT = false, bool F16 = false, bool FP8 = false, bool XINT = false>
__global__ __launch_bounds__(Cfg::NT) void lstm_fwd_step_kernel(GemmOperands p, LstmFwdParams e, int tiles_m, int tiles_n) {
  lstm_fwd_step_body<Cfg, SPLIT, F16, FP8, XINT>(p, e, tiles_m, tiles_n, blockIdx.x);
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

template <class Cfg, bool SPLIT = false, bool F16 = false, bool FP8 = false, bool XINT = false>
static inline void launch_lstm_fwd(GemmOperands p, const LstmFwdParams& e, int k1, int k2, hipStream_t st) {
  p.nk1 = k1 / kdiv<Cfg>(); p.nk2 = k2 / kdiv<Cfg>();
#ifdef EVC_STAMPS
  p.stamp_slot = e.t & 7;
#endif
  const int tm = ceil_div(e.M, Cfg::BM), tn = ceil_div(e.H, Cfg::BU);
  launch_cfg<Cfg>(lstm_fwd_step_kernel<Cfg, SPLIT, F16, FP8, XINT>, tm * tn, st, p, e, tm, tn);
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

// one bf16 forward step on the tile the cost model picks for Mt rows
static inline void launch_fwd_step_bf16(const GemmOperands& p, const LstmFwdParams& e, int k1, int k2, int Mt, int H, hipStream_t st) {
  static const bool uneven224 = getenv("EVC_FWD_EVEN_224") == nullptr;      // the 224-row tile as 6 + 8 row fragments (A/B switch: 7 + 7; 58.3 -> 58.0 us per launch)
  static const bool fwd_v2 = getenv("EVC_FWD_V2_LOOP") != nullptr;          // A/B: the 32-wide K stages for the 160-256-row tiles
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

static int lstm_layer_fwd_impl(const evc_bf16* x, const evc_bf16* wT, const float* bias, const int32_t* len,
                               int T, int M, int Kin, int H, int hoist, float* zx_ws,
                               evc_bf16* hbuf, float* c_state, float* h_state, int64_t ld_state,
                               void* gates, evc_bf16* c_all, evc_bf16* hbuf_bf16,
                               const int32_t* row_map, const int32_t* rows_per_step, void* stream, int f16 = 0, int64_t ldx = 0, int h_wide = 0,
                               int64_t w_step_stride = 0);

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
                               const int32_t* row_map, const int32_t* rows_per_step, void* stream, int f16, int64_t ldx, int h_wide,
                               int64_t w_step_stride) {
  // w_step_stride (elements; 0 = one image): step t contracts the weight image at wT + t * w_step_stride (time-dithered f16 images,
  // evc_lstm_layer_fwd_f16_dith)
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
      p.B = wT + (long)t * w_step_stride;
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
    launch_fwd_step_bf16(p, e, k1, k2, Mt, H, st);
  }
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// ===========================================================================
// Two-layer L1 level (many rows, row plans), bf16: layer 0's step s and layer 1's step s-1 in ONE launch whose workgroups walk both tiles
// (lstm_fwd_walk2_kernel) - T + 1 launches instead of 2 T, one ring fill and one kernel boundary less per pair of steps; both layers contract
// [x_t | h_{t-1}] . W^T in one K walk (layer 1's x_t is layer 0's output slab).  Same arithmetic as two evc_lstm_layer_fwd calls: bit-identical
// results.  Launches whose tile is not one of the 64-wide ring tiles (224 / 240 / 256 rows) run as two separate launches.
// ===========================================================================
// has_a / has_b: which of the two tiles exist (a level's first launch has only layer 0's step, its last only layer 1's: the same kernel with one role,
// so that every launch of the level carries one kernel name - the per-kernel averages of a profile then cover exactly the launches bench.py times)
template <class Cfg, bool F16 = false, bool FP8 = false, bool XINT = false, bool FP8B = FP8>
static inline void launch_lstm_fwd_walk2(GemmOperands pa, const LstmFwdParams& ea, int k1a, int k2a, bool has_a, GemmOperands pb, const LstmFwdParams& eb,
                                         int k1b, int k2b, bool has_b, hipStream_t st) {
  pa.nk1 = k1a / 64; pa.nk2 = k2a / 64;
  pb.nk1 = k1b / 64; pb.nk2 = k2b / 64;
  const int H = has_a ? ea.H : eb.H;
  const int tma = has_a ? ceil_div(ea.M, Cfg::BM) : 0, tmb = has_b ? ceil_div(eb.M, Cfg::BM) : 0, tn = ceil_div(H, Cfg::BU);
  launch_cfg<Cfg>(lstm_fwd_walk2_kernel<Cfg, F16, FP8, XINT, FP8B>, (tma > tmb ? tma : tmb) * tn, st, pa, ea, tma, pb, eb, tmb, tn);
}

extern "C" int evc_lstm_level2_fwd(const evc_bf16* x, const evc_bf16* wT0, const float* bias0, const evc_bf16* wT1, const float* bias1,
                                   const int32_t* len, int T, int M, int Kin, int H, evc_bf16* hbuf0, evc_bf16* hbuf1,
                                   float* c_state0, float* h_state0, float* c_state1, float* h_state1, int64_t ld_state,
                                   void* gates0, evc_bf16* c_all0, void* gates1, evc_bf16* c_all1,
                                   const int32_t* row_map, const int32_t* rows_per_step, void* stream) {
  EVC_REQUIRE(T > 0 && M > 0 && H > 0 && Kin > 0 && Kin % 64 == 0 && H % 64 == 0, EVC_ERR_BAD_SHAPE,
              "evc_lstm_level2_fwd: bad shape T=%d M=%d Kin=%d H=%d (Kin, H multiples of 64)", T, M, Kin, H);
  EVC_REQUIRE(ring_operand_ok(M, Kin > H ? Kin : H) && ring_operand_ok(4L * H, (long)Kin + H) && ring_operand_ok(4L * H, 2L * H), EVC_ERR_BAD_SHAPE,
              "evc_lstm_level2_fwd: a time slab or a kernel spans 4 GiB or more (M=%d Kin=%d H=%d)", M, Kin, H);
  EVC_REQUIRE(fwd_tail_ok(M, H, ld_state), EVC_ERR_BAD_SHAPE, "LSTM forward: a gate-record / state slab spans 4 GiB or more (M=%d H=%d ld_state=%ld)", M, H, (long)ld_state);
  EVC_REQUIRE(x && wT0 && wT1 && bias0 && bias1 && len && hbuf0 && hbuf1 && c_state0 && h_state0 && c_state1 && h_state1, EVC_ERR_BAD_ARG,
              "evc_lstm_level2_fwd: NULL operand");
  EVC_REQUIRE(ld_state % 4 == 0 && ((uintptr_t)c_state0 % 16) == 0 && ((uintptr_t)h_state0 % 16) == 0 && ((uintptr_t)c_state1 % 16) == 0 &&
              ((uintptr_t)h_state1 % 16) == 0 && ((uintptr_t)bias0 % 16) == 0 && ((uintptr_t)bias1 % 16) == 0 &&
              ((uintptr_t)hbuf0 % 8) == 0 && ((uintptr_t)hbuf1 % 8) == 0, EVC_ERR_BAD_ALIGN,
              "evc_lstm_level2_fwd: state/bias/hbuf must allow 16-byte vector access");
  EVC_REQUIRE((gates0 == nullptr) == (c_all0 == nullptr) && (gates1 == nullptr) == (c_all1 == nullptr) && (gates0 == nullptr) == (gates1 == nullptr),
              EVC_ERR_BAD_ARG, "evc_lstm_level2_fwd: gates and c_all go together, for both layers");
  EVC_REQUIRE(!gates0 || (((uintptr_t)gates0 % 16) == 0 && ((uintptr_t)gates1 % 16) == 0 && ((uintptr_t)c_all0 % 8) == 0 && ((uintptr_t)c_all1 % 8) == 0),
              EVC_ERR_BAD_ALIGN, "evc_lstm_level2_fwd: gates must be 16-byte, c_all 8-byte aligned");
  if (rows_per_step)
    for (int t = 0; t < T; ++t)
      EVC_REQUIRE(rows_per_step[t] >= 0 && rows_per_step[t] <= M && (t == 0 || rows_per_step[t] <= rows_per_step[t - 1]), EVC_ERR_BAD_ARG,
                  "evc_lstm_level2_fwd: rows_per_step[%d]=%d must be non-increasing and within [0, M=%d]", t, rows_per_step[t], M);
  hipStream_t st = (hipStream_t)stream;
  EVC_CHECK_HIP(hipMemsetAsync(hbuf0, 0, (size_t)M * H * sizeof(bf16_t), st));       // h_{-1} = 0, both layers
  EVC_CHECK_HIP(hipMemsetAsync(hbuf1, 0, (size_t)M * H * sizeof(bf16_t), st));
  static const bool uneven224 = getenv("EVC_FWD_EVEN_224") == nullptr;
  auto step_args = [&](int layer, int t, GemmOperands& p, LstmFwdParams& e, int& k1, int& k2) {
    const int kin = layer ? H : Kin;
    const evc_bf16* xin = layer ? hbuf0 + (long)(t + 1) * M * H : x + (long)t * M * Kin;        // layer 1's x_t = layer 0's output slab t+1
    evc_bf16* hb = layer ? hbuf1 : hbuf0;
    p = GemmOperands();
    p.M = rows_per_step ? rows_per_step[t] : M; p.Nu = H; p.group_stride = H; p.ldb = (long)kin + H; p.nk1 = p.nk2 = 0;
    p.A1lo = p.A2lo = p.Blo = nullptr;
    p.A1 = xin; p.lda1 = kin; k1 = kin;
    p.A2 = hb + (long)t * M * H; p.lda2 = H; k2 = (t == 0) ? 0 : H;
    p.B = layer ? wT1 : wT0;
    e = LstmFwdParams();
    e.zx = nullptr; e.ldzx = 4L * H;
    e.bias = layer ? bias1 : bias0; e.len = len; e.t = t;
    e.c_state = layer ? c_state1 : c_state0; e.h_state = layer ? h_state1 : h_state0; e.ld_state = ld_state;
    e.hout = hb + (long)(t + 1) * M * H; e.h_wide = 0; e.hout_lo = nullptr;
    void* gt = layer ? gates1 : gates0;
    evc_bf16* ca = layer ? c_all1 : c_all0;
    e.gates = gt ? (uint2*)gt + (long)t * M * H : nullptr;
    e.c_hist = ca ? ca + (long)(t + 1) * M * H : nullptr;
    e.row_map = row_map;
    e.M = p.M; e.H = H;
  };
  for (int s = 0; s <= T; ++s) {
    GemmOperands pa, pb;
    LstmFwdParams ea, eb;
    int k1a = 0, k2a = 0, k1b = 0, k2b = 0;
    bool has_a = s < T, has_b = s >= 1;
    if (has_a) { step_args(0, s, pa, ea, k1a, k2a); has_a = ea.M > 0; }
    if (has_b) { step_args(1, s - 1, pb, eb, k1b, k2b); has_b = eb.M > 0; }
    if (has_a || has_b) {
      const int pick = pick_fwd_tile(has_b ? eb.M : ea.M, H);        // layer 1 runs the earlier step: at least as many rows as layer 0
      if (!has_a) { pa = pb; ea = eb; }                              // (the absent role's arguments are never read: tiles_m = 0)
      if (!has_b) { pb = pa; eb = ea; }
      if (pick == 2) { launch_lstm_fwd_walk2<CfgLstmV3_256>(pa, ea, k1a, k2a, has_a, pb, eb, k1b, k2b, has_b, st); continue; }
      if (pick == 10) { launch_lstm_fwd_walk2<CfgLstmV3_240>(pa, ea, k1a, k2a, has_a, pb, eb, k1b, k2b, has_b, st); continue; }
      if (pick == 3) {
        if (uneven224) launch_lstm_fwd_walk2<CfgLstmV3_224u>(pa, ea, k1a, k2a, has_a, pb, eb, k1b, k2b, has_b, st);
        else launch_lstm_fwd_walk2<CfgLstmV3_224>(pa, ea, k1a, k2a, has_a, pb, eb, k1b, k2b, has_b, st);
        continue;
      }
    }
    if (has_a) launch_fwd_step_bf16(pa, ea, k1a, k2a, ea.M, H, st);
    if (has_b) launch_fwd_step_bf16(pb, eb, k1b, k2b, eb.M, H, st);
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
// h_lo = 1 (round 6): the low-order half of h is corrected too - hbuf rows are 4H bytes [f16(h) | e4m3(h 2^7) | e4m3((h - f16(h)) 2^18)] and the
// h-part of wT8's rows is [lo(Wh) | hi(Wh)] (evc_cast_f32_to_fp8_lo with hi_tail): wT8 [4H][kx8 + 2H]; a layer above reads those rows with
// x8_off = 2H, kx8 = 2H against [lo(Wx) | hi(Wx)].
extern "C" int evc_lstm_layer_fwd_f16_fp8lo(const evc_f16* x, int64_t ldx, int kx16, int64_t x8_off, int kx8, const evc_f16* wT16,
                                            const uint8_t* wT8, int w8_scale_exp, int h_lo, const float* bias, const int32_t* len,
                                            int T, int M, int H, evc_f16* hbuf, evc_bf16* hbuf_bf16, float* c_state, float* h_state,
                                            int64_t ld_state, void* gates, evc_bf16* c_all, const int32_t* row_map,
                                            const int32_t* rows_per_step, const float* x_row_scale, const float* x_col_const, int b8_gap, void* stream) {
  EVC_REQUIRE(T > 0 && M > 0 && H > 0 && kx16 > 0 && kx8 > 0, EVC_ERR_BAD_SHAPE, "evc_lstm_layer_fwd_f16_fp8lo: bad shape");
  EVC_REQUIRE((x_row_scale == nullptr) == (x_col_const == nullptr) && b8_gap >= 0 && b8_gap % 128 == 0 && (x_row_scale || b8_gap == 0) &&
              (!x_col_const || ((uintptr_t)x_col_const % 16) == 0), EVC_ERR_BAD_ARG,
              "evc_lstm_layer_fwd_f16_fp8lo: x_row_scale and x_col_const go together (16-byte aligned), b8_gap=%d (%%128) only with them", b8_gap);
  EVC_REQUIRE(kx16 % 64 == 0 && H % 128 == 0 && kx8 % 128 == 0 && kx8 >= 384, EVC_ERR_BAD_SHAPE,
              "evc_lstm_layer_fwd_f16_fp8lo: kx16=%d (%%64), H=%d (%%128), kx8=%d (%%128, >= 384: the ring must be full of e4m3 stages at t = 0)", kx16, H, kx8);
  EVC_REQUIRE(x && wT16 && wT8 && hbuf && hbuf_bf16 && bias && len, EVC_ERR_BAD_ARG, "evc_lstm_layer_fwd_f16_fp8lo: NULL operand");
  EVC_REQUIRE(ldx % 8 == 0 && x8_off % 16 == 0 && ldx >= kx16 && ldx * 2 >= x8_off + kx8 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)wT16 % 16) == 0 &&
              ((uintptr_t)wT8 % 16) == 0 && ((uintptr_t)hbuf % 16) == 0 && ((uintptr_t)hbuf_bf16 % 8) == 0, EVC_ERR_BAD_ALIGN,
              "evc_lstm_layer_fwd_f16_fp8lo: ldx=%ld (%%8), x8_off=%ld (%%16), 16-byte aligned operands", (long)ldx, (long)x8_off);
  EVC_REQUIRE(w8_scale_exp >= 0 && w8_scale_exp <= 60, EVC_ERR_BAD_ARG, "evc_lstm_layer_fwd_f16_fp8lo: w8_scale_exp=%d", w8_scale_exp);
  EVC_REQUIRE(h_lo == 0 || h_lo == 1, EVC_ERR_BAD_ARG, "evc_lstm_layer_fwd_f16_fp8lo: h_lo=%d", h_lo);
  const long ldh = h_lo ? 2L * H : 3L * H / 2;      // halfwords per hbuf row
  const int kh8 = h_lo ? 2 * H : H;                 // e4m3 bytes of the h-part per row
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
    p.A4 = (const uint8_t*)(hprev + H); p.lda4 = ldh * 2; p.nk4 = t == 0 ? 0 : kh8 / 128;
    p.B8 = wT8; p.ldb8 = (long)kx8 + b8_gap + kh8; p.b8_gap = b8_gap;
    p.scale8_exp = -(7 + w8_scale_exp);
    if (x_row_scale) {          // integer frames: acc = acc * rs[row] + bias behind the x-part of the f16 stages; the accumulators start from x_col_const
      p.row_scale = x_row_scale + (long)t * M; p.col_add = bias; p.g2_add = 1.0f;
    }
    const int k1 = kx16, k2 = t == 0 ? 0 : H;
    LstmFwdParams e;
    e.zx = nullptr; e.ldzx = 0;
    e.bias = x_col_const ? x_col_const : bias; e.len = len; e.t = t;
    e.c_state = c_state; e.h_state = h_state; e.ld_state = ld_state;
    e.hout = hb + (long)(t + 1) * M * ldh; e.h_wide = h_lo ? 3 : 2;
    e.hout_lo = hbuf_bf16 + (long)(t + 1) * M * H;
    e.gates = gates ? (uint2*)gates + (long)t * M * H : nullptr;
    e.c_hist = c_all ? c_all + (long)(t + 1) * M * H : nullptr;
    e.row_map = row_map;
    e.M = Mt; e.H = H;
    if (x_row_scale) {
      switch (pick_fwd_tile_v3(Mt, H)) {
        case 0: launch_lstm_fwd<CfgLstmV3_256, false, true, true, true>(p, e, k1, k2, st); break;
        case 1: launch_lstm_fwd<CfgLstmV3_224, false, true, true, true>(p, e, k1, k2, st); break;
        case 2: launch_lstm_fwd<CfgLstmV3_192, false, true, true, true>(p, e, k1, k2, st); break;
        case 4: launch_lstm_fwd<CfgLstmV3_240, false, true, true, true>(p, e, k1, k2, st); break;
        default: launch_lstm_fwd<CfgLstmV3_160, false, true, true, true>(p, e, k1, k2, st); break;
      }
      continue;
    }
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

// "High" precision L1 layer on TIME-DITHERED f16 weight images (round 5; DESIGN.md 7 "dither"): step t contracts
//   z = [x16 | h16] . W16_t^T  (IEEE f16; W16_t = image t of evc_cast_f32_to_f16_dither at wT16 + t * w16_step_stride)
//       + 2^-scale8_exp  x8 . W8^T  (OCP e4m3 stages behind the f16 ones, kx8 > 0: the low-order half of the INPUT frames against an e4m3
//         image of Wx - the one activation term f16 does not cover)
// A weight's f16 rounding error is the same at every step of a chunk, so a recurrence integrates it coherently ("it is the
// weights"); image t rounds every element down or up such that over any run of steps the round-ups match the element's position between
// its two f16 neighbours - the errors cancel over the steps instead of adding up, and the weights' low-order halves need no stages of their
// own (evc_lstm_layer_fwd_f16_fp8lo: 26 + 16 e4m3 stages per step pair of a two-layer level; here 9 + 0).  x rows as in
// evc_lstm_layer_fwd_f16_fp8lo (kx16 halfwords at the row start, kx8 e4m3 bytes at byte offset x8_off; kx8 = 0: none, wT8 unused);
// wT8: rows of ldb8 bytes whose first kx8 bytes are the e4m3 operand; hbuf [T+1][M][H] PLAIN f16 rows (the next layer's x with kx8 = 0),
// hbuf_bf16 the bf16 copy.
extern "C" int evc_lstm_layer_fwd_f16_dith(const evc_f16* x, int64_t ldx, int kx16, int64_t x8_off, int kx8, const evc_f16* wT16,
                                           int64_t w16_step_stride, const uint8_t* wT8, int64_t ldb8, int scale8_exp, const float* bias,
                                           const int32_t* len, int T, int M, int H, evc_f16* hbuf, evc_bf16* hbuf_bf16, float* c_state,
                                           float* h_state, int64_t ld_state, void* gates, evc_bf16* c_all, const int32_t* row_map,
                                           const int32_t* rows_per_step, void* stream) {
  EVC_REQUIRE(T > 0 && M > 0 && H > 0 && kx16 > 0 && kx8 >= 0, EVC_ERR_BAD_SHAPE, "evc_lstm_layer_fwd_f16_dith: bad shape");
  EVC_REQUIRE(x && wT16 && hbuf && hbuf_bf16 && bias && len, EVC_ERR_BAD_ARG, "evc_lstm_layer_fwd_f16_dith: NULL operand");
  EVC_REQUIRE(w16_step_stride >= 0 && w16_step_stride % 8 == 0 && ((uintptr_t)wT16 % 16) == 0, EVC_ERR_BAD_ALIGN,
              "evc_lstm_layer_fwd_f16_dith: w16_step_stride=%ld (>= 0, %%8: every image 16-byte aligned)", (long)w16_step_stride);
  EVC_REQUIRE(w16_step_stride == 0 || w16_step_stride >= 4L * H * ((long)kx16 + H), EVC_ERR_BAD_ARG,
              "evc_lstm_layer_fwd_f16_dith: w16_step_stride=%ld is smaller than one [4H][kx16 + H] image", (long)w16_step_stride);
  if (kx8 == 0) {                // no e4m3 stages: the plain f16 layer on per-step images
    EVC_REQUIRE(ldx >= kx16 && ldx % 8 == 0 && ((uintptr_t)hbuf_bf16 % 8) == 0, EVC_ERR_BAD_ALIGN, "evc_lstm_layer_fwd_f16_dith: ldx=%ld (>= kx16=%d, %%8)", (long)ldx, kx16);
    return lstm_layer_fwd_impl((const evc_bf16*)x, (const evc_bf16*)wT16, bias, len, T, M, kx16, H, 0, nullptr, (evc_bf16*)hbuf, c_state, h_state,
                               ld_state, gates, c_all, hbuf_bf16, row_map, rows_per_step, stream, 1, ldx, 0, w16_step_stride);
  }
  EVC_REQUIRE(kx16 % 64 == 0 && H % 128 == 0 && kx8 % 128 == 0 && kx8 >= 384, EVC_ERR_BAD_SHAPE,
              "evc_lstm_layer_fwd_f16_dith: kx16=%d (%%64), H=%d (%%128), kx8=%d (%%128, >= 384: the ring must be full of e4m3 stages at t = 0)", kx16, H, kx8);
  EVC_REQUIRE(wT8 && ldb8 >= kx8 && ldb8 % 16 == 0 && ((uintptr_t)wT8 % 16) == 0, EVC_ERR_BAD_ARG, "evc_lstm_layer_fwd_f16_dith: wT8 rows of ldb8=%ld bytes (>= kx8, %%16)", (long)ldb8);
  EVC_REQUIRE(ldx % 8 == 0 && x8_off % 16 == 0 && ldx >= kx16 && ldx * 2 >= x8_off + kx8 && ((uintptr_t)x % 16) == 0 &&
              ((uintptr_t)hbuf % 16) == 0 && ((uintptr_t)hbuf_bf16 % 8) == 0, EVC_ERR_BAD_ALIGN,
              "evc_lstm_layer_fwd_f16_dith: ldx=%ld (%%8), x8_off=%ld (%%16), 16-byte aligned operands", (long)ldx, (long)x8_off);
  EVC_REQUIRE(scale8_exp >= 0 && scale8_exp <= 80, EVC_ERR_BAD_ARG, "evc_lstm_layer_fwd_f16_dith: scale8_exp=%d", scale8_exp);
  const long ldh = H;                                // halfwords per hbuf row (plain)
  EVC_REQUIRE(ring_operand_ok(M, ldx > ldh ? ldx : ldh) && ring_operand_ok(4L * H, (long)kx16 + H) && ring_operand_ok(4L * H, (ldb8 + 1) / 2), EVC_ERR_BAD_SHAPE,
              "evc_lstm_layer_fwd_f16_dith: a time slab or a weight image spans 4 GiB or more");
  EVC_REQUIRE(fwd_tail_ok(M, H, ld_state), EVC_ERR_BAD_SHAPE, "LSTM forward: a gate-record / state slab spans 4 GiB or more (M=%d H=%d ld_state=%ld)", M, H, (long)ld_state);
  EVC_REQUIRE(ld_state % 4 == 0 && ((uintptr_t)c_state % 16) == 0 && ((uintptr_t)h_state % 16) == 0 && ((uintptr_t)bias % 16) == 0, EVC_ERR_BAD_ALIGN,
              "evc_lstm_layer_fwd_f16_dith: state/bias must allow 16-byte vector access");
  EVC_REQUIRE((gates == nullptr) == (c_all == nullptr), EVC_ERR_BAD_ARG, "evc_lstm_layer_fwd_f16_dith: gates and c_all go together");
  EVC_REQUIRE(!gates || (((uintptr_t)gates % 16) == 0 && ((uintptr_t)c_all % 8) == 0), EVC_ERR_BAD_ALIGN,
              "evc_lstm_layer_fwd_f16_dith: gates must be 16-byte, c_all 8-byte aligned");
  if (rows_per_step)
    for (int t = 0; t < T; ++t)
      EVC_REQUIRE(rows_per_step[t] >= 0 && rows_per_step[t] <= M && (t == 0 || rows_per_step[t] <= rows_per_step[t - 1]), EVC_ERR_BAD_ARG,
                  "evc_lstm_layer_fwd_f16_dith: rows_per_step[%d]=%d must be non-increasing and within [0, M=%d]", t, rows_per_step[t], M);
  hipStream_t st = (hipStream_t)stream;
  const bf16_t* xb = (const bf16_t*)x;
  bf16_t* hb = (bf16_t*)hbuf;
  EVC_CHECK_HIP(hipMemsetAsync(hb, 0, (size_t)M * ldh * sizeof(bf16_t), st));            // h_{-1} = 0
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
    p.B = (const bf16_t*)wT16 + (long)t * w16_step_stride; p.ldb = (long)kx16 + H;
    p.A3 = (const uint8_t*)xt + x8_off; p.lda3 = ldx * 2; p.nk3 = kx8 / 128;
    p.A4 = p.A3; p.lda4 = p.lda3; p.nk4 = 0;                                            // no e4m3 stages of h
    p.B8 = wT8; p.ldb8 = ldb8;
    p.scale8_exp = -scale8_exp;
    const int k1 = kx16, k2 = t == 0 ? 0 : H;
    LstmFwdParams e;
    e.zx = nullptr; e.ldzx = 0;
    e.bias = bias; e.len = len; e.t = t;
    e.c_state = c_state; e.h_state = h_state; e.ld_state = ld_state;
    e.hout = hb + (long)(t + 1) * M * ldh; e.h_wide = 0;
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

// The two-layer L1 level of the "high" mode in T + 1 launches (round 6): layer 0 as evc_lstm_layer_fwd_f16_fp8lo (f16 stages + the e4m3 stages of its
// weights' / activations' low-order halves; integer frames with x_row_scale), layer 1 as evc_lstm_layer_fwd_f16_dith with kx8 = 0 (plain f16 stages on the
// time-dithered image of its step), layer 0's step s and layer 1's step s-1 per launch, every workgroup walking both tiles (lstm_fwd_walk2_kernel: the second
// tile's ring fill under the first tile's gate tail).  Same arithmetic as the two layer calls: bit-identical results.  Argument meaning as in those two
// entries; hbuf0 rows are layer 1's x rows (row stride 2H halfwords with h_lo, else 3H/2).  Launches whose tile is not 224 / 240 / 256 rows run separately.
extern "C" int evc_lstm_level2_fwd_high(const evc_f16* x, int64_t ldx, int kx16, int64_t x8_off, int kx8, const evc_f16* wT16_0, const uint8_t* wT8_0,
                                        int w8_scale_exp, int h_lo, const float* bias0, const float* x_row_scale, const float* x_col_const, int b8_gap,
                                        const evc_f16* wT16_1, int64_t w16_step_stride, const float* bias1, const int32_t* len, int T, int M, int H,
                                        evc_f16* hbuf0, evc_bf16* hbuf0_bf16, evc_f16* hbuf1, evc_bf16* hbuf1_bf16, float* c_state0, float* h_state0,
                                        float* c_state1, float* h_state1, int64_t ld_state, void* gates0, evc_bf16* c_all0, void* gates1, evc_bf16* c_all1,
                                        const int32_t* row_map, const int32_t* rows_per_step, void* stream) {
  EVC_REQUIRE(T > 0 && M > 0 && H > 0 && kx16 > 0 && kx8 > 0, EVC_ERR_BAD_SHAPE, "evc_lstm_level2_fwd_high: bad shape");
  EVC_REQUIRE((x_row_scale == nullptr) == (x_col_const == nullptr) && b8_gap >= 0 && b8_gap % 128 == 0 && (x_row_scale || b8_gap == 0) &&
              (!x_col_const || ((uintptr_t)x_col_const % 16) == 0), EVC_ERR_BAD_ARG,
              "evc_lstm_level2_fwd_high: x_row_scale and x_col_const go together (16-byte aligned), b8_gap=%d (%%128) only with them", b8_gap);
  EVC_REQUIRE(kx16 % 64 == 0 && H % 128 == 0 && kx8 % 128 == 0 && kx8 >= 384, EVC_ERR_BAD_SHAPE,
              "evc_lstm_level2_fwd_high: kx16=%d (%%64), H=%d (%%128), kx8=%d (%%128, >= 384)", kx16, H, kx8);
  EVC_REQUIRE(x && wT16_0 && wT8_0 && wT16_1 && hbuf0 && hbuf0_bf16 && hbuf1 && hbuf1_bf16 && bias0 && bias1 && len && c_state0 && h_state0 && c_state1 &&
              h_state1, EVC_ERR_BAD_ARG, "evc_lstm_level2_fwd_high: NULL operand");
  EVC_REQUIRE(ldx % 8 == 0 && x8_off % 16 == 0 && ldx >= kx16 && ldx * 2 >= x8_off + kx8 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)wT16_0 % 16) == 0 &&
              ((uintptr_t)wT8_0 % 16) == 0 && ((uintptr_t)wT16_1 % 16) == 0 && ((uintptr_t)hbuf0 % 16) == 0 && ((uintptr_t)hbuf1 % 16) == 0 &&
              ((uintptr_t)hbuf0_bf16 % 8) == 0 && ((uintptr_t)hbuf1_bf16 % 8) == 0, EVC_ERR_BAD_ALIGN,
              "evc_lstm_level2_fwd_high: ldx=%ld (%%8), x8_off=%ld (%%16), 16-byte aligned operands", (long)ldx, (long)x8_off);
  EVC_REQUIRE(w8_scale_exp >= 0 && w8_scale_exp <= 60 && (h_lo == 0 || h_lo == 1), EVC_ERR_BAD_ARG, "evc_lstm_level2_fwd_high: w8_scale_exp=%d h_lo=%d",
              w8_scale_exp, h_lo);
  EVC_REQUIRE(w16_step_stride >= 0 && w16_step_stride % 8 == 0 && (w16_step_stride == 0 || w16_step_stride >= 4L * H * 2 * H), EVC_ERR_BAD_ARG,
              "evc_lstm_level2_fwd_high: w16_step_stride=%ld (0 or at least one [4H][2H] image, %%8)", (long)w16_step_stride);
  const long ldh0 = h_lo ? 2L * H : 3L * H / 2;     // halfwords per hbuf0 row
  const int kh8 = h_lo ? 2 * H : H;                 // e4m3 bytes of layer 0's h-part per row
  EVC_REQUIRE(ring_operand_ok(M, ldx > ldh0 ? ldx : ldh0) && ring_operand_ok(4L * H, (long)kx16 + H) && ring_operand_ok(4L * H, 2L * H), EVC_ERR_BAD_SHAPE,
              "evc_lstm_level2_fwd_high: a time slab or a kernel spans 4 GiB or more");
  EVC_REQUIRE(fwd_tail_ok(M, H, ld_state), EVC_ERR_BAD_SHAPE, "LSTM forward: a gate-record / state slab spans 4 GiB or more (M=%d H=%d ld_state=%ld)", M, H, (long)ld_state);
  EVC_REQUIRE(ld_state % 4 == 0 && ((uintptr_t)c_state0 % 16) == 0 && ((uintptr_t)h_state0 % 16) == 0 && ((uintptr_t)c_state1 % 16) == 0 &&
              ((uintptr_t)h_state1 % 16) == 0 && ((uintptr_t)bias0 % 16) == 0 && ((uintptr_t)bias1 % 16) == 0, EVC_ERR_BAD_ALIGN,
              "evc_lstm_level2_fwd_high: state/bias must allow 16-byte vector access");
  EVC_REQUIRE((gates0 == nullptr) == (c_all0 == nullptr) && (gates1 == nullptr) == (c_all1 == nullptr) && (gates0 == nullptr) == (gates1 == nullptr),
              EVC_ERR_BAD_ARG, "evc_lstm_level2_fwd_high: gates and c_all go together, for both layers");
  EVC_REQUIRE(!gates0 || (((uintptr_t)gates0 % 16) == 0 && ((uintptr_t)gates1 % 16) == 0 && ((uintptr_t)c_all0 % 8) == 0 && ((uintptr_t)c_all1 % 8) == 0),
              EVC_ERR_BAD_ALIGN, "evc_lstm_level2_fwd_high: gates must be 16-byte, c_all 8-byte aligned");
  if (rows_per_step)
    for (int t = 0; t < T; ++t)
      EVC_REQUIRE(rows_per_step[t] >= 0 && rows_per_step[t] <= M && (t == 0 || rows_per_step[t] <= rows_per_step[t - 1]), EVC_ERR_BAD_ARG,
                  "evc_lstm_level2_fwd_high: rows_per_step[%d]=%d must be non-increasing and within [0, M=%d]", t, rows_per_step[t], M);
  hipStream_t st = (hipStream_t)stream;
  const bf16_t* xb = (const bf16_t*)x;
  bf16_t* hb0 = (bf16_t*)hbuf0;
  bf16_t* hb1 = (bf16_t*)hbuf1;
  EVC_CHECK_HIP(hipMemsetAsync(hb0, 0, (size_t)M * ldh0 * sizeof(bf16_t), st));          // h_{-1} = 0: every image of both layers' rows
  EVC_CHECK_HIP(hipMemsetAsync(hbuf0_bf16, 0, (size_t)M * H * sizeof(bf16_t), st));
  EVC_CHECK_HIP(hipMemsetAsync(hb1, 0, (size_t)M * H * sizeof(bf16_t), st));
  EVC_CHECK_HIP(hipMemsetAsync(hbuf1_bf16, 0, (size_t)M * H * sizeof(bf16_t), st));
  auto step0 = [&](int t, GemmOperands& p, LstmFwdParams& e, int& k1, int& k2) {          // layer 0: evc_lstm_layer_fwd_f16_fp8lo's step t
    p = GemmOperands();
    p.M = rows_per_step ? rows_per_step[t] : M; p.Nu = H; p.group_stride = H; p.nk1 = p.nk2 = 0;
    p.A1lo = p.A2lo = p.Blo = nullptr;
    const bf16_t* xt = xb + (long)t * M * ldx;
    const bf16_t* hprev = hb0 + (long)t * M * ldh0;
    p.A1 = xt; p.lda1 = ldx;
    p.A2 = hprev; p.lda2 = ldh0;
    p.B = (const bf16_t*)wT16_0; p.ldb = (long)kx16 + H;
    p.A3 = (const uint8_t*)xt + x8_off; p.lda3 = ldx * 2; p.nk3 = kx8 / 128;
    p.A4 = (const uint8_t*)(hprev + H); p.lda4 = ldh0 * 2; p.nk4 = t == 0 ? 0 : kh8 / 128;
    p.B8 = wT8_0; p.ldb8 = (long)kx8 + b8_gap + kh8; p.b8_gap = b8_gap;
    p.scale8_exp = -(7 + w8_scale_exp);
    if (x_row_scale) { p.row_scale = x_row_scale + (long)t * M; p.col_add = bias0; p.g2_add = 1.0f; }
    k1 = kx16; k2 = t == 0 ? 0 : H;
    e = LstmFwdParams();
    e.zx = nullptr; e.ldzx = 0;
    e.bias = x_col_const ? x_col_const : bias0; e.len = len; e.t = t;
    e.c_state = c_state0; e.h_state = h_state0; e.ld_state = ld_state;
    e.hout = hb0 + (long)(t + 1) * M * ldh0; e.h_wide = h_lo ? 3 : 2;
    e.hout_lo = hbuf0_bf16 + (long)(t + 1) * M * H;
    e.gates = gates0 ? (uint2*)gates0 + (long)t * M * H : nullptr;
    e.c_hist = c_all0 ? c_all0 + (long)(t + 1) * M * H : nullptr;
    e.row_map = row_map;
    e.M = p.M; e.H = H;
  };
  auto step1 = [&](int t, GemmOperands& p, LstmFwdParams& e, int& k1, int& k2) {          // layer 1: the plain f16 step on image t; x_t = layer 0's row slab t + 1
    p = GemmOperands();
    p.M = rows_per_step ? rows_per_step[t] : M; p.Nu = H; p.group_stride = H; p.ldb = 2L * H; p.nk1 = p.nk2 = 0;
    p.A1lo = p.A2lo = p.Blo = nullptr;
    p.A1 = hb0 + (long)(t + 1) * M * ldh0; p.lda1 = ldh0; k1 = H;
    p.A2 = hb1 + (long)t * M * H; p.lda2 = H; k2 = t == 0 ? 0 : H;
    p.B = (const bf16_t*)wT16_1 + (long)t * w16_step_stride;
    e = LstmFwdParams();
    e.zx = nullptr; e.ldzx = 4L * H;
    e.bias = bias1; e.len = len; e.t = t;
    e.c_state = c_state1; e.h_state = h_state1; e.ld_state = ld_state;
    e.hout = hb1 + (long)(t + 1) * M * H; e.h_wide = 0;
    e.hout_lo = hbuf1_bf16 + (long)(t + 1) * M * H;
    e.gates = gates1 ? (uint2*)gates1 + (long)t * M * H : nullptr;
    e.c_hist = c_all1 ? c_all1 + (long)(t + 1) * M * H : nullptr;
    e.row_map = row_map;
    e.M = p.M; e.H = H;
  };
  static const bool uneven224 = getenv("EVC_FWD_EVEN_224") == nullptr;
  for (int s = 0; s <= T; ++s) {
    GemmOperands pa, pb;
    LstmFwdParams ea, eb;
    int k1a = 0, k2a = 0, k1b = 0, k2b = 0;
    bool has_a = s < T, has_b = s >= 1;
    if (has_a) { step0(s, pa, ea, k1a, k2a); has_a = ea.M > 0; }
    if (has_b) { step1(s - 1, pb, eb, k1b, k2b); has_b = eb.M > 0; }
    if (!has_a && !has_b) continue;
    const int pick = pick_fwd_tile_v3(has_b ? eb.M : ea.M, H);       // layer 1 runs the earlier step: at least as many rows as layer 0
    if (pick == 0 || pick == 1 || pick == 4) {
      if (!has_a) { pa = pb; ea = eb; }                              // (the absent role's arguments are never read: tiles_m = 0)
      if (!has_b) { pb = pa; eb = ea; }
#define EVC_WALK2_HIGH(CFG)                                                                                                   \
      do {                                                                                                                    \
        if (x_row_scale) launch_lstm_fwd_walk2<CFG, true, true, true, false>(pa, ea, k1a, k2a, has_a, pb, eb, k1b, k2b, has_b, st);   \
        else launch_lstm_fwd_walk2<CFG, true, true, false, false>(pa, ea, k1a, k2a, has_a, pb, eb, k1b, k2b, has_b, st);              \
      } while (0)
      if (pick == 0) EVC_WALK2_HIGH(CfgLstmV3_256);
      else if (pick == 4) EVC_WALK2_HIGH(CfgLstmV3_240);
      else if (uneven224) EVC_WALK2_HIGH(CfgLstmV3_224u);
      else EVC_WALK2_HIGH(CfgLstmV3_224);
#undef EVC_WALK2_HIGH
      continue;
    }
    if (has_a) {
      if (x_row_scale) {
        if (pick == 2) launch_lstm_fwd<CfgLstmV3_192, false, true, true, true>(pa, ea, k1a, k2a, st);
        else launch_lstm_fwd<CfgLstmV3_160, false, true, true, true>(pa, ea, k1a, k2a, st);
      } else {
        if (pick == 2) launch_lstm_fwd<CfgLstmV3_192, false, true, true>(pa, ea, k1a, k2a, st);
        else launch_lstm_fwd<CfgLstmV3_160, false, true, true>(pa, ea, k1a, k2a, st);
      }
    }
    if (has_b) {
      if (pick == 2) launch_lstm_fwd<CfgLstmV3_192, false, true>(pb, eb, k1b, k2b, st);
      else launch_lstm_fwd<CfgLstmV3_160, false, true>(pb, eb, k1b, k2b, st);
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
// h_lo = 1 (round 6): the activations' low-order halves are corrected as well - h rows of 4H bytes [f16(h) | e4m3(h 2^7) | e4m3((h - f16(h)) 2^18)],
// wT0_8 [4H][2H] = [lo(Wh0) | hi(Wh0)], wT1_8 [4H][4H] = [lo(Wx1) | hi(Wx1) | lo(Wh1) | hi(Wh1)] (evc_cast_f32_to_fp8_lo with hi_tail): layer 1
// walks 32 f16 + 32 e4m3 stages.
extern "C" int evc_lstm_stack2_fwd_f16_fp8lo(const evc_f16* x, int x_segments, const evc_f16* wT0, const uint8_t* wT0_8, const float* bias0,
                                             const evc_f16* wT1, const uint8_t* wT1_8, int w8_scale_exp, int h_lo, const float* bias1,
                                             const int32_t* len, int T, int M, int Kin, int H, float* zx_ws,
                                             evc_f16* h0_rows, evc_f16* h1_rows, evc_bf16* hbuf0, evc_bf16* hbuf1,
                                             float* c_state0, float* h_state0, float* c_state1, float* h_state1, int64_t ld_state,
                                             void* gates0, evc_bf16* c_all0, void* gates1, evc_bf16* c_all1, void* stream) {
  EVC_REQUIRE(T > 0 && M > 0 && H >= 512 && Kin > 0 && H % 128 == 0 && Kin % 64 == 0, EVC_ERR_BAD_SHAPE,
              "evc_lstm_stack2_fwd_f16_fp8lo: bad shape T=%d M=%d Kin=%d (%%64) H=%d (%%128, >= 512)", T, M, Kin, H);
  EVC_REQUIRE(x && wT0 && wT0_8 && wT1 && wT1_8 && zx_ws && h0_rows && h1_rows && hbuf0 && hbuf1, EVC_ERR_BAD_ARG, "evc_lstm_stack2_fwd_f16_fp8lo: NULL operand");
  EVC_REQUIRE(x_segments >= 1 && x_segments <= 3 && w8_scale_exp >= 0 && w8_scale_exp <= 60, EVC_ERR_BAD_ARG,
              "evc_lstm_stack2_fwd_f16_fp8lo: x_segments=%d w8_scale_exp=%d", x_segments, w8_scale_exp);
  EVC_REQUIRE(h_lo == 0 || h_lo == 1, EVC_ERR_BAD_ARG, "evc_lstm_stack2_fwd_f16_fp8lo: h_lo=%d", h_lo);
  const long Kx = (long)x_segments * Kin;
  const long ldw0 = Kx + H, ldh = h_lo ? 2L * H : 3L * H / 2;
  const int kh8 = h_lo ? 2 * H : H;                 // e4m3 bytes per h row: [h8] or [h8 | h_lo8]
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
      pa.A3 = (const uint8_t*)(hprev + H); pa.lda3 = ldh * 2; pa.nk3 = (t == 0) ? 0 : kh8 / 128;
      pa.A4 = pa.A3; pa.lda4 = pa.lda3; pa.nk4 = 0;
      pa.B8 = wT0_8; pa.ldb8 = kh8; pa.scale8_exp = -(7 + w8_scale_exp);
      ea.zx = zx_ws + (long)t * M * 4 * H; ea.ldzx = 4L * H;
      ea.bias = bias0; ea.len = len; ea.t = t;
      ea.c_state = c_state0; ea.h_state = h_state0; ea.ld_state = ld_state;
      ea.hout = h0r + (long)(t + 1) * M * ldh; ea.h_wide = h_lo ? 3 : 2;
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
      pb.A3 = (const uint8_t*)(xin + H); pb.lda3 = ldh * 2; pb.nk3 = kh8 / 128;
      pb.A4 = (const uint8_t*)(hprev + H); pb.lda4 = ldh * 2; pb.nk4 = (t == 0) ? 0 : kh8 / 128;
      pb.B8 = wT1_8; pb.ldb8 = 2L * kh8; pb.scale8_exp = -(7 + w8_scale_exp);
      eb.zx = nullptr; eb.ldzx = 0;
      eb.bias = bias1; eb.len = len; eb.t = t;
      eb.c_state = c_state1; eb.h_state = h_state1; eb.ld_state = ld_state;
      eb.hout = h1r + (long)(t + 1) * M * ldh; eb.h_wide = h_lo ? 3 : 2;
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

