// Fused BPTT step kernels and their entry points.
#include "gemm_shared.h"

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
#ifndef EVC_BWD_TAPE_NT
#define EVC_BWD_TAPE_NT 0
#endif
#ifndef EVC_BWD_TAIL_PRE
#define EVC_BWD_TAIL_PRE 8          // row slots (of BM / 8 per thread) whose tape / state loads are issued BEFORE the main loop (round 6); 0 = every load in the tail (rounds 3-5)
#endif
// What a thread loads for N of its rows: the running dc (or the state gradient dS at a row's last step), the gate records, the cell history before and
// after the step, the gradient arriving from the layer above.  None of it depends on this launch's product.
template <int N>
struct BwdTailRows {
  unsigned long long dcv[N], dhs[N];      // float2 bits (ONE 64-bit value each: as two floats hipcc assigns the halves to unrelated registers and moves - waits - after the load)
  uint4 grec[N];
  uint32_t cn[N], co[N], dha[N];
};
// row length of slot p (wave-uniform, a scalar load from the constant address space; -1: row outside the launch or units outside H)
__device__ __forceinline__ int bwd_tail_len(const LstmBwdParams& e, int m) {
  return m < e.M ? ((const __attribute__((address_space(4))) int*)e.len)[m] : -1;
}
template <int N>
__device__ __forceinline__ void lstm_bwd_tail_load(const LstmBwdParams& e, int m0, int u, bool u_in, int wave, int p0, BwdTailRows<N>& r) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int m = m0 + (p0 + i) * 8 + wave;                    // wave-uniform
    const int ln = u_in ? bwd_tail_len(e, m) : -1;
    r.dcv[i] = r.dhs[i] = 0ull;
    r.grec[i] = make_uint4(0u, 0u, 0u, 0u);
    r.cn[i] = r.co[i] = r.dha[i] = 0u;
    if (e.t < ln) {
      const long hu = (long)m * e.H + u;
      // (ONE load instruction for dc whichever buffer it comes from - the state gradient at a row's last step, the running dc otherwise: two loads merged
      //  behind a branch cost a register move, and with it a wait for the load, right here in front of the main loop)
      const bool last = e.t == ln - 1;
      long su = 0;
      if (last) su = (long)(e.row_map ? ((const __attribute__((address_space(4))) int*)e.row_map)[m] : m) * e.ld_dS + u;
      if (e.dc_bf16 && !last) {
        const uint32_t d = *(const uint32_t*)((const bf16_t*)e.dc_ws + hu);
        r.dcv[i] = (unsigned long long)(d << 16) | ((unsigned long long)(d & 0xffff0000u) << 32);
      } else {
        r.dcv[i] = *(const unsigned long long*)(last ? e.dS_c + su : e.dc_ws + hu);
      }
      if (last) r.dhs[i] = *(const unsigned long long*)(e.dS_h + su);
#if EVC_BWD_TAPE_NT      // (A/B: the tape - gate records, cell history, the gradient from the layer above - is read ONCE, milliseconds after it was written: non-temporal loads)
      if (e.dh_above) r.dha[i] = __builtin_nontemporal_load((const uint32_t*)(e.dh_above + hu));
      { const u32x4_t gq = __builtin_nontemporal_load((const u32x4_t*)(e.gates + hu)); r.grec[i] = make_uint4(gq[0], gq[1], gq[2], gq[3]); }
      r.cn[i] = __builtin_nontemporal_load((const uint32_t*)(e.c_new + hu));
      if (e.c_old) r.co[i] = __builtin_nontemporal_load((const uint32_t*)(e.c_old + hu));
#else
      if (e.dh_above) r.dha[i] = *(const uint32_t*)(e.dh_above + hu);
      r.grec[i] = *(const uint4*)(e.gates + hu);
      r.cn[i] = *(const uint32_t*)(e.c_new + hu);
      if (e.c_old) r.co[i] = *(const uint32_t*)(e.c_old + hu);
#endif
    }
  }
}
// gate derivatives of N loaded rows against their dh (in LDS), then their stores
template <int RS, int N>
__device__ __forceinline__ void lstm_bwd_tail_finish(const LstmBwdParams& e, int m0, int u, bool u_in, int lane, int wave, int p0, const BwdTailRows<N>& r,
                                                     const char* lds, float (&bs)[2][4]) {
  // compute phase, then store phase: with the stores of row i between the computations of rows i and i+1 hipcc put
  // `s_waitcnt vmcnt(0)` in front of every row (it cannot count across the per-row branches), i.e. every row waited for the
  // store acknowledgements of the row before
  float2 dcn[N];
  uint4 dzr[N];
  int what[N];                                                 // 0: nothing, 1: zero dz (inactive row), 2: dc + dz
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int rl = (p0 + i) * 8 + wave;
    const int ln = u_in ? bwd_tail_len(e, m0 + rl) : -1;
    what[i] = ln < 0 ? 0 : (e.t >= ln ? 1 : 2);
    dcn[i] = make_float2(0.f, 0.f);
    dzr[i] = make_uint4(0u, 0u, 0u, 0u);
    if (what[i] != 2) continue;
    const float2 dhv = *(const float2*)(lds + rl * RS + lane * 8);
    float dh[2] = {dhv.x, dhv.y};
    if (e.t == ln - 1) {     // nothing flows back through the recurrent product from the (inactive) later steps
      const float sx = __uint_as_float((uint32_t)r.dhs[i]), sy = __uint_as_float((uint32_t)(r.dhs[i] >> 32));
      if (e.fused_above) { dh[0] += sx; dh[1] += sy; }
      else { dh[0] = sx; dh[1] = sy; }
    }
    if (e.dh_above) { dh[0] += __uint_as_float(r.dha[i] << 16); dh[1] += __uint_as_float(r.dha[i] & 0xffff0000u); }
    const float dci[2] = {__uint_as_float((uint32_t)r.dcv[i]), __uint_as_float((uint32_t)(r.dcv[i] >> 32))};
    const uint2 recs[2] = {make_uint2(r.grec[i].x, r.grec[i].y), make_uint2(r.grec[i].z, r.grec[i].w)};
    const float cna[2] = {__uint_as_float(r.cn[i] << 16), __uint_as_float(r.cn[i] & 0xffff0000u)};
    const float coa[2] = {__uint_as_float(r.co[i] << 16), __uint_as_float(r.co[i] & 0xffff0000u)};
    float dcv2[2];
    uint2 dz2[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float gi = __uint_as_float(recs[q].x << 16), gj = __uint_as_float(recs[q].x & 0xffff0000u);
      const float gf = __uint_as_float(recs[q].y << 16), go = __uint_as_float(recs[q].y & 0xffff0000u);
      const float tcv = tanhf_(cna[q]);
      const float dc = dci[q] + dh[q] * go * (1.f - tcv * tcv);
      dcv2[q] = dc * gf;
      const float z0 = dc * gj * gi * (1.f - gi), z1 = dc * gi * (1.f - gj * gj);
      const float z2 = dc * coa[q] * gf * (1.f - gf), z3 = dh[q] * tcv * go * (1.f - go);
      bs[q][0] += z0; bs[q][1] += z1; bs[q][2] += z2; bs[q][3] += z3;
      dz2[q] = make_uint2(pack_bf16x2(z0, z1), pack_bf16x2(z2, z3));
    }
    dcn[i] = make_float2(dcv2[0], dcv2[1]);
    dzr[i] = make_uint4(dz2[0].x, dz2[0].y, dz2[1].x, dz2[1].y);
  }
  // every load of the group has been consumed above; saying so (vmcnt(0), encoded 0x0F70) lets the stores below issue back
  // to back - across the per-row branches hipcc otherwise keeps some load destinations "pending" and waits before each row
  __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (what[i] == 0) continue;
    const long hu = (long)(m0 + (p0 + i) * 8 + wave) * e.H + u;
    if (what[i] == 2) {
      if (e.dc_bf16) *(uint32_t*)((bf16_t*)e.dc_ws + hu) = pack_bf16x2(dcn[i].x, dcn[i].y);
      else *(float2*)(e.dc_ws + hu) = dcn[i];
    }
    *(uint4*)(e.dz4 + hu) = dzr[i];                            // zeros for an inactive row: state passes through, no gate gradient
  }
}

// PRE row slots arrive preloaded (lstm_bwd_step_body issues their loads before the main loop: they land under the product instead of at its end - 4 groups
// of 4 rows used to pay a memory round trip each, one after the other, with nothing else on the CU); the next group's loads are issued before the preloaded
// rows are finished.
template <class Cfg, int PRE>
__device__ __forceinline__ void lstm_bwd_tail_rowmajor(f32x4 (&acc)[Cfg::MI][1][Cfg::NI], const LstmBwdParams& e, int m0, int u0, char* lds,
                                                       const BwdTailRows<(PRE > 0 ? PRE : 1)>& pre) {
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
  constexpr int SLOTS = Cfg::BM / 8;
  constexpr int GROUP = PRE > 0 ? 8 : 4;                      // rows per thread in flight in the tail itself (rounds 3-5: 2: 67.0 us per step, 4: 65.1)
  static_assert(PRE % 4 == 0 && PRE <= SLOTS && (SLOTS - PRE) % GROUP == 0, "tile rows must divide into 8 waves x (PRE + k GROUP)");
  if constexpr (PRE > 0) {
    if constexpr (PRE < SLOTS) {
      BwdTailRows<GROUP> nb;
      lstm_bwd_tail_load<GROUP>(e, m0, u, u_in, wave, PRE, nb);
      lstm_bwd_tail_finish<RS, PRE>(e, m0, u, u_in, lane, wave, 0, pre, lds, bs);
#pragma unroll
      for (int p0 = PRE; p0 < SLOTS; p0 += GROUP) {
        lstm_bwd_tail_finish<RS, GROUP>(e, m0, u, u_in, lane, wave, p0, nb, lds, bs);
        if (p0 + GROUP < SLOTS) lstm_bwd_tail_load<GROUP>(e, m0, u, u_in, wave, p0 + GROUP, nb);
      }
    } else {
      lstm_bwd_tail_finish<RS, PRE>(e, m0, u, u_in, lane, wave, 0, pre, lds, bs);
    }
  } else {
#pragma unroll 1
    for (int p0 = 0; p0 < SLOTS; p0 += GROUP) {
      BwdTailRows<GROUP> cur;
      lstm_bwd_tail_load<GROUP>(e, m0, u, u_in, wave, p0, cur);
      lstm_bwd_tail_finish<RS, GROUP>(e, m0, u, u_in, lane, wave, p0, cur, lds, bs);
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
#if !defined(EVC_ABLATE_BWD_EPI) && !defined(EVC_BWD_TAIL_FRAGMENTS)
  constexpr bool ROWMAJOR = is_v2<Cfg>::value && Cfg::BU == 128 && Cfg::NT == 512 && Cfg::BM % 32 == 0 && BATCH_LOADS;
#else
  constexpr bool ROWMAJOR = false;
#endif
  // the row-major tail's first PRE row slots: loads issued here, consumed after the product (older than every ring load: the ring's counted waits hold)
  constexpr int PRE = (ROWMAJOR && (Cfg::BM / 8 - EVC_BWD_TAIL_PRE) % 8 == 0 && EVC_BWD_TAIL_PRE <= Cfg::BM / 8) ? EVC_BWD_TAIL_PRE : 0;
  BwdTailRows<(PRE > 0 ? PRE : 1)> pre;
  if constexpr (PRE > 0) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int u = u0 + lane * 2;
    lstm_bwd_tail_load<PRE>(e, m0, u, u < e.H, wave, 0, pre);
  }
#ifdef EVC_ABLATE_BWD_MAIN     // debug build: epilogue only
#pragma unroll
  for (int mi = 0; mi < Cfg::MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < Cfg::NI; ++ni) acc[mi][0][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#else
  run_mainloop<Cfg, 1, true, true, EVC_BWD_LOOP_MODE>(p, m0, u0, acc);     // transposed accumulators: lane = one row, 4 consecutive units
#endif
  if constexpr (ROWMAJOR) {
    lstm_bwd_tail_rowmajor<Cfg, PRE>(acc, e, m0, u0, lds_dyn, pre);
    return;
  }
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
// TWO_SEG (round 5, the wavefront pair launches of the M ~ batch stacks): K walks nk1 steps of A1 against B, then nk2 steps of A2 against B2
// (B2's k index restarts at 0; B2 == NULL: B's runs on) - layer 0's step with the gradient from the layer above contracted in the same
// launch, [dz0_{t+1} | dz1_t] . [Wh0 ; Wx1]^T.  The wave's K slice may lie in either segment or straddle them (scalar selects per step).
template <int KW, int STAGES, bool TWO_SEG>
__device__ __forceinline__ void lstm_bwd_skinny_lds_body(const GemmOperands& p, const LstmBwdParams& e, const int tiles_m, const int tiles_n, const int bid) {
  constexpr int NT = 64 * KW;
  constexpr int STAGE = 8192, RING = STAGES * STAGE;                  // A 32 rows x 128 B | B 32 rows x 128 B
  float (*part)[32][36] = (float (*)[32][36])lds_dyn;                 // [wave][row][unit] (+4 pad): ALIASES the rings (18 KB of KW x RING >= 64 KB),
                                                                      // written behind a barrier once every wave has left its loop
  const int nwg = tiles_m * tiles_n;
  const int id = xcd_remap(bid, nwg);
  const int tm = id % tiles_m, tn = id / tiles_m;
  const int m0 = tm * 32, u0 = tn * 32;
  if (m0 >= e.m_active) {
    lstm_bwd_zero_tile<32, 32, NT>(e, m0, u0);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* ring = lds_dyn + wave * RING;
  const int nk = TWO_SEG ? p.nk1 + p.nk2 : p.nk1;                     // 64-wide K steps
  const int per = (nk + KW - 1) / KW;
  const int k0 = min(wave * per, nk), k1 = min(nk, wave * per + per);
  const int n = k1 - k0;                                              // this wave's K steps (may be 0)
  // staging sources: chunk c = lane + i*64 -> row c>>3, physical 16-B chunk c&7 holds logical chunk (c&7)^(row&7)
  const int lc8 = ((lane & 7) ^ ((lane >> 3) & 7)) * 8;
  const bf16_t* asrc[4];
  const bf16_t* bsrc[4];
  long a2off[4], boff[4];                                             // TWO_SEG: the lane's row offsets into A2 / B2
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (lane >> 3) + i * 8;
    const int m = min(m0 + r, p.M - 1), u = min(u0 + r, p.Nu - 1);
    asrc[i] = p.A1 + (long)m * p.lda1 + (long)k0 * 64 + lc8;
    bsrc[i] = p.B + (long)u * p.ldb + (long)k0 * 64 + lc8;
    a2off[i] = (long)m * p.lda2 + lc8;
    boff[i] = (long)u * p.ldb + lc8;
  }
  const bf16_t* const b2 = (TWO_SEG && p.B2) ? p.B2 : p.B + (long)p.nk1 * 64;
  auto stage = [&](int j) {                                           // K step j of this wave -> ring slot j % STAGES
    char* sb = ring + (j % STAGES) * STAGE;
    const int kk = k0 + j;
    const bool s2 = TWO_SEG && kk >= p.nk1;                           // wave-uniform
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16_t* src = s2 ? p.A2 + a2off[i] + (long)(kk - p.nk1) * 64 : asrc[i] + (long)j * 64;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sb + i * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16_t* src = s2 ? b2 + boff[i] + (long)(kk - p.nk1) * 64 : bsrc[i] + (long)j * 64;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sb + 4096 + i * 1024), 16, 0, 0);
    }
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

template <int KW, int STAGES>
__global__ __launch_bounds__(64 * KW) void lstm_bwd_step_skinny_lds_kernel(GemmOperands p, LstmBwdParams e, int tiles_m, int tiles_n) {
  lstm_bwd_skinny_lds_body<KW, STAGES, false>(p, e, tiles_m, tiles_n, blockIdx.x);
}

// Two independent steps in one launch (the first tiles_m*tiles_n workgroups run step a, the rest step b): layer 0's step t+1 - with the
// gradient from layer 1 contracted in its own K walk - and layer 1's step t of a two-layer M ~ batch stack.  These steps are chains of
// dependent ~12 us launches; at 64 KB of LDS two of their workgroups share a CU, so the stack's BPTT is T + 1 dependent launches instead of
// 2 T + a hoisted dX product (the forward twin: lstm_fwd_pair_kernel).
template <int KW, int STAGES>
__global__ __launch_bounds__(64 * KW) void lstm_bwd_skinny_pair_kernel(GemmOperands pa, LstmBwdParams ea, GemmOperands pb, LstmBwdParams eb,
                                                                       int tiles_m, int tiles_n) {
  const int n = tiles_m * tiles_n;
  const bool first = blockIdx.x < n;                   // workgroup-uniform
  const GemmOperands p = first ? pa : pb;
  const LstmBwdParams e = first ? ea : eb;
  lstm_bwd_skinny_lds_body<KW, STAGES, true>(p, e, tiles_m, tiles_n, first ? blockIdx.x : blockIdx.x - n);
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
  const bool skinny = M <= 512;                      // M ~ batch stacks (the L2 levels): 32 x 32 tiles, K split over four waves, 64 KB of LDS
  const int kdv = skinny ? 64 : 32;                  // K step of the loop the launch runs on
  const int tn = skinny ? ceil_div(H, 32) : ceil_div(H, Cfg::BU), tm = skinny ? ceil_div(M, 32) : ceil_div(M, Cfg::BM);
  constexpr int SKW = 4, SSTG = 2;
  if (skinny) {
    allow_big_lds((const void*)lstm_bwd_skinny_pair_kernel<SKW, SSTG>, SKW * SSTG * 8192);
    allow_big_lds((const void*)lstm_bwd_step_skinny_lds_kernel<SKW, SSTG>, SKW * SSTG * 8192);
  }
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
      pa.nk1 = (t0 == T - 1) ? 0 : 4 * H / kdv;
      pa.A2 = dz1 + (long)t0 * slab * 4;
      pa.nk2 = 4 * H / kdv;
      pa.B = w_il0 + (long)Kin0 * 4 * H;              // Wh0: rows Kin0 .. Kin0+H-1 of layer 0's kernel
      pa.B2 = w_il1;                                  // Wx1: rows 0 .. H-1 of layer 1's kernel
      tail(ea, 0, t0);
    }
    if (has_b) {                                      // layer 1, step t1: dz1_{t1+1} . Wh1^T
      base(pb);
      pb.A1 = dz1 + (long)(t1 + 1 < T ? t1 + 1 : t1) * slab * 4;
      pb.nk1 = (t1 == T - 1) ? 0 : 4 * H / kdv;
      pb.A2 = pb.A1;
      pb.B = w_il1 + (long)H * 4 * H;                 // Wh1
      tail(eb, 1, t1);
    }
    if (skinny) {
      constexpr int LDS = SKW * SSTG * 8192;
      if (has_a && has_b) hipLaunchKernelGGL((lstm_bwd_skinny_pair_kernel<SKW, SSTG>), dim3(2 * tm * tn), dim3(64 * SKW), LDS, st, pa, ea, pb, eb, tm, tn);
      else if (has_a) hipLaunchKernelGGL((lstm_bwd_skinny_pair_kernel<SKW, SSTG>), dim3(tm * tn), dim3(64 * SKW), LDS, st, pa, ea, pa, ea, tm, tn);   // (two-segment body, one role)
      else hipLaunchKernelGGL((lstm_bwd_step_skinny_lds_kernel<SKW, SSTG>), dim3(tm * tn), dim3(64 * SKW), LDS, st, pb, eb, tm, tn);
      continue;
    }
    if (has_a && has_b) launch_cfg<Cfg>(lstm_bwd_pair_kernel<Cfg>, 2 * tm * tn, st, pa, ea, tm, pb, eb, tm, tn);
    else if (has_a) launch_cfg<Cfg>(lstm_bwd_step_kernel<Cfg>, tm * tn, st, pa, ea, tm, tn);
    else launch_cfg<Cfg>(lstm_bwd_step_kernel<Cfg>, tm * tn, st, pb, eb, tm, tn);
  }
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

