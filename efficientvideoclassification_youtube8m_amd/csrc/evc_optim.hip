// One LSTM layer's train op in two launches (a9: slim.learning.create_train_op with clip_gradient_norm, cs/train.py:329-334,
// 413-418; tf.train.AdamOptimizer, cs/train.py:241-242) instead of five to eight:
//   evc_sqnorm2_partials   per-block sums of squares of the kernel gradient and of the bias gradient (plain stores, no atomics)
//   evc_lstm_adam_fused    per-tensor clip_by_norm (the partials are summed in index order by every workgroup: run-to-run identical
//                          for identical gradients) + TF-Adam of kernel AND bias, and every operand image the next step reads, from
//                          the same registers: the bf16 forward shadow, the transposed gate-interleaved bf16 backward shadow (through
//                          an LDS transpose: whole 128-byte lines), and - "high" precision - the IEEE f16 image (optionally with its
//                          K-extension blocks) and the e4m3 low-order image of evc_lstm_layer_fwd_f16_fp8lo / evc_lstm_stack2_fwd_f16_fp8lo.
// Before: grad_sqnorm x 2, clip_adam x 2, transpose (+ cast_f16 / cast_f16_wide + cast_fp8_lo in "high") per layer: 36 + 12 launches
// per training step on both towers, the weights re-read for every image.  Same arithmetic as clip_adam_kernel (evc_elementwise.hip).
#include "evc_common.h"

#define EVC_SQN_BLOCKS 1024     // partial sums of the first tensor (4 workgroups per CU: the pass is a plain HBM stream); the second tensor (a bias) is one more block

__device__ __forceinline__ float block_sum_256(float v, float* sh) {      // butterflies inside a wave, the 4 wave totals in order
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void sqnorm2_partials_kernel(const float* __restrict__ a, long na, const float* __restrict__ b, long nb, int nba,
                                                               float* __restrict__ part) {
  __shared__ float sh[4];
  const bool second = (int)blockIdx.x >= nba;
  const float* g = second ? b : a;
  const long n = second ? nb : na;
  const long first = second ? 0 : blockIdx.x, stride = second ? 1 : nba;
  float s = 0.f;
  const long n4 = n >> 2;
  long i = first * 256 + threadIdx.x;
  for (; i + 3 * stride * 256 < n4; i += 4 * stride * 256) {       // four 16-byte loads in flight per lane
    const float4 v0 = ((const float4*)g)[i], v1 = ((const float4*)g)[i + stride * 256], v2 = ((const float4*)g)[i + 2 * stride * 256],
                 v3 = ((const float4*)g)[i + 3 * stride * 256];
    s += v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w;
    s += v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w;
    s += v2.x * v2.x + v2.y * v2.y + v2.z * v2.z + v2.w * v2.w;
    s += v3.x * v3.x + v3.y * v3.y + v3.z * v3.z + v3.w * v3.w;
  }
  for (; i < n4; i += stride * 256) {
    const float4 v = ((const float4*)g)[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  for (long k = (n4 << 2) + first * 256 + threadIdx.x; k < n; k += stride * 256) s += g[k] * g[k];
  s = block_sum_256(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

extern "C" int evc_sqnorm2_partials(const float* ga, int64_t na, const float* gb, int64_t nb, float* part, void* stream) {
  EVC_REQUIRE(ga && na > 0 && part && (gb != nullptr) == (nb > 0), EVC_ERR_BAD_ARG, "evc_sqnorm2_partials: na=%ld nb=%ld", (long)na, (long)nb);
  EVC_REQUIRE(((uintptr_t)ga % 16) == 0 && ((uintptr_t)gb % 16) == 0, EVC_ERR_BAD_ALIGN, "evc_sqnorm2_partials: 16-byte alignment");
  hipLaunchKernelGGL(sqnorm2_partials_kernel, dim3(EVC_SQN_BLOCKS + (gb ? 1 : 0)), dim3(256), 0, (hipStream_t)stream, ga, (long)na, gb, (long)nb,
                     EVC_SQN_BLOCKS, part);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

struct LstmAdamParams {
  float* p; const float* g; float* m; float* v;        // kernel W [R = 4H][C] f32, contiguous
  float* pb; const float* gb; float* mb; float* vb;    // bias [R] f32
  int R, C, H;
  const float* part;                                   // [EVC_SQN_BLOCKS + 1] from evc_sqnorm2_partials
  float* sums_w; float* sums_b;                        // norm rows of the two tensors: [0] = sum g^2 (assigned), [1] untouched
  float clip, lr_t, b1, b2, eps;
  bf16_t* p_bf16;                                      // forward shadow [R][C]
  bf16_t* pT; long ldT;                                // backward shadow [C][ldT]: column u*4+g <- row g*H+u
  f16_t* p16; long ld16; int nin, nseg;                // or NULL: f16 image rows [f16(Wx) | f16(Wx)/64 | (Wx - f16(Wx))*64 (first nseg blocks) | f16(Wh)]
  uint8_t* p8; long ld8; int col0, hi_cols;            // or NULL: e4m3 image of W[:, col0:]: [lo(first hi_cols) | hi(first hi_cols) | lo(rest)]
  int hi_tail = 0;                                     // ... | hi(rest)] as well (evc_cast_f32_to_fp8_lohi)
  float lo_scale, hi_scale;
  int tiles_c, n_tiles;
};

__device__ __forceinline__ float sum_partials(const float* __restrict__ part, int n) {     // every caller gets the same bits: index order per lane, butterfly
  float s = 0.f;
  for (int i = threadIdx.x & 63; i < n; i += 64) s += part[i];
  return wave_sum(s);
}

// IL = true: an LSTM kernel [4H][C] (row g*H+u; backward shadow column u*4+g) with its bias; IL = false: any 2-D weight [R][C] stored as
// the forward GEMM's B operand (backward shadow = its plain transpose, zero-padded to ldT columns), no bias
#ifndef EVC_ADAM_NT
#define EVC_ADAM_NT 1       // (round 5: 1 = the master / moment streams of the fused update - read once, written once, 3 GB per step - as non-temporal accesses: same box, alternated three times, 10.00 -> 9.93 ms per step; 0 = plain accesses; 2 = the bf16 shadows too - they are read again soon: 9.82 -> 9.93)
#endif
__device__ __forceinline__ float4 ld_stream(const float* p) {
#if EVC_ADAM_NT
  const f32x4 v = __builtin_nontemporal_load((const f32x4*)p);
  return make_float4(v[0], v[1], v[2], v[3]);
#else
  return *(const float4*)p;
#endif
}
__device__ __forceinline__ void st_stream(float* p, float a, float b, float c, float d) {
#if EVC_ADAM_NT
  __builtin_nontemporal_store(f32x4{a, b, c, d}, (f32x4*)p);
#else
  *(float4*)p = make_float4(a, b, c, d);
#endif
}
template <bool IL>
__global__ __launch_bounds__(256) void lstm_adam_fused_kernel(LstmAdamParams u) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64][72];      // [k][u*4+g] of one 16-unit x 64-column tile
  const int t = threadIdx.x;
  if (IL && (int)blockIdx.x >= u.n_tiles) {            // bias blocks: 1024 elements each
    const float ss = u.part[EVC_SQN_BLOCKS];
    const float scale = u.clip > 0.f ? u.clip / fmaxf(sqrtf(ss), u.clip) : 1.f;
    const int j = (blockIdx.x - u.n_tiles) * 1024 + t * 4;
    if (blockIdx.x == (unsigned)u.n_tiles && t == 0) u.sums_b[0] = ss;
    if (j >= u.R) return;
    const float4 pv = *(const float4*)(u.pb + j), gv = *(const float4*)(u.gb + j), mv = *(const float4*)(u.mb + j), vv = *(const float4*)(u.vb + j);
    const float pa[4] = {pv.x, pv.y, pv.z, pv.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w}, ma[4] = {mv.x, mv.y, mv.z, mv.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
    float pn[4], mn[4], vn[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float gc = ga[r] * scale;
      mn[r] = u.b1 * ma[r] + (1.f - u.b1) * gc;
      vn[r] = u.b2 * va[r] + (1.f - u.b2) * gc * gc;
      pn[r] = adam_step_(pa[r], mn[r], vn[r], u.lr_t, u.eps);
    }
    *(float4*)(u.mb + j) = make_float4(mn[0], mn[1], mn[2], mn[3]);
    *(float4*)(u.vb + j) = make_float4(vn[0], vn[1], vn[2], vn[3]);
    *(float4*)(u.pb + j) = make_float4(pn[0], pn[1], pn[2], pn[3]);
    return;
  }
  const int tr = blockIdx.x / u.tiles_c, tcn = blockIdx.x % u.tiles_c;
  const int u0 = tr * 16, k0 = tcn * 64;
  const int c4 = t & 15, i = t >> 4;                   // this thread: unit u0 + i, columns k0 + 4 c4 .. + 3, all four gates
  const int col = k0 + c4 * 4;
  const bool okc = col < u.C;                          // (C % 4 == 0: a lane's 4 columns are all valid or all not)
  auto row_of = [&](int g) { return IL ? g * u.H + u0 + i : tr * 64 + g * 16 + i; };
  float4 pv[4], gv[4], mv[4], vv[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {                        // all loads first (16 x 16 bytes in flight per lane)
    const long o = (long)row_of(g) * u.C + col;
    if (okc && row_of(g) < u.R) { pv[g] = ld_stream(u.p + o); gv[g] = ld_stream(u.g + o); mv[g] = ld_stream(u.m + o); vv[g] = ld_stream(u.v + o); }
    else pv[g] = gv[g] = mv[g] = vv[g] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float ss = sum_partials(u.part, EVC_SQN_BLOCKS);
  const float scale = u.clip > 0.f ? u.clip / fmaxf(sqrtf(ss), u.clip) : 1.f;      // tf.clip_by_norm
  if (blockIdx.x == 0 && t == 0) u.sums_w[0] = ss;
  uint32_t pbits[4][2];                                // bf16 of the new weights: [gate][pair of columns]
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float pa[4] = {pv[g].x, pv[g].y, pv[g].z, pv[g].w}, ga[4] = {gv[g].x, gv[g].y, gv[g].z, gv[g].w};
    const float ma[4] = {mv[g].x, mv[g].y, mv[g].z, mv[g].w}, va[4] = {vv[g].x, vv[g].y, vv[g].z, vv[g].w};
    float pn[4], mn[4], vn[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {                      // same operation order as clip_adam_kernel (l2 = 0 for these tensors)
      const float gc = ga[r] * scale;
      mn[r] = u.b1 * ma[r] + (1.f - u.b1) * gc;
      vn[r] = u.b2 * va[r] + (1.f - u.b2) * gc * gc;
      pn[r] = adam_step_(pa[r], mn[r], vn[r], u.lr_t, u.eps);
    }
    pbits[g][0] = pack_bf16x2_hw(pn[0], pn[1]);
    pbits[g][1] = pack_bf16x2_hw(pn[2], pn[3]);
    const long row = row_of(g);
    if (!okc || row >= u.R) continue;
    const long o = row * u.C + col;
    st_stream(u.m + o, mn[0], mn[1], mn[2], mn[3]);
    st_stream(u.v + o, vn[0], vn[1], vn[2], vn[3]);
    st_stream(u.p + o, pn[0], pn[1], pn[2], pn[3]);
#if EVC_ADAM_NT >= 2
    __builtin_nontemporal_store(u32x2_t{pbits[g][0], pbits[g][1]}, (u32x2_t*)(u.p_bf16 + o));
#else
    *(uint2*)(u.p_bf16 + o) = make_uint2(pbits[g][0], pbits[g][1]);
#endif
    if (u.p16 || u.p8) {
      const uint32_t h01 = pack_f16x2_hw(pn[0], pn[1]), h23 = pack_f16x2_hw(pn[2], pn[3]);
      const float hf[4] = {f16_to_f32((f16_t)(h01 & 0xffffu)), f16_to_f32((f16_t)(h01 >> 16)), f16_to_f32((f16_t)(h23 & 0xffffu)), f16_to_f32((f16_t)(h23 >> 16))};
      if (u.p16) {                                     // evc_cast_f32_to_f16 / _f16_wide (h_ext = 0)
        f16_t* o16 = u.p16 + row * u.ld16;
        if (col >= u.nin) {
          *(uint2*)(o16 + (long)u.nseg * u.nin + (col - u.nin)) = make_uint2(h01, h23);
        } else {
          *(uint2*)(o16 + col) = make_uint2(h01, h23);
          if (u.nseg >= 2)
            *(uint2*)(o16 + u.nin + col) = make_uint2(pack_f16x2_hw(hf[0] * (1.0f / 64.0f), hf[1] * (1.0f / 64.0f)), pack_f16x2_hw(hf[2] * (1.0f / 64.0f), hf[3] * (1.0f / 64.0f)));
          if (u.nseg >= 3)
            *(uint2*)(o16 + 2L * u.nin + col) = make_uint2(pack_f16x2_hw((pn[0] - hf[0]) * 64.0f, (pn[1] - hf[1]) * 64.0f), pack_f16x2_hw((pn[2] - hf[2]) * 64.0f, (pn[3] - hf[3]) * 64.0f));
        }
      }
      if (u.p8 && col >= u.col0) {                     // evc_cast_f32_to_fp8_lo on W[:, col0:]
        const int cc = col - u.col0;
        float d[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] = fminf(fmaxf((pn[r] - hf[r]) * u.lo_scale, -448.f), 448.f);
        int w8 = __builtin_amdgcn_cvt_pk_fp8_f32(d[0], d[1], 0, false);
        w8 = __builtin_amdgcn_cvt_pk_fp8_f32(d[2], d[3], w8, true);
        uint8_t* o8 = u.p8 + row * u.ld8;
        *(int*)(o8 + (cc < u.hi_cols ? cc : cc + u.hi_cols)) = w8;
        if (cc < u.hi_cols || u.hi_tail) {
#pragma unroll
          for (int r = 0; r < 4; ++r) d[r] = fminf(fmaxf(pn[r] * u.hi_scale, -448.f), 448.f);
          int h8 = __builtin_amdgcn_cvt_pk_fp8_f32(d[0], d[1], 0, false);
          h8 = __builtin_amdgcn_cvt_pk_fp8_f32(d[2], d[3], h8, true);
          *(int*)(o8 + (cc < u.hi_cols ? u.hi_cols + cc : (u.C - u.col0) + cc)) = h8;      // (hi_tail: behind the C - col0 columns of [lo(A) | hi(A) | lo(B)])
        }
      }
    }
  }
  // transposed, gate-interleaved backward shadow: tile[k][i*4 + g], then whole rows of 64 bf16 (128 bytes) per k
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    uint32_t lo, hi;                                   // gates 0,1 and 2,3 of column c4*4 + e
    if (e & 1) {
      lo = (pbits[0][e >> 1] >> 16) | (pbits[1][e >> 1] & 0xffff0000u);
      hi = (pbits[2][e >> 1] >> 16) | (pbits[3][e >> 1] & 0xffff0000u);
    } else {
      lo = (pbits[0][e >> 1] & 0xffffu) | (pbits[1][e >> 1] << 16);
      hi = (pbits[2][e >> 1] & 0xffffu) | (pbits[3][e >> 1] << 16);
    }
    if (IL) {
      *(uint2*)&tile[c4 * 4 + e][i * 4] = make_uint2(lo, hi);
    } else {                                           // plain transpose: tile[k][local row g*16 + i]
      tile[c4 * 4 + e][i] = (bf16_t)(lo & 0xffffu);
      tile[c4 * 4 + e][16 + i] = (bf16_t)(lo >> 16);
      tile[c4 * 4 + e][32 + i] = (bf16_t)(hi & 0xffffu);
      tile[c4 * 4 + e][48 + i] = (bf16_t)(hi >> 16);
    }
  }
  __syncthreads();
  const int kk = t >> 2, part4 = t & 3;
  if (k0 + kk < u.C) {
    const uint4 q0 = *(const uint4*)&tile[kk][part4 * 16], q1 = *(const uint4*)&tile[kk][part4 * 16 + 8];
    bf16_t* dst = u.pT + (long)(k0 + kk) * u.ldT + (IL ? (long)u0 * 4 : (long)tr * 64) + part4 * 16;
#if EVC_ADAM_NT >= 2
    __builtin_nontemporal_store(u32x4_t{q0.x, q0.y, q0.z, q0.w}, (u32x4_t*)dst);
    __builtin_nontemporal_store(u32x4_t{q1.x, q1.y, q1.z, q1.w}, (u32x4_t*)(dst + 8));
#else
    *(uint4*)dst = q0;
    *(uint4*)(dst + 8) = q1;
#endif
  }
}

extern "C" int evc_lstm_adam_fused(float* p, const float* g, float* m, float* v, float* pb, const float* gb, float* mb, float* vb, int H, int C,
                                   const float* part, float* sums_w, float* sums_b, float clip_norm, float lr_t, float beta1, float beta2, float eps,
                                   evc_bf16* p_bf16, evc_bf16* pT_bf16, int64_t ldT, evc_f16* p_f16, int64_t ld16, int nin, int nseg,
                                   uint8_t* p_fp8, int64_t ld8, int fp8_col0, int fp8_hi_cols, int fp8_lo_exp, int fp8_hi_exp, int fp8_hi_tail, void* stream) {
  EVC_REQUIRE(H > 0 && H % 16 == 0 && C > 0 && C % 4 == 0, EVC_ERR_BAD_SHAPE, "evc_lstm_adam_fused: H=%d (%%16) C=%d (%%4)", H, C);
  EVC_REQUIRE(fp8_hi_tail == 0 || (fp8_hi_tail == 1 && p_fp8 && ld8 >= 2L * (C - fp8_col0)), EVC_ERR_BAD_ARG,
              "evc_lstm_adam_fused: fp8_hi_tail=%d needs p_fp8 with ld8 >= 2 (C - col0) (ld8=%ld)", fp8_hi_tail, (long)ld8);
  EVC_REQUIRE(p && g && m && v && pb && gb && mb && vb && part && sums_w && sums_b && p_bf16 && pT_bf16, EVC_ERR_BAD_ARG, "evc_lstm_adam_fused: NULL argument");
  EVC_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0 && ((uintptr_t)pb % 16) == 0 &&
              ((uintptr_t)gb % 16) == 0 && ((uintptr_t)mb % 16) == 0 && ((uintptr_t)vb % 16) == 0 && ((uintptr_t)p_bf16 % 8) == 0 &&
              ((uintptr_t)pT_bf16 % 16) == 0 && ldT % 8 == 0 && ldT >= 4L * H, EVC_ERR_BAD_ALIGN, "evc_lstm_adam_fused: alignment / ldT=%ld", (long)ldT);
  EVC_REQUIRE(!p_f16 || (((uintptr_t)p_f16 % 8) == 0 && nseg >= 1 && nseg <= 3 && nin >= 0 && nin <= C && nin % 4 == 0 && ld16 % 4 == 0 &&
                         ld16 >= (long)nseg * nin + (C - nin)), EVC_ERR_BAD_ARG, "evc_lstm_adam_fused: f16 image nin=%d nseg=%d ld16=%ld", nin, nseg, (long)ld16);
  EVC_REQUIRE(!p_fp8 || (((uintptr_t)p_fp8 % 4) == 0 && fp8_col0 >= 0 && fp8_col0 < C && fp8_col0 % 4 == 0 && fp8_hi_cols >= 0 && fp8_hi_cols % 4 == 0 &&
                         fp8_hi_cols <= C - fp8_col0 && ld8 % 4 == 0 && ld8 >= (long)(C - fp8_col0) + fp8_hi_cols && fp8_lo_exp >= 0 && fp8_lo_exp <= 60 &&
                         fp8_hi_exp >= -30 && fp8_hi_exp <= 30), EVC_ERR_BAD_ARG,
              "evc_lstm_adam_fused: e4m3 image col0=%d hi_cols=%d ld8=%ld lo_exp=%d hi_exp=%d", fp8_col0, fp8_hi_cols, (long)ld8, fp8_lo_exp, fp8_hi_exp);
  LstmAdamParams u;
  u.p = p; u.g = g; u.m = m; u.v = v; u.pb = pb; u.gb = gb; u.mb = mb; u.vb = vb;
  u.R = 4 * H; u.C = C; u.H = H; u.part = part; u.sums_w = sums_w; u.sums_b = sums_b;
  u.clip = clip_norm; u.lr_t = lr_t; u.b1 = beta1; u.b2 = beta2; u.eps = eps;
  u.p_bf16 = (bf16_t*)p_bf16; u.pT = (bf16_t*)pT_bf16; u.ldT = ldT;
  u.p16 = (f16_t*)p_f16; u.ld16 = ld16; u.nin = nin; u.nseg = nseg;
  u.p8 = p_fp8; u.ld8 = ld8; u.col0 = fp8_col0; u.hi_cols = fp8_hi_cols; u.hi_tail = fp8_hi_tail;
  u.lo_scale = ldexpf(1.0f, fp8_lo_exp); u.hi_scale = ldexpf(1.0f, fp8_hi_exp);
  u.tiles_c = (C + 63) / 64;
  u.n_tiles = (H / 16) * u.tiles_c;
  const int bias_blocks = (4 * H + 1023) / 1024;
  hipLaunchKernelGGL(lstm_adam_fused_kernel<true>, dim3(u.n_tiles + bias_blocks), dim3(256), 0, (hipStream_t)stream, u);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// The same pass for a plain 2-D weight p [R][C] (stored as the forward GEMM's B operand: DBoF cluster / hidden weights): clip + TF-Adam from
// the partials of evc_sqnorm2_partials(g, n, NULL, 0), bf16 forward shadow, transposed bf16 backward shadow pT [C][ldT] (ldT >= round_up(R, 64);
// the pad columns are written as zeros), and the optional f16 / e4m3 images (arguments as evc_lstm_adam_fused with nin = C: no K-extension blocks).
extern "C" int evc_adam2d_fused(float* p, const float* g, float* m, float* v, int R, int C, const float* part, float* sums_w, float clip_norm, float lr_t,
                                float beta1, float beta2, float eps, evc_bf16* p_bf16, evc_bf16* pT_bf16, int64_t ldT, evc_f16* p_f16, int64_t ld16,
                                uint8_t* p_fp8, int64_t ld8, int fp8_hi_cols, int fp8_lo_exp, int fp8_hi_exp, void* stream) {
  EVC_REQUIRE(R > 0 && C > 0 && C % 4 == 0, EVC_ERR_BAD_SHAPE, "evc_adam2d_fused: R=%d C=%d (%%4)", R, C);
  EVC_REQUIRE(p && g && m && v && part && sums_w && p_bf16 && pT_bf16, EVC_ERR_BAD_ARG, "evc_adam2d_fused: NULL argument");
  EVC_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0 && ((uintptr_t)p_bf16 % 8) == 0 &&
              ((uintptr_t)pT_bf16 % 16) == 0 && ldT % 8 == 0 && ldT >= (R + 63) / 64 * 64, EVC_ERR_BAD_ALIGN, "evc_adam2d_fused: alignment / ldT=%ld (>= R rounded up to 64)", (long)ldT);
  EVC_REQUIRE(!p_f16 || (((uintptr_t)p_f16 % 8) == 0 && ld16 % 4 == 0 && ld16 >= C), EVC_ERR_BAD_ARG, "evc_adam2d_fused: f16 image ld16=%ld", (long)ld16);
  EVC_REQUIRE(!p_fp8 || (((uintptr_t)p_fp8 % 4) == 0 && fp8_hi_cols >= 0 && fp8_hi_cols % 4 == 0 && fp8_hi_cols <= C && ld8 % 4 == 0 && ld8 >= (long)C + fp8_hi_cols &&
                         fp8_lo_exp >= 0 && fp8_lo_exp <= 60 && fp8_hi_exp >= -30 && fp8_hi_exp <= 30), EVC_ERR_BAD_ARG,
              "evc_adam2d_fused: e4m3 image hi_cols=%d ld8=%ld lo_exp=%d hi_exp=%d", fp8_hi_cols, (long)ld8, fp8_lo_exp, fp8_hi_exp);
  LstmAdamParams u;
  u.p = p; u.g = g; u.m = m; u.v = v; u.pb = nullptr; u.gb = nullptr; u.mb = nullptr; u.vb = nullptr;
  u.R = R; u.C = C; u.H = 0; u.part = part; u.sums_w = sums_w; u.sums_b = nullptr;
  u.clip = clip_norm; u.lr_t = lr_t; u.b1 = beta1; u.b2 = beta2; u.eps = eps;
  u.p_bf16 = (bf16_t*)p_bf16; u.pT = (bf16_t*)pT_bf16; u.ldT = ldT;
  u.p16 = (f16_t*)p_f16; u.ld16 = ld16; u.nin = C; u.nseg = 1;
  u.p8 = p_fp8; u.ld8 = ld8; u.col0 = 0; u.hi_cols = fp8_hi_cols;
  u.lo_scale = ldexpf(1.0f, fp8_lo_exp); u.hi_scale = ldexpf(1.0f, fp8_hi_exp);
  u.tiles_c = (C + 63) / 64;
  u.n_tiles = ((R + 63) / 64) * u.tiles_c;
  hipLaunchKernelGGL(lstm_adam_fused_kernel<false>, dim3(u.n_tiles), dim3(256), 0, (hipStream_t)stream, u);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}
