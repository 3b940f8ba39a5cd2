// DbofModel hot path (cs/frame_level_models.py:108-195, cs/model_utils.py:39-83):
// SampleRandomFrames -> input batch-norm -> cluster GEMM -> cluster batch-norm -> relu6 -> max over frames,
// with the [frames x clusters] activation never leaving the chip in f32:
//
//   evc_dbof_gather            sampled frames (uint8 or f32 input, l2-normalised) into the padded frame layout
//                              + per-workgroup column partial sums for the input batch-norm
//   evc_dbof_input_bn_apply    r -> r_bn (bf16, the cluster GEMM's A operand) and xhat (bf16, the weight-gradient
//                              product's operand)
//   evc_dbof_cluster_pool_fwd  the cluster GEMM on 256x256 MFMA tiles whose epilogue reduces, per column, sum x and
//                              sum x^2 over the tile's frames (batch-norm statistics) and, per (video, column), the
//                              max over the video's frames of sign(gamma) * x with its frame index - batch-norm with
//                              a positive (negative) scale and relu6 are monotone, so max_s relu6(bn(x_s)) =
//                              relu6(bn(max_s x_s)) (min for a negative scale) and the statistics need not be known
//                              before the reduction.  Training additionally keeps x as bf16 for the backward pass.
//   evc_dbof_pool_finish       pooled = relu6(bn(selected x)) once the statistics are final
//   evc_dbof_dact              backward of max-pool + relu6 + cluster batch-norm, in place on the bf16 activation
//   evc_dbof_wgrad_finish      cluster-weight gradient and input batch-norm scale gradient from G = dact^T . xhat
//
// Padded frame layout ("video-in-quad"): every video owns SP = 32 frame slots (S <= 32 sampled frames, the rest
// zero); 4 videos form a 128-row block and frame s of video b sits at row
//     (b >> 2) * 128 + (s >> 2) * 16 + (b & 3) * 4 + (s & 3).
// In the transposed 16x16 MFMA accumulator layout (lane = 16 g + l holds row mi*16 + l, columns 4g..4g+3) the 32
// frames of one video are then the 8 accumulator blocks (mi) of the 4 lanes of one quad (l >> 2 == b & 3): the
// reduction over frames is in-lane plus two quad-permute DPP steps, no LDS traffic.
#include "gemm_launch.h"

static constexpr int SP = 32;                      // frame slots per video
__host__ __device__ static inline long dbof_row(int b, int s) {
  return (long)(b >> 2) * 128 + (s >> 2) * 16 + (b & 3) * 4 + (s & 3);
}

// ---------------------------------------------------------------------------------------------------------------
// DPP helpers (wave64; a DPP "row" is 16 lanes)
// ---------------------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
static constexpr int QP_XOR1 = 0xB1, QP_XOR2 = 0x4E, ROW_ROR4 = 0x124, ROW_ROR8 = 0x128;
__device__ __forceinline__ float row16_sum(float v) {     // every lane of a 16-lane row gets the row's sum
  v += dpp_f<QP_XOR1>(v);
  v += dpp_f<QP_XOR2>(v);
  v += dpp_f<ROW_ROR4>(v);
  v += dpp_f<ROW_ROR8>(v);
  return v;
}

// ---------------------------------------------------------------------------------------------------------------
// K1: SampleRandomFrames (cs/model_utils.py:39-58) + tf.nn.l2_normalize of the gathered frames (cs/train.py:256)
// into the padded frame layout, f32, with the column sums for the input batch-norm.
// One wave per sampled frame; a workgroup of 4 waves handles ROWS_PER_WG frames and leaves one partial row.
// ---------------------------------------------------------------------------------------------------------------
static constexpr int GATHER_ROWS_PER_WG = 32;
__global__ __launch_bounds__(256) void dbof_gather_kernel(const float* __restrict__ xf, const uint8_t* __restrict__ xq,
                                                          const float* __restrict__ u, const int* __restrict__ nfr, int B, int T,
                                                          int F, int S, int normalize, float* __restrict__ r,
                                                          int* __restrict__ idx_out, float* __restrict__ part) {
  __shared__ float4 sh[2][4][64 * 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int F4 = F >> 2;
  float4 sum[8], sq[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) sum[i] = sq[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const long rows = (long)B * S;
  for (int k = wave; k < GATHER_ROWS_PER_WG; k += 4) {
    const long row = (long)blockIdx.x * GATHER_ROWS_PER_WG + k;
    if (row >= rows) break;
    const int b = (int)(row / S), s = (int)(row % S);
    const int n = nfr[b];
    // tf.cast(tf.multiply(random_uniform, tf.cast(num_frames, tf.float32)), tf.int32)
    int idx = (int)(u[row] * (float)n);
    if (lane == 0 && idx_out) idx_out[row] = idx;
    idx = idx < 0 ? 0 : (idx >= T ? T - 1 : idx);
    const bool padded = xq && idx >= n;               // uint8 input: rows >= num_frames are padding (zero after Dequantize)
    float4 v[8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = lane + 64 * i;
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < F4 && !padded) {
        if (xq) {                                    // Dequantize: q * 4/255 + (4/512 - 2), cs/utils.py:22-25
          const uchar4 q = ((const uchar4*)(xq + ((long)b * T + idx) * F))[j];
          const float sc = 4.0f / 255.0f, bi = 4.0f / 512.0f - 2.0f;
          v[i] = make_float4(q.x * sc + bi, q.y * sc + bi, q.z * sc + bi, q.w * sc + bi);
        } else {
          v[i] = ((const float4*)(xf + ((long)b * T + idx) * F))[j];
        }
        ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
      }
    }
    const float inv = normalize ? rsqrtf(fmaxf(wave_sum(ss), 1e-12f)) : 1.f;
    float4* dst = (float4*)(r + dbof_row(b, s) * F);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = lane + 64 * i;
      if (j < F4) {
        float4 w = v[i];
        w.x *= inv; w.y *= inv; w.z *= inv; w.w *= inv;
        dst[j] = w;
        sum[i].x += w.x; sum[i].y += w.y; sum[i].z += w.z; sum[i].w += w.w;
        sq[i].x += w.x * w.x; sq[i].y += w.y * w.y; sq[i].z += w.z * w.z; sq[i].w += w.w * w.w;
      }
    }
  }
  if (!part) return;
#pragma unroll
  for (int i = 0; i < 8; ++i) { sh[0][wave][lane + 64 * i] = sum[i]; sh[1][wave][lane + 64 * i] = sq[i]; }
  __syncthreads();
  for (int j = threadIdx.x; j < F4; j += 256) {
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      float4 a = sh[st][0][j];
      const float4 b1 = sh[st][1][j], b2 = sh[st][2][j], b3 = sh[st][3][j];
      a.x += b1.x + b2.x + b3.x; a.y += b1.y + b2.y + b3.y; a.z += b1.z + b2.z + b3.z; a.w += b1.w + b2.w + b3.w;
      ((float4*)(part + ((long)blockIdx.x * 2 + st) * F))[j] = a;
    }
  }
}

static inline int dbof_gather_parts(int B, int S) { return (int)(((long)B * S + GATHER_ROWS_PER_WG - 1) / GATHER_ROWS_PER_WG); }
static inline int dbof_padded_rows(int B) { return ((B + 3) / 4) * 128; }
static inline int dbof_gemm_parts(int B) { return 2 * ((dbof_padded_rows(B) + 255) / 256); }

extern "C" int evc_dbof_workspace(int B, int S, int32_t* padded_rows, int32_t* gather_part_rows, int32_t* gemm_part_rows) {
  EVC_REQUIRE(B > 0 && S > 0 && S <= SP, EVC_ERR_BAD_SHAPE, "evc_dbof_workspace: iterations=%d sampled frames, at most %d are supported", S, SP);
  if (padded_rows) *padded_rows = dbof_padded_rows(B);
  if (gather_part_rows) *gather_part_rows = dbof_gather_parts(B, S);
  if (gemm_part_rows) *gemm_part_rows = dbof_gemm_parts(B);
  return EVC_OK;
}

extern "C" int evc_dbof_gather(const float* x_f32, const uint8_t* x_u8, const float* u, const int32_t* num_frames, int B, int T,
                               int F, int S, int normalize, float* r, int32_t* idx_out, float* part, void* stream) {
  EVC_REQUIRE(B > 0 && T > 0 && F > 0 && F % 4 == 0 && F <= 2048, EVC_ERR_BAD_SHAPE, "evc_dbof_gather: F=%d must be a multiple of 4, <= 2048", F);
  EVC_REQUIRE(S > 0 && S <= SP, EVC_ERR_BAD_SHAPE, "evc_dbof_gather: iterations=%d sampled frames, at most %d are supported", S, SP);
  EVC_REQUIRE((x_f32 != nullptr) != (x_u8 != nullptr), EVC_ERR_BAD_ARG, "evc_dbof_gather: exactly one of x_f32 / x_u8");
  hipLaunchKernelGGL(dbof_gather_kernel, dim3(dbof_gather_parts(B, S)), dim3(256), 0, (hipStream_t)stream, x_f32, x_u8, u, num_frames,
                     B, T, F, S, normalize, r, idx_out, part);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// Column partial sums [P][2][C] f32 (row p: sum x, then sum x^2) -> ws f64 [2C], rows added in index order
// (run-to-run identical statistics; under data parallelism ws is what the ranks all-reduce).
__global__ __launch_bounds__(1024) void partials_to_f64_kernel(const float* __restrict__ part, int P, int C, double* __restrict__ ws) {
  // block = 64 columns x 16 row phases; phase y adds rows y, y+16, ... in order, the phase sums are added in phase order
  __shared__ double sh[2][16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  double s = 0.0, q = 0.0;
  if (c < C) {
#pragma unroll 4
    for (int p = ty; p < P; p += 16) { s += part[((long)p * 2) * C + c]; q += part[((long)p * 2 + 1) * C + c]; }
  }
  sh[0][ty][tx] = s;
  sh[1][ty][tx] = q;
  __syncthreads();
  if (ty < 2 && c < C) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sh[ty][i][tx];
    ws[ty * C + c] = t;
  }
}
extern "C" int evc_bn_partials_reduce(const float* part, int P, int C, double* ws, void* stream) {
  EVC_REQUIRE(P > 0 && C > 0 && part && ws, EVC_ERR_BAD_SHAPE, "evc_bn_partials_reduce: bad args");
  hipLaunchKernelGGL(partials_to_f64_kernel, dim3((C + 63) / 64), dim3(1024), 0, (hipStream_t)stream, part, P, C, ws);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}
// mean / biased variance from the (all-reduced) sums + slim.batch_norm's moving-average update in one launch
__global__ void bn_finalize_ema_kernel(const double* __restrict__ ws, int R, int C, float* __restrict__ mean, float* __restrict__ var,
                                       float* __restrict__ mov_mean, float* __restrict__ mov_var, float decay) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mu = ws[c] / R;
  double vv = ws[C + c] / R - mu * mu;
  vv = vv > 0 ? vv : 0;
  mean[c] = (float)mu;
  var[c] = (float)vv;
  if (mov_mean) mov_mean[c] -= (1.f - decay) * (mov_mean[c] - (float)mu);
  if (mov_var) mov_var[c] -= (1.f - decay) * (mov_var[c] - (float)vv);
}
extern "C" int evc_bn_finalize_ema(const double* ws, int R_total, int C, float* mean, float* var, float* moving_mean,
                                   float* moving_var, float decay, void* stream) {
  EVC_REQUIRE(R_total > 0 && C > 0 && ws && mean && var, EVC_ERR_BAD_SHAPE, "evc_bn_finalize_ema: bad args");
  hipLaunchKernelGGL(bn_finalize_ema_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws, R_total, C, mean, var,
                     moving_mean, moving_var, decay);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// K3: input batch-norm applied: r (f32, padded layout) -> r_bn = gamma*xhat+beta and xhat, both bf16 [Mp][F];
// frame slots without a sampled frame are zero in both (they are contracted over by the GEMMs).  One wave per row.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dbof_input_bn_apply_kernel(const float* __restrict__ r, int Mp, int B, int S, int F,
                                                                  const float* __restrict__ mean, const float* __restrict__ var,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  bf16_t* __restrict__ rbn, bf16_t* __restrict__ rbn_lo,
                                                                  bf16_t* __restrict__ xhat) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= Mp) return;
  const int w = row & 127;
  const int s = (w >> 4) * 4 + (w & 3), b = (row >> 7) * 4 + ((w >> 2) & 3);
  const bool live = b < B && s < S;
  const int F4 = F >> 2;
  for (int j = lane; j < F4; j += 64) {
    uint2 o = make_uint2(0u, 0u), ol = make_uint2(0u, 0u), oh = make_uint2(0u, 0u);
    if (live) {
      const float4 v = ((const float4*)(r + (long)row * F))[j];
      const float4 mu = ((const float4*)mean)[j], va = ((const float4*)var)[j], ga = ((const float4*)gamma)[j], be = ((const float4*)beta)[j];
      const float h0 = (v.x - mu.x) * rsqrtf(va.x + 1e-3f), h1 = (v.y - mu.y) * rsqrtf(va.y + 1e-3f);
      const float h2 = (v.z - mu.z) * rsqrtf(va.z + 1e-3f), h3 = (v.w - mu.w) * rsqrtf(va.w + 1e-3f);
      const float y0 = h0 * ga.x + be.x, y1 = h1 * ga.y + be.y, y2 = h2 * ga.z + be.z, y3 = h3 * ga.w + be.w;
      o = make_uint2(pack_bf16x2_hw(y0, y1), pack_bf16x2_hw(y2, y3));
      oh = make_uint2(pack_bf16x2_hw(h0, h1), pack_bf16x2_hw(h2, h3));
      if (rbn_lo) {      // split-bf16 ("high" precision): lo = bf16(y - bf16(y))
        const float r0 = y0 - __uint_as_float(o.x << 16), r1 = y1 - __uint_as_float(o.x & 0xffff0000u);
        const float r2 = y2 - __uint_as_float(o.y << 16), r3 = y3 - __uint_as_float(o.y & 0xffff0000u);
        ol = make_uint2(pack_bf16x2_hw(r0, r1), pack_bf16x2_hw(r2, r3));
      }
    }
    ((uint2*)(rbn + (long)row * F))[j] = o;
    if (rbn_lo) ((uint2*)(rbn_lo + (long)row * F))[j] = ol;
    if (xhat) ((uint2*)(xhat + (long)row * F))[j] = oh;
  }
}
extern "C" int evc_dbof_input_bn_apply(const float* r, int B, int S, int F, const float* mean, const float* var, const float* gamma,
                                       const float* beta, evc_bf16* r_bn, evc_bf16* r_bn_lo, evc_bf16* xhat, void* stream) {
  EVC_REQUIRE(B > 0 && S > 0 && S <= SP && F > 0 && F % 4 == 0 && r && r_bn, EVC_ERR_BAD_SHAPE, "evc_dbof_input_bn_apply: bad args");
  const int Mp = dbof_padded_rows(B);
  hipLaunchKernelGGL(dbof_input_bn_apply_kernel, dim3((Mp + 3) / 4), dim3(256), 0, (hipStream_t)stream, r, Mp, B, S, F, mean, var, gamma,
                     beta, r_bn, r_bn_lo, xhat);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// K3 for the "high" precision forward on f16 + e4m3 operands (evc_dbof_cluster_pool_fwd_f16fp8): rows of 4F bytes
// [f16(y) (F halfwords) | e4m3(y 2^hi_exp) (F bytes) | e4m3((y - f16(y)) 2^lo_exp) (F bytes)], y = gamma*xhat+beta; xhat as in K3.
__global__ __launch_bounds__(256) void dbof_input_bn_apply_f16fp8_kernel(const float* __restrict__ r, int Mp, int B, int S, int F,
                                                                         const float* __restrict__ mean, const float* __restrict__ var,
                                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                         bf16_t* __restrict__ rows, float hi_scale, float lo_scale,
                                                                         bf16_t* __restrict__ xhat) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= Mp) return;
  const int w = row & 127;
  const int s = (w >> 4) * 4 + (w & 3), b = (row >> 7) * 4 + ((w >> 2) & 3);
  const bool live = b < B && s < S;
  const int F4 = F >> 2;
  bf16_t* out = rows + (long)row * 2 * F;
  for (int j = lane; j < F4; j += 64) {
    uint2 o = make_uint2(0u, 0u), oh = make_uint2(0u, 0u);
    int w8 = 0, v8 = 0;
    if (live) {
      const float4 v = ((const float4*)(r + (long)row * F))[j];
      const float4 mu = ((const float4*)mean)[j], va = ((const float4*)var)[j], ga = ((const float4*)gamma)[j], be = ((const float4*)beta)[j];
      const float h[4] = {(v.x - mu.x) * rsqrtf(va.x + 1e-3f), (v.y - mu.y) * rsqrtf(va.y + 1e-3f), (v.z - mu.z) * rsqrtf(va.z + 1e-3f),
                          (v.w - mu.w) * rsqrtf(va.w + 1e-3f)};
      const float y[4] = {h[0] * ga.x + be.x, h[1] * ga.y + be.y, h[2] * ga.z + be.z, h[3] * ga.w + be.w};
      o = make_uint2(pack_f16x2_hw(y[0], y[1]), pack_f16x2_hw(y[2], y[3]));
      oh = make_uint2(pack_bf16x2_hw(h[0], h[1]), pack_bf16x2_hw(h[2], h[3]));
      const float yf[4] = {f16_to_f32((f16_t)(o.x & 0xffffu)), f16_to_f32((f16_t)(o.x >> 16)), f16_to_f32((f16_t)(o.y & 0xffffu)), f16_to_f32((f16_t)(o.y >> 16))};
      float a[4], c[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        a[k] = fminf(fmaxf(y[k] * hi_scale, -448.f), 448.f);
        c[k] = fminf(fmaxf((y[k] - yf[k]) * lo_scale, -448.f), 448.f);
      }
      w8 = __builtin_amdgcn_cvt_pk_fp8_f32(a[0], a[1], 0, false);
      w8 = __builtin_amdgcn_cvt_pk_fp8_f32(a[2], a[3], w8, true);
      v8 = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], 0, false);
      v8 = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], v8, true);
    }
    ((uint2*)out)[j] = o;
    ((int*)(out + F))[j] = w8;
    ((int*)(out + F))[F4 + j] = v8;
    if (xhat) ((uint2*)(xhat + (long)row * F))[j] = oh;
  }
}
extern "C" int evc_dbof_input_bn_apply_f16fp8(const float* r, int B, int S, int F, const float* mean, const float* var, const float* gamma,
                                              const float* beta, evc_f16* r_rows, int hi_exp, int lo_exp, evc_bf16* xhat, void* stream) {
  EVC_REQUIRE(B > 0 && S > 0 && S <= SP && F > 0 && F % 32 == 0 && r && r_rows && ((uintptr_t)r_rows % 16) == 0, EVC_ERR_BAD_SHAPE,
              "evc_dbof_input_bn_apply_f16fp8: bad args (F %% 32 == 0: 16-byte aligned row parts)");
  EVC_REQUIRE(hi_exp >= -30 && hi_exp <= 30 && lo_exp >= 0 && lo_exp <= 60, EVC_ERR_BAD_ARG, "evc_dbof_input_bn_apply_f16fp8: hi_exp=%d lo_exp=%d", hi_exp, lo_exp);
  const int Mp = dbof_padded_rows(B);
  hipLaunchKernelGGL(dbof_input_bn_apply_f16fp8_kernel, dim3((Mp + 3) / 4), dim3(256), 0, (hipStream_t)stream, r, Mp, B, S, F, mean, var, gamma,
                     beta, (bf16_t*)r_rows, ldexpf(1.0f, hi_exp), ldexpf(1.0f, lo_exp), xhat);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// K4: the cluster GEMM with the statistics / max-pool epilogue
// ---------------------------------------------------------------------------------------------------------------
#ifndef EVC_DBOF_ACT_POLICY
#define EVC_DBOF_ACT_POLICY 2     // the 268 MB bf16 tape as non-temporal stores (round 6; profiles/r06_dbof_act_policy_ab.txt, same box, alternating: the cluster kernel
                                   // in the cfg-4 step 317 -> 300 us, the step 1.70 -> 1.67 ms; sc1 write-through: no change; 0 = plain stores)
#endif
struct DbofPoolParams {
  bf16_t* act; long ld_act;        // [Mp][C] bf16 (training tape) or null
  float* part;                     // [2 * tiles_m][2][C] column partial sums or null (evaluation: moving statistics)
  const float* gamma;              // [C] cluster_bn scale: its sign selects max or min over the frames
  float* xsel;                     // [B][C] the selected pre-batch-norm activation of each (video, cluster)
  uint8_t* arg;                    // [B][C] its frame slot
  int B, S, C;
};

#ifdef EVC_DBOF_V2_LOOP
typedef TileCfg2<256, 1, 256, 2, 4, 5, true> CfgDbof;      // 256 frame rows (8 videos) x 256 clusters, 8 waves (2 x 4)
#else
struct CfgDbof : TileCfg3<256, 1, 256, 2, 4, 2> {};        // the same tile on two 64-wide K stages (gemm_core_v3.h: 128 KB = the epilogue's eight 16 KB transpose slices)
template <> struct is_v2<CfgDbof> { static constexpr bool value = true; };
template <> struct is_v3<CfgDbof> { static constexpr bool value = true; };
#endif

template <class Cfg, bool INIT, int EXTRA = 0>
__device__ __forceinline__ void dbof_mainloop(const GemmOperands& p, int m0, int u0, f32x4 (&acc)[Cfg::MI][1][Cfg::NI]) {
  constexpr int MODE = LOOP_DMA_FIRST | LOOP_NO_PRIO | EXTRA;      // (producer waves: 1.80 -> 2.30 ms per step here; these two: 1.80 -> 1.79)
  if constexpr (is_v3<Cfg>::value) gemm_mainloop_v3<Cfg, true, INIT, MODE>(p, m0, u0, lds_dyn, acc);
  else gemm_mainloop_v2<Cfg, true, INIT, MODE>(p, m0, u0, lds_dyn, acc);
}

// ---- epilogue of one output tile.  Transposed accumulators: lane 16g + l holds row mi*16 + l, columns ni*16 + 4g .. 4g+3 ----
// SLICE_ROWS = rows of a wave's transpose slice: Cfg::WM (the whole 128 x 64 sub-tile at once, in the idle ring at `scratch`: the one-tile
// kernels) or 32 (four passes through a 4 KB slice BEHIND the ring: the tile walk, whose ring is already filling with the next tile).
template <class Cfg, int SLICE_ROWS>
__device__ __forceinline__ void dbof_tile_epilogue(const GemmOperands& p, const DbofPoolParams& e, f32x4 (&acc)[Cfg::MI][1][Cfg::NI],
                                                   const int tm, const int m0, const int u0, char* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave / Cfg::WC, wc = wave % Cfg::WC;
  const int l = lane & 15, g = lane >> 4, j = l & 3;
  const int rbase = m0 + wr * Cfg::WM;                         // first row of this wave: a multiple of 128 = 4 videos
  const int b = (rbase >> 5) + (l >> 2);                       // this lane's video
  const int lim = b < e.B ? (e.S - j + 3) >> 2 : 0;            // accumulator blocks mi < lim hold sampled frames (slot mi*4 + j < S)
#ifdef EVC_ABLATE_DBOF_EPILOGUE      // timing ablation (profiles/r06_dbof_ablation.txt): the main loop alone, one store per lane keeps the accumulators alive
  {
    float keep = 0.f;                                            // (every accumulator feeds the never-taken store: no MFMA is dead code)
#pragma unroll
    for (int mi = 0; mi < Cfg::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < Cfg::NI; ++ni) keep += acc[mi][0][ni][0] + acc[mi][0][ni][1] + acc[mi][0][ni][2] + acc[mi][0][ni][3];
    if (e.S == 0x7fffffff) e.xsel[threadIdx.x] = keep;
    return;
  }
#endif
  if (e.act) {
    // bf16 activation for the backward pass.  The accumulator layout gives a lane 4 consecutive columns of 16 different
    // rows per block: stored directly, every wave-instruction touches 16 lines with 32 bytes each (measured: +0.21 ms on
    // the 0.31 ms kernel).  Each wave transposes its 128 x 64 sub-tile through its own LDS slice instead and writes whole
    // 128-byte lines, 16 bytes per lane.  (The caller has passed the barrier behind the main loop's last LDS reads.)
    // 128-byte rows, 16-byte chunk c of row r stored at chunk c ^ (r & 7) (round 4; before: 144-byte padded rows, whose ds_read_b128 lane
    // groups - rows r .. r+3 with chunk halves 0-3 / 4-7 / 4-7 / 0-3 - overlapped in 4 of 16 slots: the 9 % LDS bank conflicts of
    // profiles/r03_pmc_kernels_dbof.json).  Reads: slots (8 r + (c ^ r)) mod 16 of a group are 0-3 | 12-15 | 4-7 | 8-11: conflict-free; the 8-byte
    // writes of a 16-lane group (16 rows, one chunk) land 2-way (rows r and r + 8), which ds_write_b64 absorbs.
    constexpr int RS = Cfg::WU * 2;
    static_assert(RS == 128, "the swizzle is written for 64-column wave tiles");
    static_assert(SLICE_ROWS % 16 == 0 && Cfg::WM % SLICE_ROWS == 0, "whole accumulator blocks per pass");
    char* wl = scratch + wave * (SLICE_ROWS * RS);
    const int colw = u0 + wc * Cfg::WU;
#pragma unroll
    for (int r0 = 0; r0 < Cfg::WM; r0 += SLICE_ROWS) {          // (LDS operations of one wave execute in order: a pass's writes cannot overtake the previous pass's reads)
#pragma unroll
      for (int mi = r0 / 16; mi < (r0 + SLICE_ROWS) / 16; ++mi)
#pragma unroll
        for (int ni = 0; ni < Cfg::NI; ++ni) {
          const f32x4 v = acc[mi][0][ni];
          const int row = mi * 16 - r0 + l, chunk = ni * 2 + (g >> 1);
          *(uint2*)(wl + row * RS + ((chunk ^ (row & 7)) << 4) + ((g & 1) << 3)) = make_uint2(pack_bf16x2_hw(v[0], v[1]), pack_bf16x2_hw(v[2], v[3]));
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's own writes
      if (colw < e.C) {
#pragma unroll
        for (int it = 0; it < SLICE_ROWS / 8; ++it) {
          const int rl = it * 8 + (lane >> 3);
          const uint4 q = *(const uint4*)(wl + rl * RS + (((lane & 7) ^ (rl & 7)) << 4));
          const int row = rbase + r0 + rl;
          if (row < p.M) {
#if EVC_DBOF_ACT_POLICY      // A/B (round 6): the 268 MB bf16 tape leaves with a cache policy (1 = sc1 write-through, 2 = nt) instead of staying dirty in the XCD's L2
            const u32x4_t qv = {q.x, q.y, q.z, q.w};
            store16<EVC_DBOF_ACT_POLICY>(e.act, (uint32_t)(((long)row * e.ld_act + colw + (lane & 7) * 8) * 2), qv);
#else
            *(uint4*)(e.act + (long)row * e.ld_act + colw + (lane & 7) * 8) = q;
#endif
          }
        }
      }
    }
  }
#pragma unroll
  for (int ni = 0; ni < Cfg::NI; ++ni) {
    const int col = u0 + wc * Cfg::WU + ni * 16 + g * 4;
    if (col >= e.C) continue;                                  // wave-uniform: C % 64 == 0
    const float4 gm = *(const float4*)(e.gamma + col);
    const float gmr[4] = {gm.x, gm.y, gm.z, gm.w};
    float xs[4], ssum[4], ssq[4];
    uint32_t args = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float sgn = gmr[r] >= 0.f ? 1.f : -1.f;
      float sum = 0.f, sq = 0.f, best = -INFINITY;
      int bi = 0;
#pragma unroll
      for (int mi = 0; mi < Cfg::MI; ++mi) {
        if (mi < lim) {
          const float v = acc[mi][0][ni][r];
#ifndef EVC_ABLATE_DBOF_STATS
          sum += v;
          sq += v * v;
#endif
#ifndef EVC_ABLATE_DBOF_SELECT
          const float t = sgn * v;
          if (t > best) { best = t; bi = mi; }                 // strict: the first maximum wins
#else
          best = fmaxf(best, v);                               // (timing ablation: keeps the accumulators alive, no index / sign / tie logic)
#endif
        }
      }
      int sidx = bi * 4 + j;
      // the video's other three lanes (same quad): maximum, ties to the smaller frame slot
#ifndef EVC_ABLATE_DBOF_SELECT
      {
        const float ob = dpp_f<QP_XOR1>(best);
        const int oi = dpp_i<QP_XOR1>(sidx);
        if (ob > best || (ob == best && oi < sidx)) { best = ob; sidx = oi; }
      }
      {
        const float ob = dpp_f<QP_XOR2>(best);
        const int oi = dpp_i<QP_XOR2>(sidx);
        if (ob > best || (ob == best && oi < sidx)) { best = ob; sidx = oi; }
      }
#endif
      xs[r] = sgn * best;
      args |= (uint32_t)(sidx & 0xff) << (8 * r);
      ssum[r] = row16_sum(sum);                                // the wave's 128 rows (4 videos) of this column
      ssq[r] = row16_sum(sq);
    }
    if (j == 0 && lim > 0) {
      *(float4*)(e.xsel + (long)b * e.C + col) = make_float4(xs[0], xs[1], xs[2], xs[3]);
      *(uint32_t*)(e.arg + (long)b * e.C + col) = args;
    }
    if (e.part && l == 0) {
      float* pp = e.part + ((long)(tm * 2 + wr) * 2) * e.C + col;
      *(float4*)pp = make_float4(ssum[0], ssum[1], ssum[2], ssum[3]);
      *(float4*)(pp + e.C) = make_float4(ssq[0], ssq[1], ssq[2], ssq[3]);
    }
  }
}

// FP8: IEEE f16 operands with both operands' low-order corrections as e4m3 stages behind them (LOOP_FP8_TAIL; the "high" precision forward)
template <bool FP8>
__global__ __launch_bounds__(CfgDbof::NT) void dbof_cluster_pool_kernel(GemmOperands p, DbofPoolParams e, int tiles_m, int tiles_n) {
  typedef CfgDbof Cfg;
  const int nwg = tiles_m * tiles_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  int tm, tn;
  tile_of(id, tiles_m, tiles_n, tm, tn, 8);
  const int m0 = tm * Cfg::BM, u0 = tn * Cfg::BU;
  f32x4 acc[Cfg::MI][1][Cfg::NI];
  dbof_mainloop<Cfg, true, FP8 ? (LOOP_F16 | LOOP_FP8_TAIL) : 0>(p, m0, u0, acc);
  if (!FP8 && p.A1lo) {             // split-bf16 operands: + hi.lo + lo.hi (f32-operand accuracy, 3x the MFMA work)
    GemmOperands q = p;
    q.B = p.Blo;
    __syncthreads();
    dbof_mainloop<Cfg, false>(q, m0, u0, acc);
    q = p;
    q.A1 = p.A1lo;
    __syncthreads();
    dbof_mainloop<Cfg, false>(q, m0, u0, acc);
  }
  if (e.act) __syncthreads();                                  // every wave has read its last ring slot: the ring is the transpose scratch
  static_assert(8 * Cfg::WM * Cfg::WU * 2 <= Cfg::LDS_BYTES, "per-wave transpose slices must fit the ring");
  dbof_tile_epilogue<Cfg, Cfg::WM>(p, e, acc, tm, m0, u0, lds_dyn);
}

// Tile walk (round 5; plain bf16 operands): one workgroup per CU walks the row tiles of ONE column panel of W_c.  Once every wave has left
// the ring, the first two stages of the NEXT tile are issued (gemm_mainloop_v3 PHASE 1) and land under this tile's statistics / arg-max / tape
// epilogue, whose transpose goes through 4 KB per wave BEHIND the ring (four passes of 32 rows) - 7 of 8 ring fills leave the critical path.
// Panel tn lives on XCD tn % 8 (workgroup b runs on XCD b % 8): a column panel of W_c is fetched by one L2 instead of eight; the `walkers`
// workgroups of a panel start on different row tiles and the XCD's panels walk the rows in step, so a row panel of the frames is in that L2
// for all of them.
struct CfgDbofWalk : CfgDbof { static constexpr int RING_BYTES = CfgDbof::LDS_BYTES, LDS_BYTES = RING_BYTES + 8 * 32 * 128; };
template <> struct is_v2<CfgDbofWalk> { static constexpr bool value = true; };
template <> struct is_v3<CfgDbofWalk> { static constexpr bool value = true; };
static_assert(CfgDbofWalk::LDS_BYTES <= 160 * 1024, "ring + eight 4 KB transpose slices");

// ROWPIN (round 6, EVC_DBOF_PIN=rows): the transposed assignment - ROW tile tm lives on XCD tm % 8 and a workgroup walks the COLUMN panels of its
// row tile: an XCD streams W_c (19 MB) instead of all frame rows (38 MB); panels_per_xcd then counts row tiles per XCD.
template <bool ROWPIN>
__global__ __launch_bounds__(CfgDbof::NT) void dbof_cluster_pool_walk_kernel(GemmOperands p, DbofPoolParams e, int tiles_m, int tiles_n,
                                                                             int panels_per_xcd, int walkers) {
  typedef CfgDbof Cfg;
  constexpr int MODE = LOOP_DMA_FIRST | LOOP_NO_PRIO;
  const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
  int tn = ROWPIN ? i / panels_per_xcd : (i % panels_per_xcd) * 8 + xcd;
  int tm = ROWPIN ? (i % panels_per_xcd) * 8 + xcd : i / panels_per_xcd;
  if (tn >= tiles_n || tm >= tiles_m) return;                   // (workgroup-uniform)
  f32x4 acc[Cfg::MI][1][Cfg::NI];
  gemm_mainloop_v3<Cfg, true, true, MODE, 1>(p, tm * Cfg::BM, tn * Cfg::BU, lds_dyn, acc);
  for (; tm < tiles_m && tn < tiles_n; (ROWPIN ? tn : tm) += walkers) {
    const int m0 = tm * Cfg::BM, u0 = tn * Cfg::BU;
    gemm_mainloop_v3<Cfg, true, true, MODE, 2>(p, m0, u0, lds_dyn, acc);
    __syncthreads();                                             // every wave has read its last ring slot
    const int tm2 = ROWPIN ? tm : tm + walkers, tn2 = ROWPIN ? tn + walkers : tn;
    if (tm2 < tiles_m && tn2 < tiles_n) gemm_mainloop_v3<Cfg, true, true, MODE, 1>(p, tm2 * Cfg::BM, tn2 * Cfg::BU, lds_dyn, acc);
    dbof_tile_epilogue<Cfg, 32>(p, e, acc, tm, m0, u0, lds_dyn + CfgDbofWalk::RING_BYTES);
  }
}

extern "C" int evc_dbof_cluster_pool_fwd(const evc_bf16* r_bn, const evc_bf16* r_bn_lo, const evc_bf16* wT, const evc_bf16* wT_lo,
                                         int B, int S, int F, int C, const float* gamma, evc_bf16* act, float* part, float* xsel,
                                         uint8_t* arg, void* stream) {
  EVC_REQUIRE(B > 0 && S > 0 && S <= SP && F > 0 && F % CfgDbof::BK == 0 && C > 0 && C % 64 == 0, EVC_ERR_BAD_SHAPE,
              "evc_dbof_cluster_pool_fwd: needs iterations <= %d, F %% 64 == 0, clusters %% 64 == 0 (S=%d F=%d C=%d)", SP, S, F, C);
  EVC_REQUIRE(ring_operand_ok(((long)B + 3) / 4 * 4 * SP, F) && ring_operand_ok(C, F), EVC_ERR_BAD_SHAPE,
              "evc_dbof_cluster_pool_fwd: the frame matrix or the cluster weights span 4 GiB or more (B=%d F=%d C=%d)", B, F, C);
  EVC_REQUIRE((r_bn_lo != nullptr) == (wT_lo != nullptr), EVC_ERR_BAD_ARG, "evc_dbof_cluster_pool_fwd: both low halves or none");
  const int Mp = dbof_padded_rows(B);
  GemmOperands p;
  p.A1 = r_bn; p.lda1 = F; p.nk1 = F / CfgDbof::BK; p.A2 = r_bn; p.lda2 = F; p.nk2 = 0;
  p.B = wT; p.ldb = F; p.group_stride = 0; p.M = Mp; p.Nu = C;
  p.A1lo = r_bn_lo; p.A2lo = nullptr; p.Blo = wT_lo;
  DbofPoolParams e{act, (long)C, part, gamma, xsel, arg, B, S, C};
  const int tm = ceil_div(Mp, CfgDbof::BM), tn = ceil_div(C, CfgDbof::BU);
#ifndef EVC_DBOF_V2_LOOP
  // tile walk: plain bf16 operands and enough tiles to give 256 workgroups several each (EVC_DBOF_WALK=0: one tile per workgroup, A/B)
  const char* wenv = getenv("EVC_DBOF_WALK");                    // (read per call: the tests switch it; 2 = also below 512 tiles)
  const int walk_on = wenv ? atoi(wenv) : 1;
  const int ppx = ceil_div(tn, 8);                               // column panels per XCD
  const char* penv = getenv("EVC_DBOF_PIN");                     // rows: row tiles pinned to XCDs, workgroups walk the column panels (A/B, round 6)
  const int rpx = ceil_div(tm, 8);                               // row tiles per XCD
  if (walk_on && !r_bn_lo && penv && penv[0] == 'r' && rpx <= 32 && ((long)tm * tn >= 512 || walk_on == 2)) {
    int walkers = 32 / rpx;
    walkers = walkers < tn ? walkers : tn;
    launch_cfg<CfgDbofWalk>(dbof_cluster_pool_walk_kernel<true>, 8 * rpx * walkers, (hipStream_t)stream, p, e, tm, tn, rpx, walkers);
    EVC_LAUNCH_CHECK();
    return EVC_OK;
  }
  if (walk_on && !r_bn_lo && ppx <= 32 && ((long)tm * tn >= 512 || walk_on == 2)) {
    int walkers = 32 / ppx;                                      // 32 workgroups (= CUs) per XCD
    walkers = walkers < tm ? walkers : tm;
    launch_cfg<CfgDbofWalk>(dbof_cluster_pool_walk_kernel<false>, 8 * ppx * walkers, (hipStream_t)stream, p, e, tm, tn, ppx, walkers);
    EVC_LAUNCH_CHECK();
    return EVC_OK;
  }
#endif
  launch_cfg<CfgDbof>(dbof_cluster_pool_kernel<false>, tm * tn, (hipStream_t)stream, p, e, tm, tn);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

#ifndef EVC_DBOF_V2_LOOP
// The cluster GEMM on f16 operands with both operands' low-order corrections as e4m3 operands behind the f16 stages of the same launch
// (evc_gemm_nt_f16_fp8's arithmetic with this kernel's epilogue): r_rows from evc_dbof_input_bn_apply_f16fp8 (rows of 4F bytes), wT16 [C][F]
// f16, wT8 [C][2F] = [e4m3((W - f16(W)) 2^w_lo_exp) | e4m3(W 2^w_hi_exp)] (evc_cast_f32_to_fp8_lo, hi_cols = F); scale_exp = -(x_hi_exp +
// w_lo_exp) = -(x_lo_exp + w_hi_exp).  2 x the MFMA time of the bf16 product instead of the split-bf16 form's 3 x.  F % 128 == 0.
extern "C" int evc_dbof_cluster_pool_fwd_f16fp8(const evc_f16* r_rows, const evc_f16* wT16, const uint8_t* wT8, int scale_exp,
                                                int B, int S, int F, int C, const float* gamma, evc_bf16* act, float* part, float* xsel,
                                                uint8_t* arg, void* stream) {
  EVC_REQUIRE(B > 0 && S > 0 && S <= SP && F >= 256 && F % 128 == 0 && C > 0 && C % 64 == 0, EVC_ERR_BAD_SHAPE,
              "evc_dbof_cluster_pool_fwd_f16fp8: needs iterations <= %d, F %% 128 == 0 (>= 256), clusters %% 64 == 0 (S=%d F=%d C=%d)", SP, S, F, C);
  EVC_REQUIRE(r_rows && wT16 && wT8 && ((uintptr_t)r_rows % 16) == 0 && ((uintptr_t)wT16 % 16) == 0 && ((uintptr_t)wT8 % 16) == 0, EVC_ERR_BAD_ALIGN,
              "evc_dbof_cluster_pool_fwd_f16fp8: 16-byte aligned operands");
  EVC_REQUIRE(ring_operand_ok(((long)B + 3) / 4 * 4 * SP, 2L * F) && ring_operand_ok(C, F), EVC_ERR_BAD_SHAPE,
              "evc_dbof_cluster_pool_fwd_f16fp8: the frame matrix or the cluster weights span 4 GiB or more (B=%d F=%d C=%d)", B, F, C);
  const int Mp = dbof_padded_rows(B);
  GemmOperands p;
  p.A1 = (const bf16_t*)r_rows; p.lda1 = 2L * F; p.nk1 = F / 64; p.A2 = p.A1; p.lda2 = p.lda1; p.nk2 = 0;
  p.B = (const bf16_t*)wT16; p.ldb = F; p.group_stride = 0; p.M = Mp; p.Nu = C;
  p.A1lo = p.A2lo = p.Blo = nullptr;
  p.A3 = (const uint8_t*)r_rows + 2L * F; p.lda3 = 4L * F; p.nk3 = 2 * F / 128; p.A4 = p.A3; p.lda4 = p.lda3; p.nk4 = 0;
  p.B8 = wT8; p.ldb8 = 2L * F; p.scale8_exp = scale_exp;
  DbofPoolParams e{act, (long)C, part, gamma, xsel, arg, B, S, C};
  const int tm = ceil_div(Mp, CfgDbof::BM), tn = ceil_div(C, CfgDbof::BU);
  launch_cfg<CfgDbof>(dbof_cluster_pool_kernel<true>, tm * tn, (hipStream_t)stream, p, e, tm, tn);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}
#else
// (A/B build on the 32-wide K stages: the e4m3 tail exists in the 64-wide ring loop only.  The symbol stays exported - the ctypes
//  table binds every entry point of include/evc.h at import - and says so when called.)
extern "C" int evc_dbof_cluster_pool_fwd_f16fp8(const evc_f16*, const evc_f16*, const uint8_t*, int, int, int, int, int, const float*, evc_bf16*, float*,
                                                float*, uint8_t*, void*) {
  evc_set_error("evc_dbof_cluster_pool_fwd_f16fp8: this library was built with -DEVC_DBOF_V2_LOOP (no e4m3 stages); set EVC_HIGH_FP8_LO=0");
  return EVC_ERR_UNSUPPORTED_ARCH;
}
#endif

// ---------------------------------------------------------------------------------------------------------------
// K6: pooled = relu6(gamma * (x_sel - mean) * rsqrt(var + eps) + beta)   [B][C]
// ---------------------------------------------------------------------------------------------------------------
__global__ void dbof_pool_finish_kernel(const float* __restrict__ xsel, long n, int C, const float* __restrict__ mean,
                                        const float* __restrict__ var, const float* __restrict__ gamma, const float* __restrict__ beta,
                                        float* __restrict__ pf, bf16_t* __restrict__ pb, bf16_t* __restrict__ pb_lo) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    float y = (xsel[i] - mean[c]) * rsqrtf(var[c] + 1e-3f) * gamma[c] + beta[c];
    y = fminf(fmaxf(y, 0.f), 6.f);
    pf[i] = y;
    const bf16_t h = f32_to_bf16(y);
    if (pb) pb[i] = h;
    if (pb_lo) pb_lo[i] = f32_to_bf16(y - bf16_to_f32(h));
  }
}
extern "C" int evc_dbof_pool_finish(const float* xsel, int B, int C, const float* mean, const float* var, const float* gamma,
                                    const float* beta, float* pooled_f32, evc_bf16* pooled_bf16, evc_bf16* pooled_lo, void* stream) {
  EVC_REQUIRE(B > 0 && C > 0 && xsel && pooled_f32, EVC_ERR_BAD_SHAPE, "evc_dbof_pool_finish: bad args");
  const long n = (long)B * C;
  long nb = (n + 255) / 256;
  nb = nb > 4096 ? 4096 : nb;
  hipLaunchKernelGGL(dbof_pool_finish_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, xsel, n, C, mean, var, gamma, beta,
                     pooled_f32, pooled_bf16, pooled_lo);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// K10: backward of FramePooling('max') + relu6 + cluster batch-norm, in place on the bf16 activation:
//   d[b,s,c]  = dpooled[b,c] if s == arg[b,c] and 0 < pooled[b,c] < 6, else 0
//   dact      = gamma * rstd * (d - S1/R - xhat * S2/R),   xhat = (act - mean) * rstd,
// S1 = sum d, S2 = sum d * xhat over all R_total sampled frames (ws, from the [B][C] arrays alone, all-reduced under
// data parallelism); empty frame slots get 0.  A workgroup owns one video x 2048 clusters: its per-column constants
// and the video's (d, arg) stay in registers while it walks the video's 32 rows.
// ---------------------------------------------------------------------------------------------------------------
#ifndef EVC_DBOF_DACT_ROWS
#define EVC_DBOF_DACT_ROWS 32
#endif
#ifndef EVC_DBOF_DACT_NT
#define EVC_DBOF_DACT_NT 1
#endif
__global__ __launch_bounds__(256) void dbof_dact_kernel(bf16_t* __restrict__ act, const float* __restrict__ dpooled,
                                                        const float* __restrict__ pooled, const uint8_t* __restrict__ arg,
                                                        const float* __restrict__ mean, const float* __restrict__ var,
                                                        const float* __restrict__ gamma, const double* __restrict__ ws, int R_total,
                                                        int B, int S, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int b = blockIdx.y;
  const int c0 = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (c0 >= C) return;
  if (b == 0) {                                        // cluster_bn's own gradients are the two sums
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (dbeta) dbeta[c0 + i] = (float)ws[c0 + i];
      if (dgamma) dgamma[c0 + i] = (float)ws[C + c0 + i];
    }
  }
  float mu[8], rs[8], a[8], k1[8], k2[8], d[8];
  int ar[8];
  const float invR = 1.0f / (float)R_total;
  const bool live = b < B;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = c0 + i;
    mu[i] = mean[c];
    rs[i] = rsqrtf(var[c] + 1e-3f);
    a[i] = gamma[c] * rs[i];
    k1[i] = (float)ws[c] * invR;
    k2[i] = (float)ws[C + c] * invR;
    d[i] = 0.f;
    ar[i] = -1;
    if (live) {
      const float y = pooled[(long)b * C + c];
      if (y > 0.f && y < 6.f) { d[i] = dpooled[(long)b * C + c]; ar[i] = arg[(long)b * C + c]; }
    }
  }
  // The walk is in place: written as load -> store per row, hipcc keeps each row's load behind the previous row's store (one 16-byte load in flight per lane:
  // 3.5 TB/s).  DACT_ROWS rows are loaded before the first of them is stored (round 6; all 32 = the whole video in registers, nt loads: 0.154 -> 0.126 ms, 4.3 TB/s, profiles/r06_dbof_dact_ab.txt).
  constexpr int DACT_ROWS = EVC_DBOF_DACT_ROWS;
  static_assert(SP % DACT_ROWS == 0, "whole row groups");
  for (int s0 = 0; s0 < SP; s0 += DACT_ROWS) {
    uint4 q[DACT_ROWS];
#pragma unroll
    for (int j = 0; j < DACT_ROWS; ++j) {
      q[j] = make_uint4(0u, 0u, 0u, 0u);
      if (live && s0 + j < S) {
#if EVC_DBOF_DACT_NT
        const u32x4_t t = __builtin_nontemporal_load((const u32x4_t*)(act + dbof_row(b, s0 + j) * C + c0));   // the tape is read once
#else
        const u32x4_t t = *(const u32x4_t*)(act + dbof_row(b, s0 + j) * C + c0);
#endif
        q[j] = make_uint4(t[0], t[1], t[2], t[3]);
      }
    }
#pragma unroll
    for (int j = 0; j < DACT_ROWS; ++j) {
      const int s = s0 + j;
      uint4 o = make_uint4(0u, 0u, 0u, 0u);
      if (live && s < S) {
        const uint32_t w[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
        uint32_t r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float x0 = __uint_as_float(w[i] << 16), x1 = __uint_as_float(w[i] & 0xffff0000u);
          const float h0 = (x0 - mu[2 * i]) * rs[2 * i], h1 = (x1 - mu[2 * i + 1]) * rs[2 * i + 1];
          const float g0 = a[2 * i] * ((ar[2 * i] == s ? d[2 * i] : 0.f) - k1[2 * i] - h0 * k2[2 * i]);
          const float g1 = a[2 * i + 1] * ((ar[2 * i + 1] == s ? d[2 * i + 1] : 0.f) - k1[2 * i + 1] - h1 * k2[2 * i + 1]);
          r[i] = pack_bf16x2_hw(g0, g1);
        }
        o = make_uint4(r[0], r[1], r[2], r[3]);
      }
      *(uint4*)(act + dbof_row(b, s) * C + c0) = o;
    }
  }
}
extern "C" int evc_dbof_dact(evc_bf16* act, const float* dpooled, const float* pooled, const uint8_t* arg, const float* mean,
                             const float* var, const float* gamma, const double* ws, int R_total, int B, int S, int C, float* dgamma,
                             float* dbeta, void* stream) {
  EVC_REQUIRE(B > 0 && S > 0 && S <= SP && C > 0 && C % 8 == 0 && R_total >= B * S, EVC_ERR_BAD_SHAPE, "evc_dbof_dact: bad args");
  const int Bp = ((B + 3) / 4) * 4;                             // the padded rows of the last 128-row block are zeroed too
  hipLaunchKernelGGL(dbof_dact_kernel, dim3((C / 8 + 255) / 256, Bp), dim3(256), 0, (hipStream_t)stream, act, dpooled, pooled, arg, mean,
                     var, gamma, ws, R_total, B, S, C, dgamma, dbeta);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// K12: G = dact^T . xhat arrives as `nslab` split-K partial slabs [nslab][C][F] f32 (evc_gemm_tn_slabs).  With
// r_bn = gamma_in * xhat + beta_in and the column sums of dact zero (the batch-norm backward's output sums to zero
// over the batch):   dWc[c][f] = gamma_in[f] * G[c][f]        (cluster weights, stored [C][F])
//                    dgamma_in[f] = sum_c Wc[c][f] * G[c][f]  (= sum_r (dact . Wc)[r][f] * xhat[r][f], without forming dact . Wc)
//                    dbeta_in = 0
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dbof_wgrad_finish_kernel(const float* __restrict__ slabs, int nslab, long slab_stride, int C, int F,
                                                                const float* __restrict__ W, const float* __restrict__ gamma_in,
                                                                float* __restrict__ dW, float* __restrict__ part, int rows_per_block) {
  // a block walks rows_per_block rows of [C][F]; thread t owns the float4 columns t, t + 256, ... (consecutive threads read
  // consecutive 16-byte pieces of a row); its column sums of W * G go to part[block][F] (summed afterwards: no atomics)
  const int F4 = F >> 2;
  const int c0 = blockIdx.x * rows_per_block, c1 = min(C, c0 + rows_per_block);
  for (int f4 = threadIdx.x; f4 < F4; f4 += 256) {
    const float4 ga = ((const float4*)gamma_in)[f4];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int c = c0; c < c1; ++c) {
      const long o = (long)c * F4 + f4;
      float4 gsum = ((const float4*)slabs)[o];
      for (int s = 1; s < nslab; ++s) {
        const float4 t = ((const float4*)(slabs + s * slab_stride))[o];
        gsum.x += t.x; gsum.y += t.y; gsum.z += t.z; gsum.w += t.w;
      }
      const float4 w = ((const float4*)W)[o];
      acc.x += w.x * gsum.x; acc.y += w.y * gsum.y; acc.z += w.z * gsum.z; acc.w += w.w * gsum.w;
      ((float4*)dW)[o] = make_float4(ga.x * gsum.x, ga.y * gsum.y, ga.z * gsum.z, ga.w * gsum.w);
    }
    ((float4*)(part + (long)blockIdx.x * F))[f4] = acc;
  }
}
__global__ __launch_bounds__(1024) void rowsum_partials_kernel(const float* __restrict__ part, int P, int F, float* __restrict__ out) {
  __shared__ float sh[16][64];                          // 64 columns x 16 row phases, phase sums added in phase order
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int f = blockIdx.x * 64 + tx;
  float s = 0.f;
  if (f < F) {
#pragma unroll 8
    for (int p = ty; p < P; p += 16) s += part[(long)p * F + f];
  }
  sh[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && f < F) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sh[i][tx];
    out[f] = t;
  }
}
static constexpr int WGRAD_ROWS_PER_BLOCK = 8;
extern "C" int evc_dbof_wgrad_finish(const float* slabs, int nslab, int C, int F, const float* W, const float* gamma_in, float* dW,
                                     float* dgamma_in, float* dbeta_in, float* part_ws, void* stream) {
  EVC_REQUIRE(nslab > 0 && C > 0 && F > 0 && F % 4 == 0 && part_ws, EVC_ERR_BAD_SHAPE, "evc_dbof_wgrad_finish: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (dbeta_in) EVC_CHECK_HIP(hipMemsetAsync(dbeta_in, 0, sizeof(float) * F, st));
  const int nb = (C + WGRAD_ROWS_PER_BLOCK - 1) / WGRAD_ROWS_PER_BLOCK;
  hipLaunchKernelGGL(dbof_wgrad_finish_kernel, dim3(nb), dim3(256), 0, st, slabs, nslab, (long)C * F, C, F, W, gamma_in, dW, part_ws,
                     WGRAD_ROWS_PER_BLOCK);
  hipLaunchKernelGGL(rowsum_partials_kernel, dim3((F + 63) / 64), dim3(1024), 0, st, part_ws, nb, F, dgamma_in);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}
