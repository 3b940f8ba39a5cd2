// Clip norm of a MoE weight gradient WITHOUT a pass over the weights (replaces pass 1 of evc_moe_grad_update, one process).
//
// The gradient of a MoE weight matrix W [V][K] is g = A^T . X with the batch-row factors A = dlogits [R][V] and X = the head's
// input [R][K] (bf16, R = batch rows: rank <= R).  What per-tensor clip_by_norm needs (cs/train.py:329-334, slim.learning
// create_train_op with clip_gradient_norm; the l2 regulariser's gradient l2 W is part of the clipped gradient,
// cs/video_level_models.py:428,434) is
//     |g + l2 W|^2 = |g|^2 + 2 l2 <g, W> + l2^2 |W|^2
// and every term has a form that never touches the 58 M / 39 M weights:
//     |g|^2   = < A A^T , X X^T >_F                      two R x R Gram matrices (1.9 + 0.5 GFLOP at R = 256)
//     <g, W>  = < A , X W^T >_F = < A , logits - bias >  the forward logits the head already holds
//     |W|^2   = carried from the epilogue of the previous update (evc_moe_grad_update_apply writes sum of the NEW weights squared)
// Pass 1 streamed W (232 MB for the gates matrix) and recomputed every gradient tile: 0.36 ms per training step on both towers.
// Everything here is summed in a fixed order (K slabs stored plainly, per-block partials, one combining thread): run-to-run identical.
#include "evc_common.h"

// G_s[i][j] = sum over the k steps of slab s of A[i][k] A[j][k].  One wave = one 32 x 32 tile of one slab, four 16x16x32 MFMAs per
// k step with the fragments straight from global memory (16 rows x 64 B per load instruction; the factors are L2 / MALL resident:
// 7 MB).  grid (tiles, ceil(S/4)), 256 threads: wave w of block (t, y) owns slab 4y + w.
__global__ __launch_bounds__(256) void gram_slabs_kernel(const bf16_t* __restrict__ A, long lda, int R, int nk, int S, float* __restrict__ slabs) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int s = blockIdx.y * 4 + wave;
  if (s >= S) return;
  const int nt = R / 32;
  const int ti = blockIdx.x / nt, tj = blockIdx.x % nt;
  const int per = (nk + S - 1) / S;
  const int k0 = s * per, k1 = min(nk, k0 + per);
  const int l = lane & 15, g = lane >> 4;
  const bf16_t* ai0 = A + (long)(ti * 32 + l) * lda + g * 8;
  const bf16_t* ai1 = ai0 + 16 * lda;
  const bf16_t* aj0 = A + (long)(tj * 32 + l) * lda + g * 8;
  const bf16_t* aj1 = aj0 + 16 * lda;
  f32x4 c00 = {0.f, 0.f, 0.f, 0.f}, c01 = c00, c10 = c00, c11 = c00;
#pragma unroll 4
  for (int k = k0; k < k1; ++k) {
    const long o = (long)k * 32;
    const bf16x8 a0 = *(const bf16x8*)(ai0 + o), a1 = *(const bf16x8*)(ai1 + o);
    const bf16x8 b0 = *(const bf16x8*)(aj0 + o), b1 = *(const bf16x8*)(aj1 + o);
    c00 = mfma16<false>(a0, b0, c00);
    c01 = mfma16<false>(a0, b1, c01);
    c10 = mfma16<false>(a1, b0, c10);
    c11 = mfma16<false>(a1, b1, c11);
  }
  // accumulator layout: lane 16 g + l holds rows 4 g .. 4 g + 3 (of the first operand's 16), column l (of the second's)
  float* out = slabs + (long)s * R * R;
  const int i0 = ti * 32 + g * 4, j0 = tj * 32 + l;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    out[(long)(i0 + r) * R + j0] = c00[r];
    out[(long)(i0 + r) * R + j0 + 16] = c01[r];
    out[(long)(i0 + 16 + r) * R + j0] = c10[r];
    out[(long)(i0 + 16 + r) * R + j0 + 16] = c11[r];
  }
}

extern "C" int evc_gram_slabs(const evc_bf16* A, int64_t lda, int R, int Kc, int S, float* slabs, void* stream) {
  EVC_REQUIRE(R > 0 && R % 32 == 0 && R <= 1024 && Kc > 0 && Kc % 32 == 0 && S >= 1 && S <= Kc / 32, EVC_ERR_BAD_SHAPE,
              "evc_gram_slabs: R=%d (%%32, <= 1024) Kc=%d (%%32) S=%d (1..Kc/32)", R, Kc, S);
  EVC_REQUIRE(lda % 8 == 0 && lda >= Kc && ((uintptr_t)A % 16) == 0 && slabs != nullptr, EVC_ERR_BAD_ALIGN,
              "evc_gram_slabs: rows of Kc bf16 at a 16-byte aligned stride (lda=%ld)", (long)lda);
  const int nk = Kc / 32, per = (nk + S - 1) / S;
  EVC_REQUIRE((long)per * (S - 1) < nk, EVC_ERR_BAD_SHAPE, "evc_gram_slabs: S=%d leaves an empty slab at Kc=%d", S, Kc);
  hipLaunchKernelGGL(gram_slabs_kernel, dim3((R / 32) * (R / 32), (S + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)A, (long)lda, R, nk, S, slabs);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

__device__ __forceinline__ float block_sum_256(float v, float* sh) {      // fixed order: butterflies inside a wave, the 4 wave totals in order
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// blocks [0, nbf): part[b] = sum over this block's elements of (sum_s GA_s) (sum_s GX_s)          (one element per thread at R = 256)
// blocks [nbf, nbf + nbd): part[b] = sum over segment (b - nbf) % 4 of row (b - nbf) / 4 of A[r][v] (logits[r][v] - bias[v])
__global__ __launch_bounds__(256) void moe_norm_partials_kernel(const float* __restrict__ ga, int SA, const float* __restrict__ gx, int SX, int R,
                                                                const bf16_t* __restrict__ A, long lda, const float* __restrict__ logits, long ldl,
                                                                const float* __restrict__ bias, int B, int V, int nbf, float* __restrict__ part) {
  __shared__ float sh[4];
  float acc = 0.f;
  const long n = (long)R * R;
  if ((int)blockIdx.x < nbf) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)nbf * 256) {
      float a = 0.f, x = 0.f;
#pragma unroll 8
      for (int s = 0; s < SA; ++s) a += ga[(long)s * n + i];
#pragma unroll 8
      for (int s = 0; s < SX; ++s) x += gx[(long)s * n + i];
      acc += a * x;
    }
  } else {
    const int b = blockIdx.x - nbf, r = b >> 2, q = b & 3;
    const int vq = ((V + 3) / 4 + 7) / 8 * 8;                     // columns per segment (multiple of 8: 16-byte aligned bf16 runs)
    const int v0 = q * vq, v1 = min(V, v0 + vq);
    const bf16_t* ar = A + (long)r * lda;
    const float* lr = logits + (long)r * ldl;
    for (int v = v0 + threadIdx.x; v < v1; v += 256) acc += bf16_to_f32(ar[v]) * (lr[v] - (bias ? bias[v] : 0.f));
  }
  acc = block_sum_256(acc, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// sums[0] += |g|^2 + 2 l2 <g, W> + l2^2 |W|^2 ; sums[1] += |W|^2 (the row layout of evc_grad_sqnorm / pass 1).  One wave; every lane
// sums a strided share of the partials in index order, the 64 lane totals meet in a butterfly: fixed order.
__global__ __launch_bounds__(64) void moe_norm_combine_kernel(const float* __restrict__ part, int nbf, int nbd, float l2, const float* __restrict__ wsq,
                                                              float* __restrict__ sums) {
  double gg = 0.0, gw = 0.0;
  for (int b = threadIdx.x; b < nbf; b += 64) gg += (double)part[b];
  for (int b = threadIdx.x; b < nbd; b += 64) gw += (double)part[nbf + b];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    gg += __shfl_xor(gg, o, 64);
    gw += __shfl_xor(gw, o, 64);
  }
  if (threadIdx.x == 0) {
    const double w2 = (double)wsq[0];
    const double tot = gg + 2.0 * (double)l2 * gw + (double)l2 * (double)l2 * w2;
    sums[0] += (float)(tot > 0.0 ? tot : 0.0);
    sums[1] += (float)w2;
  }
}

extern "C" int evc_moe_grad_norms(const float* gram_a, int SA, const float* gram_x, int SX, int R, const evc_bf16* dlogits, int64_t ld_dlogits,
                                  const float* logits, int64_t ld_logits, const float* bias, int B, int V, float l2_coeff, const float* wsq,
                                  float* part_ws, float* sums, void* stream) {
  EVC_REQUIRE(gram_a && gram_x && dlogits && logits && wsq && part_ws && sums, EVC_ERR_BAD_ARG, "evc_moe_grad_norms: NULL argument");
  EVC_REQUIRE(R > 0 && R % 32 == 0 && SA >= 1 && SX >= 1 && B > 0 && B <= R && V > 0 && ld_dlogits >= V && ld_logits >= V, EVC_ERR_BAD_SHAPE,
              "evc_moe_grad_norms: R=%d SA=%d SX=%d B=%d V=%d", R, SA, SX, B, V);
  EVC_REQUIRE(B <= 512, EVC_ERR_BAD_SHAPE, "evc_moe_grad_norms: B=%d (at most 512 batch rows; part_ws holds 256 + 4 B floats)", B);
  const int nbf = (int)(((long)R * R / 256) < 256 ? ((long)R * R / 256) : 256), nbd = 4 * B;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(moe_norm_partials_kernel, dim3(nbf + nbd), dim3(256), 0, st, gram_a, SA, gram_x, SX, R, (const bf16_t*)dlogits, (long)ld_dlogits,
                     logits, (long)ld_logits, bias, B, V, nbf, part_ws);
  hipLaunchKernelGGL(moe_norm_combine_kernel, dim3(1), dim3(64), 0, st, (const float*)part_ws, nbf, nbd, l2_coeff, wsq, sums);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}
