// NetVLAD aggregation kernels (EXTENSION - SURVEY.md 8f row 4: the reference's NetVLADModel is an empty stub,
// cs/frame_level_models.py:341-347, so there is no reference math; the oracle is oracle/model_math.py::netvlad_fwd/bwd,
// checked against finite differences).  The GEMM-shaped parts of the tower (cluster assignment, hidden layer, their
// gradients) run on the library's NT / TN kernels; these are the f32 pieces in between:
//
//   evc_netvlad_softmax_fwd/bwd      a = softmax_k(cluster_bn(act)) per sampled frame, and its reverse mode
//   evc_netvlad_aggregate_fwd/bwd    V[b][k][:] = sum_s a[b,s,k] * (x_bn[b,s,:] - c2[k][:]); da, dx_bn from dV
//   evc_netvlad_dcenters             dc2[k][:] = - sum_b asum[b][k] * dV[b][k][:]
//   evc_netvlad_normalize_fwd/bwd    intra-normalisation per (video, cluster) then l2 over the whole descriptor
//
// Layouts: sampled frames row-major [B*S][..] (row b*S+s); V / dV / Y [B][K][F] (cluster-major: the hidden layer's
// weight rows are permuted accordingly, towers.NetVladTower converts to the TF order f*K+k in state_dict()).
#include "evc_common.h"

static inline int nv_grid(long n) { long nb = (n + 255) / 256; return (int)(nb < 1 ? 1 : (nb < 8192 ? nb : 8192)); }

// ---- softmax over the clusters of the batch-normalised assignment logits: one wave per sampled frame ----
__global__ __launch_bounds__(256) void netvlad_softmax_fwd_kernel(const float* __restrict__ act, int R, int K,
                                                                  const float* __restrict__ mean, const float* __restrict__ var,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  float* __restrict__ a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  float y[16];                                   // K <= 1024
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k = lane + 64 * i;
    y[i] = -INFINITY;
    if (k < K) {
      y[i] = (act[(long)row * K + k] - mean[k]) * rsqrtf(var[k] + 1e-3f) * gamma[k] + beta[k];
      mx = fmaxf(mx, y[i]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k = lane + 64 * i;
    if (k < K) { y[i] = __expf(y[i] - mx); sum += y[i]; }
  }
  const float inv = 1.0f / wave_sum(sum);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k = lane + 64 * i;
    if (k < K) a[(long)row * K + k] = y[i] * inv;
  }
}
extern "C" int evc_netvlad_softmax_fwd(const float* act, int R, int K, const float* mean, const float* var, const float* gamma,
                                       const float* beta, float* a, void* stream) {
  EVC_REQUIRE(R > 0 && K > 0 && K <= 1024, EVC_ERR_BAD_SHAPE, "evc_netvlad_softmax_fwd: 1 <= clusters <= 1024 (K=%d)", K);
  hipLaunchKernelGGL(netvlad_softmax_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, act, R, K, mean, var, gamma, beta, a);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// dz = a * (da - sum_k a * da): gradient wrt the batch-normalised logits
__global__ __launch_bounds__(256) void netvlad_softmax_bwd_kernel(const float* __restrict__ a, const float* __restrict__ da, int R, int K,
                                                                  float* __restrict__ dz) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  float dot = 0.f;
  for (int k = lane; k < K; k += 64) dot += a[(long)row * K + k] * da[(long)row * K + k];
  dot = wave_sum(dot);
  for (int k = lane; k < K; k += 64) dz[(long)row * K + k] = a[(long)row * K + k] * (da[(long)row * K + k] - dot);
}
extern "C" int evc_netvlad_softmax_bwd(const float* a, const float* da, int R, int K, float* dz, void* stream) {
  EVC_REQUIRE(R > 0 && K > 0, EVC_ERR_BAD_SHAPE, "evc_netvlad_softmax_bwd: bad shape");
  hipLaunchKernelGGL(netvlad_softmax_bwd_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, da, R, K, dz);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// ---- aggregation.  x_bn is recomputed from the sampled frames r and input_bn's statistics (no f32 copy of it is kept).
// grid (B, K / KT): a workgroup accumulates KT clusters of one video; thread t owns the float4 feature columns t, t+256, ...
static constexpr int NV_KT = 8;
__global__ __launch_bounds__(256) void netvlad_aggregate_fwd_kernel(const float* __restrict__ a, const float* __restrict__ r, int S, int K,
                                                                    int F, const float* __restrict__ mean, const float* __restrict__ var,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    const float* __restrict__ c2, float* __restrict__ V,
                                                                    float* __restrict__ asum) {
  const int b = blockIdx.x, k0 = blockIdx.y * NV_KT;
  const int F4 = F >> 2;
  __shared__ float as[64][NV_KT];                // a[b, s, k0..k0+KT) for up to 64 frames
  for (int i = threadIdx.x; i < S * NV_KT; i += 256) {
    const int s = i / NV_KT, kk = i % NV_KT;
    as[s][kk] = (k0 + kk < K) ? a[((long)b * S + s) * K + k0 + kk] : 0.f;
  }
  __syncthreads();
  float sk[NV_KT];
#pragma unroll
  for (int kk = 0; kk < NV_KT; ++kk) {
    float t = 0.f;
    for (int s = 0; s < S; ++s) t += as[s][kk];
    sk[kk] = t;
  }
  if (threadIdx.x < NV_KT && k0 + threadIdx.x < K) asum[(long)b * K + k0 + threadIdx.x] = sk[threadIdx.x];
  for (int f4 = threadIdx.x; f4 < F4; f4 += 256) {
    const float4 mu = ((const float4*)mean)[f4], va = ((const float4*)var)[f4], ga = ((const float4*)gamma)[f4], be = ((const float4*)beta)[f4];
    const float4 sc = make_float4(rsqrtf(va.x + 1e-3f) * ga.x, rsqrtf(va.y + 1e-3f) * ga.y, rsqrtf(va.z + 1e-3f) * ga.z, rsqrtf(va.w + 1e-3f) * ga.w);
    float4 acc[NV_KT];
#pragma unroll
    for (int kk = 0; kk < NV_KT; ++kk) acc[kk] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < S; ++s) {
      const float4 x = ((const float4*)(r + ((long)b * S + s) * F))[f4];
      const float4 xb = make_float4((x.x - mu.x) * sc.x + be.x, (x.y - mu.y) * sc.y + be.y, (x.z - mu.z) * sc.z + be.z, (x.w - mu.w) * sc.w + be.w);
#pragma unroll
      for (int kk = 0; kk < NV_KT; ++kk) {
        const float w = as[s][kk];
        acc[kk].x += w * xb.x; acc[kk].y += w * xb.y; acc[kk].z += w * xb.z; acc[kk].w += w * xb.w;
      }
    }
#pragma unroll
    for (int kk = 0; kk < NV_KT; ++kk) {
      if (k0 + kk >= K) break;
      const float4 c = ((const float4*)(c2 + (long)(k0 + kk) * F))[f4];
      ((float4*)(V + ((long)b * K + k0 + kk) * F))[f4] =
          make_float4(acc[kk].x - sk[kk] * c.x, acc[kk].y - sk[kk] * c.y, acc[kk].z - sk[kk] * c.z, acc[kk].w - sk[kk] * c.w);
    }
  }
}
extern "C" int evc_netvlad_aggregate_fwd(const float* a, const float* r, int B, int S, int K, int F, const float* mean, const float* var,
                                         const float* gamma, const float* beta, const float* c2, float* V, float* asum, void* stream) {
  EVC_REQUIRE(B > 0 && S > 0 && S <= 64 && K > 0 && F > 0 && F % 4 == 0, EVC_ERR_BAD_SHAPE, "evc_netvlad_aggregate_fwd: needs S <= 64, F %% 4 == 0");
  hipLaunchKernelGGL(netvlad_aggregate_fwd_kernel, dim3(B, (K + NV_KT - 1) / NV_KT), dim3(256), 0, (hipStream_t)stream, a, r, S, K, F, mean,
                     var, gamma, beta, c2, V, asum);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// backward of the aggregation, one workgroup per video:
//   dx_bn[b,s,:] = sum_k a[b,s,k] * dV[b,k,:]                      (thread per float4 feature column, all S frames in registers)
//   da[b,s,k]    = sum_f dV[b,k,f] * (x_bn[b,s,f] - c2[k][f])       (wave w takes clusters w, w+4, ...; wave-wide dot products)
__global__ __launch_bounds__(256) void netvlad_aggregate_bwd_kernel(const float* __restrict__ a, const float* __restrict__ r, int S, int K,
                                                                    int F, const float* __restrict__ mean, const float* __restrict__ var,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    const float* __restrict__ c2, const float* __restrict__ dV,
                                                                    float* __restrict__ da, float* __restrict__ dx) {
  const int b = blockIdx.x;
  const int F4 = F >> 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // ---- dx_bn ----
  for (int f4 = threadIdx.x; f4 < F4; f4 += 256) {
    for (int s0 = 0; s0 < S; s0 += 16) {                       // 16 frames at a time (register budget)
      float4 acc[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k = 0; k < K; ++k) {
        const float4 d = ((const float4*)(dV + ((long)b * K + k) * F))[f4];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (s0 + i < S) {
            const float w = a[((long)b * S + s0 + i) * K + k];
            acc[i].x += w * d.x; acc[i].y += w * d.y; acc[i].z += w * d.z; acc[i].w += w * d.w;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (s0 + i < S) ((float4*)(dx + ((long)b * S + s0 + i) * F))[f4] = acc[i];
    }
  }
  // ---- da ----
  for (int k = wave; k < K; k += 4) {
    for (int s = 0; s < S; ++s) {
      float dot = 0.f;
      for (int f4 = lane; f4 < F4; f4 += 64) {
        const float4 d = ((const float4*)(dV + ((long)b * K + k) * F))[f4];
        const float4 x = ((const float4*)(r + ((long)b * S + s) * F))[f4];
        const float4 mu = ((const float4*)mean)[f4], va = ((const float4*)var)[f4], ga = ((const float4*)gamma)[f4], be = ((const float4*)beta)[f4];
        const float4 c = ((const float4*)(c2 + (long)k * F))[f4];
        dot += d.x * ((x.x - mu.x) * rsqrtf(va.x + 1e-3f) * ga.x + be.x - c.x) + d.y * ((x.y - mu.y) * rsqrtf(va.y + 1e-3f) * ga.y + be.y - c.y) +
               d.z * ((x.z - mu.z) * rsqrtf(va.z + 1e-3f) * ga.z + be.z - c.z) + d.w * ((x.w - mu.w) * rsqrtf(va.w + 1e-3f) * ga.w + be.w - c.w);
      }
      dot = wave_sum(dot);
      if (lane == 0) da[((long)b * S + s) * K + k] = dot;
    }
  }
}
extern "C" int evc_netvlad_aggregate_bwd(const float* a, const float* r, int B, int S, int K, int F, const float* mean, const float* var,
                                         const float* gamma, const float* beta, const float* c2, const float* dV, float* da, float* dx,
                                         void* stream) {
  EVC_REQUIRE(B > 0 && S > 0 && K > 0 && F > 0 && F % 4 == 0, EVC_ERR_BAD_SHAPE, "evc_netvlad_aggregate_bwd: bad shape");
  hipLaunchKernelGGL(netvlad_aggregate_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, a, r, S, K, F, mean, var, gamma, beta, c2, dV, da, dx);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// dc2[k][f] = - sum_b asum[b][k] * dV[b][k][f]    (thread per (k, float4 column), b in order: run-to-run identical)
__global__ void netvlad_dcenters_kernel(const float* __restrict__ asum, const float* __restrict__ dV, int B, int K, int F,
                                        float* __restrict__ dc2) {
  const long n4 = (long)K * (F >> 2);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i / (F >> 2));
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = 0; b < B; ++b) {
      const float w = asum[(long)b * K + k];
      const float4 d = ((const float4*)(dV + (long)b * K * F))[i];
      acc.x -= w * d.x; acc.y -= w * d.y; acc.z -= w * d.z; acc.w -= w * d.w;
    }
    ((float4*)dc2)[i] = acc;
  }
}
extern "C" int evc_netvlad_dcenters(const float* asum, const float* dV, int B, int K, int F, float* dc2, void* stream) {
  EVC_REQUIRE(B > 0 && K > 0 && F > 0 && F % 4 == 0, EVC_ERR_BAD_SHAPE, "evc_netvlad_dcenters: bad shape");
  hipLaunchKernelGGL(netvlad_dcenters_kernel, dim3(nv_grid((long)K * (F >> 2))), dim3(256), 0, (hipStream_t)stream, asum, dV, B, K, F, dc2);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// ---- normalisation: U_k = V_k / max(|V_k|, 1e-6) per cluster, Y = U / max(|U|, 1e-6) (tf.nn.l2_normalize's epsilon 1e-12 on the
// squared norms).  One workgroup per video; n1 [B][K], n2 [B] are kept for the backward pass.
__device__ __forceinline__ float block_sum_256(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(256) void netvlad_normalize_fwd_kernel(const float* __restrict__ V, int K, int F, float* __restrict__ n1,
                                                                    float* __restrict__ n2, float* __restrict__ Yf, bf16_t* __restrict__ Yb) {
  extern __shared__ float sn1[];                 // [K] cluster norms
  __shared__ float red[4];
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* Vb = V + (long)b * K * F;
  for (int k = wave; k < K; k += 4) {
    float s = 0.f;
    for (int f = lane; f < F; f += 64) { const float v = Vb[(long)k * F + f]; s += v * v; }
    s = wave_sum(s);
    if (lane == 0) sn1[k] = sqrtf(fmaxf(s, 1e-12f));
  }
  __syncthreads();
  float t = 0.f;                                 // |U|^2 = sum_k |V_k|^2 / n1_k^2
  for (long i = threadIdx.x; i < (long)K * F; i += 256) { const float u = Vb[i] / sn1[i / F]; t += u * u; }
  const float nn2 = sqrtf(fmaxf(block_sum_256(t, red), 1e-12f));
  if (threadIdx.x == 0) n2[b] = nn2;
  for (int k = threadIdx.x; k < K; k += 256) n1[(long)b * K + k] = sn1[k];
  for (long i = threadIdx.x; i < (long)K * F; i += 256) {
    const float y = Vb[i] / (sn1[i / F] * nn2);
    if (Yf) Yf[(long)b * K * F + i] = y;
    Yb[(long)b * K * F + i] = f32_to_bf16(y);
  }
}
extern "C" int evc_netvlad_normalize_fwd(const float* V, int B, int K, int F, float* n1, float* n2, float* Y_f32, evc_bf16* Y_bf16,
                                         void* stream) {
  EVC_REQUIRE(B > 0 && K > 0 && K <= 4096 && F > 0 && Y_bf16, EVC_ERR_BAD_SHAPE, "evc_netvlad_normalize_fwd: bad shape");
  hipLaunchKernelGGL(netvlad_normalize_fwd_kernel, dim3(B), dim3(256), K * sizeof(float), (hipStream_t)stream, V, K, F, n1, n2, Y_f32, Y_bf16);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}

// dU = (dY - Y (Y . dY)) / n2;  dV_k = (dU_k - U_k (U_k . dU_k)) / n1_k        (U, Y recomputed from V, n1, n2)
__global__ __launch_bounds__(256) void netvlad_normalize_bwd_kernel(const float* __restrict__ V, const float* __restrict__ n1,
                                                                    const float* __restrict__ n2, const float* __restrict__ dY, int K, int F,
                                                                    float* __restrict__ dV) {
  extern __shared__ float tk[];                  // [K] U_k . dU_k
  __shared__ float red[4];
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* Vb = V + (long)b * K * F;
  const float* dYb = dY + (long)b * K * F;
  const float* n1b = n1 + (long)b * K;
  const float nn2 = n2[b];
  float t = 0.f;
  for (long i = threadIdx.x; i < (long)K * F; i += 256) t += Vb[i] / (n1b[i / F] * nn2) * dYb[i];
  const float ydy = block_sum_256(t, red);
  for (int k = wave; k < K; k += 4) {
    float s = 0.f;
    const float in1 = 1.0f / n1b[k];
    for (int f = lane; f < F; f += 64) {
      const float u = Vb[(long)k * F + f] * in1;
      const float du = (dYb[(long)k * F + f] - (u / nn2) * ydy) / nn2;
      s += u * du;
    }
    s = wave_sum(s);
    if (lane == 0) tk[k] = s;
  }
  __syncthreads();
  for (long i = threadIdx.x; i < (long)K * F; i += 256) {
    const int k = (int)(i / F);
    const float in1 = 1.0f / n1b[k];
    const float u = Vb[i] * in1;
    const float du = (dYb[i] - (u / nn2) * ydy) / nn2;
    dV[(long)b * K * F + i] = (du - u * tk[k]) * in1;
  }
}
extern "C" int evc_netvlad_normalize_bwd(const float* V, const float* n1, const float* n2, const float* dY, int B, int K, int F, float* dV,
                                         void* stream) {
  EVC_REQUIRE(B > 0 && K > 0 && K <= 4096 && F > 0, EVC_ERR_BAD_SHAPE, "evc_netvlad_normalize_bwd: bad shape");
  hipLaunchKernelGGL(netvlad_normalize_bwd_kernel, dim3(B), dim3(256), K * sizeof(float), (hipStream_t)stream, V, n1, n2, dY, K, F, dV);
  EVC_LAUNCH_CHECK();
  return EVC_OK;
}
