// Probe for v_mfma_scale_f32_16x16x128_f8f6f4 with OCP e2m3 (FP6) operands - the round-5 candidate for the "high" mode's low-order
// correction stages (Next 1):
//  1. packing: a lane's 32 K elements are 192 contiguous bits (element i at bits 6i..6i+5) in the first 6 of the 8 operand registers;
//     the K position of (lane group, element) is the same function for A and B, so operands staged through one pattern meet;
//  2. the scale operands are PER LANE (a VGPR; opsel picks the byte): lane 16q + r scales the 32 elements it holds of row r;
//  3. cycles per instruction: e2m3 x e2m3 next to e4m3 x e4m3, the mixed forms, e2m1 and the 16-bit MFMA;
//  4. an LDS-fed stage of the forward step's tile (8 waves; per wave 8 + 4 fragments, 32 MFMAs per K = 128): ticks per stage with
//     96-byte e2m3 rows (3 x ds_read_b64 per fragment) next to 128-byte e4m3 rows (2 x ds_read_b128) - no global loads, so this is
//     the LDS-read / MFMA side of a stage alone.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/probes/fp6_mfma_probe.hip -o scripts/probes/fp6_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// A, B: [16 rows][4 blocks][24 bytes] (block q = the 32 elements lane 16q + r holds); sa, sb: [64] per-lane scale bytes
__global__ void contract6(const uint8_t* A, const uint8_t* B, const int* sa, const int* sb, float* C) {
  const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
  const int* pa = (const int*)(A + (r * 4 + q) * 24);
  const int* pb = (const int*)(B + (r * 4 + q) * 24);
  for (int i = 0; i < 6; ++i) { a[i] = pa[i]; b[i] = pb[i]; }
  a[6] = a[7] = 0x7fffffff; b[6] = b[7] = 0x55555555;       // must be ignored
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 2, 2, 0, sa[lane], 0, sb[lane]);
  for (int i = 0; i < 4; ++i) C[(q * 4 + i) * 16 + r] = acc[i];     // D[row = 4q + i][col = r]: row from A, col from B
}

template <int FA, int FB, bool F16>
__global__ void rate(float* out, int iters, int sa, int sb) {
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x18181818 + threadIdx.x; b[i] = 0x18181818 + i; }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (!F16) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], FA, FB, 0, sa, 0, sb);
      else {
        f16x8 fa, fb;
        __builtin_memcpy(&fa, &a, 16); __builtin_memcpy(&fb, &b, 16);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[i], 0, 0, 0);
      }
    }
  }
  const long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (float)(iters * 8);
}

// ---- LDS-fed stage: 8 waves (2 wave rows x 4 wave columns), tile 256 rows x 256 columns, per wave MI = 8 row fragments and 4 column fragments ----
// FP6: rows of 96 bytes, lane (r, q) reads 24 bytes at row*96 + ((q + 2*((row>>3)&1)) & 3)*24 as 3 x ds_read_b64
// FP8: rows of 128 bytes, chunk c at c ^ (row & 7), lane reads chunks q and 4+q as 2 x ds_read_b128
template <bool FP6>
__global__ __launch_bounds__(512) void stage_loop(float* out, int stages) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int RB = FP6 ? 96 : 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3;
  for (int i = tid; i < 512 * RB / 4; i += 512) ((int*)lds)[i] = 0x09090909 + (i & 3);
  __syncthreads();
  const int r = lane & 15, q = lane >> 4;
  f32x4 acc[8][4];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long t0 = clock64();
  for (int s = 0; s < stages; ++s) {
    v8i a[8], b[4];
#pragma unroll
    for (int f = 0; f < 12; ++f) {
      const int row = f < 8 ? wr * 128 + f * 16 + r : 256 + wc * 64 + (f - 8) * 16 + r;
      v8i v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (FP6) {
        const char* p = lds + row * 96 + ((q + 2 * ((row >> 3) & 1)) & 3) * 24;
        const v2i x0 = *(const v2i*)p, x1 = *(const v2i*)(p + 8), x2 = *(const v2i*)(p + 16);
        v = v8i{x0[0], x0[1], x1[0], x1[1], x2[0], x2[1], 0, 0};
      } else {
        const v4i x0 = *(const v4i*)(lds + row * 128 + ((q ^ (row & 7)) * 16)), x1 = *(const v4i*)(lds + row * 128 + (((4 + q) ^ (row & 7)) * 16));
        v = v8i{x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
      }
      if (f < 8) a[f] = v; else b[f - 8] = v;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = FP6 ? __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b[j], a[i], acc[i][j], 2, 2, 0, 127, 0, 127)
                        : __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b[j], a[i], acc[i][j], 0, 0, 0, 127, 0, 127);
    __builtin_amdgcn_s_barrier();
  }
  const long t1 = clock64();
  float sum = 0.f;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * 512 + tid] = sum;
  if (tid == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (float)stages;
}

static float e2m3_to_f32(int v) {
  const int s = (v >> 5) & 1, e = (v >> 3) & 3, m = v & 7;
  const float f = e == 0 ? m * 0.125f : ldexpf(1.0f + m / 8.0f, e - 1);
  return s ? -f : f;
}

int main() {
  // operands as 6-bit codes per (row, k), k = 0..127; lane block q = k / 32
  std::vector<int> ca(16 * 128), cb(16 * 128);
  srand(11);
  for (auto& v : ca) v = rand() & 63;
  for (auto& v : cb) v = rand() & 63;
  auto pack = [](const std::vector<int>& c) {
    std::vector<uint8_t> out(16 * 4 * 24, 0);
    for (int r = 0; r < 16; ++r)
      for (int k = 0; k < 128; ++k) {
        const int q = k / 32, i = k % 32, bit = 6 * i;
        uint8_t* p = out.data() + (r * 4 + q) * 24;
        for (int t = 0; t < 6; ++t)
          if ((c[r * 128 + k] >> t) & 1) p[(bit + t) / 8] |= 1u << ((bit + t) % 8);
      }
    return out;
  };
  std::vector<uint8_t> A = pack(ca), B = pack(cb);
  uint8_t *dA, *dB; float* dC; int *dsa, *dsb;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, 256 * 4); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  int bad = 0;
  for (int pass = 0; pass < 3; ++pass) {
    int sa[64], sb[64];
    for (int l = 0; l < 64; ++l) {
      sa[l] = pass == 0 ? 127 : pass == 1 ? 120 : 118 + (l * 7) % 13;
      sb[l] = pass == 0 ? 127 : pass == 1 ? 131 : 121 + (l * 5) % 11;
      if (pass == 2) { sa[l] |= 0x11223300; sb[l] |= 0x44556600; }       // the upper bytes must not matter at opsel 0
    }
    hipMemcpy(dsa, sa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb, 256, hipMemcpyHostToDevice);
    contract6<<<1, 64>>>(dA, dB, dsa, dsb, dC);
    float C[256];
    hipMemcpy(C, dC, sizeof(C), hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double ref = 0, mag = 0;
        for (int k = 0; k < 128; ++k) {
          const int q = k / 32;
          const double sc = ldexp(1.0, ((sa[16 * q + i] & 255) - 127) + ((sb[16 * q + j] & 255) - 127));
          const double p = (double)e2m3_to_f32(ca[i * 128 + k]) * e2m3_to_f32(cb[j * 128 + k]) * sc;
          ref += p; mag += fabs(p);
        }
        const double err = fabs(C[i * 16 + j] - ref) / mag;
        if (err > worst) worst = err;
      }
    printf("e2m3 contract, %s scales: worst |err| / sum|products| = %.2e %s\n", pass == 0 ? "unit" : pass == 1 ? "uniform" : "per-lane", worst,
           worst < 1e-6 ? "OK" : "MISMATCH");
    bad += worst >= 1e-6;
  }
  float* dout; hipMalloc(&dout, 256 * 4 * 1024);
  const char* names[] = {"e2m3 x e2m3", "e4m3 x e4m3", "e4m3 x e2m3", "e2m1 x e2m1", "e2m3 x e2m1", "f16 16x16x32", "e3m2 x e2m3", "e3m2 x e3m2"};
  for (int kind = 0; kind < 8; ++kind) {
    float cyc;
    for (int rep = 0; rep < 2; ++rep) {
      switch (kind) {
        case 0: rate<2, 2, false><<<256, 256>>>(dout, 2000, 127, 127); break;
        case 1: rate<0, 0, false><<<256, 256>>>(dout, 2000, 127, 127); break;
        case 2: rate<0, 2, false><<<256, 256>>>(dout, 2000, 127, 127); break;
        case 3: rate<4, 4, false><<<256, 256>>>(dout, 2000, 127, 127); break;
        case 4: rate<2, 4, false><<<256, 256>>>(dout, 2000, 127, 127); break;
        case 6: rate<3, 2, false><<<256, 256>>>(dout, 2000, 127, 127); break;
        case 7: rate<3, 3, false><<<256, 256>>>(dout, 2000, 127, 127); break;
        default: rate<0, 0, true><<<256, 256>>>(dout, 2000, 127, 127); break;
      }
      hipDeviceSynchronize();
    }
    hipMemcpy(&cyc, dout, 4, hipMemcpyDeviceToHost);
    printf("%-14s %.1f clock64 ticks per instruction per wave (4 waves per CU: one per SIMD)\n", names[kind], cyc);
  }
  hipFuncSetAttribute((const void*)stage_loop<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 512 * 128);
  hipFuncSetAttribute((const void*)stage_loop<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 512 * 128);
  for (int kind = 0; kind < 2; ++kind) {
    float cyc;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (kind == 0) stage_loop<true><<<256, 512, 512 * 128>>>(dout, 4000); else stage_loop<false><<<256, 512, 512 * 128>>>(dout, 4000);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(&cyc, dout, 4, hipMemcpyDeviceToHost);
    printf("LDS-fed stage (256x256 tile, K = 128, 8 waves), %s rows: %.0f ticks = %.3f us per stage\n", kind == 0 ? "96-byte e2m3" : "128-byte e4m3", cyc, ms * 1e3 / 4000);
  }
  return bad;
}
