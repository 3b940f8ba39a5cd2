// Probe for v_mfma_scale_f32_16x16x128_f8f6f4 with OCP e4m3 operands (the "high" mode's low-order weight term):
//  1. the K index of an operand byte is the same function of (lane group, register byte) for A and B - so any staging that
//     feeds A and B through the SAME chunk -> register pattern contracts the right pairs (the ring loop's two 16-byte chunks
//     q and 4 + q of a 128-byte stage row);  2. the e8m0 scale operands multiply the product by 2^(sa - 127) * 2^(sb - 127);
//  3. cycles per instruction next to v_mfma_f32_16x16x32_f16 (expected: 2x the cycles at 4x the K).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/probes/fp8_mfma_probe.hip -o scripts/probes/fp8_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void contract(const uint8_t* A, const uint8_t* B, float* C, int sa, int sb) {   // A, B: [16][128] e4m3 bytes, row-major
  const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  v8i a, b;
  const v4i a0 = *(const v4i*)(A + r * 128 + q * 16), a1 = *(const v4i*)(A + r * 128 + (4 + q) * 16);
  const v4i b0 = *(const v4i*)(B + r * 128 + q * 16), b1 = *(const v4i*)(B + r * 128 + (4 + q) * 16);
  a = v8i{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
  b = v8i{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, sa, 0, sb);
  for (int i = 0; i < 4; ++i) C[(q * 4 + i) * 16 + r] = acc[i];     // D[row = 4q + i][col = r]: row from A, col from B
}

__global__ void cvt(const float* x, uint8_t* y, int n) {
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (i + 1 < n) {
    const int p = __builtin_amdgcn_cvt_pk_fp8_f32(x[i], x[i + 1], 0, false);
    y[i] = p & 0xff; y[i + 1] = (p >> 8) & 0xff;
  }
}

template <int KIND>
__global__ void rate(float* out, int iters, int sa, int sb) {
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x; b[i] = 0x38383838 + i; }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, sa, 0, sb);
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

static float e4m3_to_f32(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -f : f;
}

int main() {
  std::vector<uint8_t> A(16 * 128), B(16 * 128);
  srand(5);
  for (auto& v : A) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v = 0x30; }     // no NaN codes
  for (auto& v : B) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v = 0x31; }
  uint8_t *dA, *dB; float* dC;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, 256 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  int bad = 0;
  for (int pass = 0; pass < 2; ++pass) {
    const int sa = pass ? 120 : 127, sb = pass ? 110 : 127;
    contract<<<1, 64>>>(dA, dB, dC, sa, sb);
    float C[256];
    hipMemcpy(C, dC, sizeof(C), hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double ref = 0, mag = 0;
        for (int k = 0; k < 128; ++k) { const double p = (double)e4m3_to_f32(A[i * 128 + k]) * e4m3_to_f32(B[j * 128 + k]); ref += p; mag += fabs(p); }
        ref = ldexp(ref, (sa - 127) + (sb - 127)); mag = ldexp(mag, (sa - 127) + (sb - 127));
        const double err = fabs(C[i * 16 + j] - ref) / mag;
        if (err > worst) worst = err;
      }
    printf("contract sa=%d sb=%d: worst |err| / sum|products| = %.2e %s\n", sa, sb, worst, worst < 1e-6 ? "OK" : "MISMATCH");
    bad += worst >= 1e-6;
  }
  {   // conversion: round-to-nearest-even e4m3, what happens above 448
    const float xs[] = {0.f, 1.f, 1.0625f, 1.1875f, 0.0019531f, 0.001f, 447.f, 448.f, 449.f, 480.f, 1000.f, -3.3f, 17.5f, 0.0146484f};
    const int n = sizeof(xs) / 4;
    float* dx; uint8_t* dy; uint8_t y[32];
    hipMalloc(&dx, sizeof(xs)); hipMalloc(&dy, 32);
    hipMemcpy(dx, xs, sizeof(xs), hipMemcpyHostToDevice);
    cvt<<<1, 16>>>(dx, dy, n);
    hipMemcpy(y, dy, n, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("cvt %g -> 0x%02x = %g\n", xs[i], y[i], e4m3_to_f32(y[i]));
  }
  float* dout; hipMalloc(&dout, 256 * 4 * 1024);
  for (int kind = 0; kind < 2; ++kind) {
    float cyc;
    for (int rep = 0; rep < 2; ++rep) {
      if (kind == 0) rate<0><<<256, 256>>>(dout, 2000, 127, 127); else rate<1><<<256, 256>>>(dout, 2000, 127, 127);
      hipDeviceSynchronize();
    }
    hipMemcpy(&cyc, dout, 4, hipMemcpyDeviceToHost);
    printf("%s: %.1f clock64 ticks per instruction per wave (4 waves per CU: one per SIMD)\n", kind == 0 ? "mfma_scale 16x16x128 e4m3" : "mfma 16x16x32 f16", cyc);
  }
  return bad;
}
