// Probe of ds_read_b64_tr_b16 semantics on gfx950: which (row, col) does each lane receive?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s4;
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 64];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = i;   // value = row*64 + col
  __syncthreads();
  int lane = threadIdx.x;
  int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  short* addr = lds + (g * 4 + q) * 64 + p * 4;               // lane 4q+p of group g: row g*4+q, cols 4p..4p+3
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)addr);
  *(s4*)(out + lane * 4) = v;
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", h[l * 4 + j] / 64, h[l * 4 + j] % 64);
    printf("\n");
  }
  return 0;
}
