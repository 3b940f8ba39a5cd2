// Does global_load_lds_dwordx3 (gfx950) write lane l's 12 bytes at M0-base + 12 l, i.e. 768 contiguous bytes per wave-instruction?
// (the FP6 stages of the "high" forward stage rows of 96 bytes = 8 lanes x 12 bytes: one piece = 8 whole rows)
// Build: hipcc --offload-arch=gfx950 -O3 scripts/probes/dma12_probe.hip -o scripts/probes/dma12_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(const uint8_t* src, uint8_t* out, int stride) {
  __shared__ __attribute__((aligned(16))) char lds[4096];
  for (int i = threadIdx.x; i < 1024; i += 64) ((int*)lds)[i] = 0xEEEEEEEE;
  __syncthreads();
  const int lane = threadIdx.x;
  // lane -> row lane >> 3, 12-byte chunk lane & 7 of a [8][stride] source
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (lane >> 3) * stride + (lane & 7) * 12),
                                   (__attribute__((address_space(3))) void*)(lds + 256), 12, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 4096; i += 64) out[i] = lds[i];
}
int main() {
  const int stride = 200;
  std::vector<uint8_t> h(8 * stride);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)(i * 7 + 3);
  uint8_t *d, *o; hipMalloc(&d, h.size()); hipMalloc(&o, 4096);
  hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o, stride);
  std::vector<uint8_t> r(4096);
  hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int row = 0; row < 8; ++row)
    for (int b = 0; b < 96; ++b) bad += r[256 + row * 96 + b] != h[row * stride + b];
  int untouched = 0;
  for (int i = 0; i < 256; ++i) untouched += r[i] == 0xEE;
  for (int i = 256 + 768; i < 4096; ++i) untouched += r[i] == 0xEE;
  printf("dwordx3 LDS-DMA: %d mismatching bytes of 768 (contiguous 12-byte lanes), %d of %d other bytes untouched -> %s\n", bad, untouched, 4096 - 768,
         (bad == 0 && untouched == 4096 - 768) ? "OK" : "DIFFERENT LAYOUT");
  if (bad) { for (int i = 0; i < 64; ++i) printf("%02x ", r[256 + i]); printf("\n"); for (int i = 0; i < 64; ++i) printf("%02x ", h[i]); printf("\n"); }
  return bad != 0;
}
