// Per-CU ingest rate of L2-resident data on gfx950, by load path: LDS-DMA (global_load_lds_dwordx4: what the GEMM ring
// loops use), plain 16-byte loads into registers, and registers + ds_write_b128.  Every workgroup streams its own REGION
// bytes over and over (resident in its XCD's L2 after the first pass; larger than the CU's L1).
//   hipcc --offload-arch=gfx950 -O3 -o ingest_probe ingest_probe.hip && ./ingest_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

extern __shared__ __attribute__((aligned(16))) char lds[];

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MODE 0: LDS-DMA, DEPTH pieces (1 KiB each) in flight per wave; MODE 1: registers; MODE 2: registers + ds_write_b128
template <int MODE, int DEPTH>
__global__ __launch_bounds__(1024) void probe(const char* buf, long region, int iters, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const char* base = buf + (long)blockIdx.x * region;
  const int pieces = (int)(region / 1024);           // 1 KiB per wave-instruction
  const int per_wave = pieces / nw;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      char* dst = lds + wave * (DEPTH * 1024);
      for (int p = 0; p < per_wave; p += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          const char* src = base + ((long)(wave * per_wave + p + d) * 1024) + lane * 16;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(dst + d * 1024), 16, 0, 0);
        }
        wait_vm<0>();
      }
    } else {
      float4 v[DEPTH];
      for (int p = 0; p < per_wave; p += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
          v[d] = *(const float4*)(base + ((long)(wave * per_wave + p + d) * 1024) + lane * 16);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          if (MODE == 2) *(float4*)(lds + wave * (DEPTH * 1024) + d * 1024 + lane * 16) = v[d];
          else asm volatile("" ::"v"(v[d].x), "v"(v[d].y), "v"(v[d].z), "v"(v[d].w));
        }
      }
      if (MODE == 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    }
  }
  if (MODE == 2 || MODE == 0) acc += *(float*)(lds + threadIdx.x * 4);
  if (acc == 123.456f) sink[0] = acc;
}

// GEMM-like staging of a [ROWS][KB bytes] K-contiguous panel shared by all workgroups (L2-resident): K step = CH bytes of
// every row (CH = 64: what the BK = 32 ring loops do - half of each 128-byte line per K step, the other half one K step
// later; CH = 128: whole lines).  One LDS-DMA piece = 1024 / CH rows x CH bytes; DEPTH K steps in flight.
template <int CH, int DEPTH>
__global__ __launch_bounds__(512) void probe_panel(const char* buf, int rows, long ld, int ksteps, int iters, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int RPP = 1024 / CH;                     // rows per piece
  constexpr int LPR = CH / 16;                       // lanes per row
  const int pieces = rows / RPP;                     // per K step
  const int per_wave = pieces / 8;
  const int stage_bytes = rows * CH;
  for (int it = 0; it < iters; ++it) {
    for (int k = 0; k < ksteps; ++k) {
      char* sb = lds + (k % DEPTH) * stage_bytes;
      for (int p = 0; p < per_wave; ++p) {
        const int piece = wave * per_wave + p;
        const int row = piece * RPP + lane / LPR;
        const char* src = buf + (long)row * ld + (long)k * CH + (lane % LPR) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(sb + piece * 1024), 16, 0, 0);
      }
      if (k >= DEPTH - 1) {                           // wait for the oldest stage (per_wave pieces each)
        if (per_wave == 4) { if (DEPTH == 4) wait_vm<12>(); else if (DEPTH == 2) wait_vm<4>(); else wait_vm<0>(); }
        else if (per_wave == 2) { if (DEPTH == 4) wait_vm<6>(); else if (DEPTH == 2) wait_vm<2>(); else wait_vm<0>(); }
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
      }
    }
    wait_vm<0>();
  }
  float acc = *(float*)(lds + threadIdx.x * 4);
  if (acc == 123.456f) sink[0] = acc;
}

// CH = 64 with the two halves of every line requested back to back (the second K step's piece right behind the first's)
template <int DEPTH>
__global__ __launch_bounds__(512) void probe_panel_pair(const char* buf, int rows, long ld, int ksteps, int iters, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pieces = rows / 16;
  const int per_wave = pieces / 8;
  const int stage_bytes = rows * 64;
  for (int it = 0; it < iters; ++it) {
    for (int k = 0; k < ksteps; k += 2) {
      char* sb0 = lds + (k % DEPTH) * stage_bytes;
      char* sb1 = lds + ((k + 1) % DEPTH) * stage_bytes;
      for (int p = 0; p < per_wave; ++p) {
        const int piece = wave * per_wave + p;
        const int row = piece * 16 + lane / 4;
        const char* src = buf + (long)row * ld + (long)k * 64 + (lane % 4) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(sb0 + piece * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 64),
                                         (__attribute__((address_space(3))) void*)(sb1 + piece * 1024), 16, 0, 0);
      }
      if (k >= DEPTH - 2) {
        if (per_wave == 4) wait_vm<8>(); else if (per_wave == 2) wait_vm<4>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
      }
    }
    wait_vm<0>();
  }
  float acc = *(float*)(lds + threadIdx.x * 4);
  if (acc == 123.456f) sink[0] = acc;
}

template <int DEPTH>
static void run_panel_pair(const char* buf, int rows, long ld, int kbytes, float* sink) {
  const int iters = 20, wgs = 256;
  const int ksteps = kbytes / 64;
  const int ldsb = DEPTH * rows * 64;
  CHECK(hipFuncSetAttribute((const void*)probe_panel_pair<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((probe_panel_pair<DEPTH>), dim3(wgs), dim3(512), ldsb, 0, buf, rows, ld, ksteps, 2, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((probe_panel_pair<DEPTH>), dim3(wgs), dim3(512), ldsb, 0, buf, rows, ld, ksteps, iters, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)wgs * rows * kbytes * iters;
  printf("panel %4d rows x %5d B, 64 B/row pieces issued in line pairs, %d K steps of LDS: %7.1f GB/s per workgroup, %6.2f TB/s total\n", rows,
         kbytes, DEPTH, bytes / wgs / ms / 1e6, bytes / ms / 1e9);
}

template <int CH, int DEPTH>
static void run_panel(const char* buf, int rows, long ld, int kbytes, float* sink) {
  const int iters = 20, wgs = 256;
  const int ksteps = kbytes / CH;
  const int ldsb = DEPTH * rows * CH;
  CHECK(hipFuncSetAttribute((const void*)probe_panel<CH, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((probe_panel<CH, DEPTH>), dim3(wgs), dim3(512), ldsb, 0, buf, rows, ld, ksteps, 2, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((probe_panel<CH, DEPTH>), dim3(wgs), dim3(512), ldsb, 0, buf, rows, ld, ksteps, iters, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)wgs * rows * kbytes * iters;
  printf("panel %4d rows x %5d B (ld %ld), K step = %3d B/row, %d K steps in flight: %7.1f GB/s per workgroup, %6.2f TB/s total\n", rows, kbytes, ld,
         CH, DEPTH, bytes / wgs / ms / 1e6, bytes / ms / 1e9);
}

template <int MODE, int DEPTH>
static void run(const char* name, const char* buf, long region, int wgs, int threads, float* sink) {
  const int iters = region > 65536 ? 40 : 200;
  const int ldsb = (threads / 64) * DEPTH * 1024;
  CHECK(hipFuncSetAttribute((const void*)probe<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((probe<MODE, DEPTH>), dim3(wgs), dim3(threads), ldsb, 0, buf, region, 20, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((probe<MODE, DEPTH>), dim3(wgs), dim3(threads), ldsb, 0, buf, region, iters, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)wgs * region * iters;
  printf("%-34s depth %2d  wgs %4d x %4d thr  region %4ld KB: %7.1f GB/s per workgroup, %6.2f TB/s total\n", name, DEPTH, wgs, threads,
         region / 1024, bytes / wgs / ms / 1e6, bytes / ms / 1e9);
}

int main() {
  const long region = 64 * 1024;
  const int maxwg = 512;
  char* buf; float* sink;
  CHECK(hipMalloc(&buf, (size_t)maxwg * region));
  CHECK(hipMemset(buf, 1, (size_t)maxwg * region));
  CHECK(hipMalloc(&sink, 16));
  for (int wgs : {256, 512}) {
    run<0, 2>("LDS-DMA", buf, region, wgs, 512, sink);
    run<0, 4>("LDS-DMA", buf, region, wgs, 512, sink);
    run<0, 8>("LDS-DMA", buf, region, wgs, 512, sink);
    run<1, 4>("registers", buf, region, wgs, 512, sink);
    run<1, 8>("registers", buf, region, wgs, 512, sink);
    run<2, 4>("registers + ds_write_b128", buf, region, wgs, 512, sink);
    run<2, 8>("registers + ds_write_b128", buf, region, wgs, 512, sink);
  }
  run<0, 16>("LDS-DMA", buf, region, 256, 512, sink);
  run<1, 16>("registers", buf, region, 256, 512, sink);
  run<0, 1>("LDS-DMA", buf, region, 256, 512, sink);
  run<0, 4>("LDS-DMA, 4 waves", buf, region, 256, 256, sink);
  run<1, 8>("registers, 4 waves", buf, region, 256, 256, sink);
  run<1, 8>("registers, 16 waves", buf, region, 256, 1024, sink);
  {   // Infinity-Cache-resident streaming: every workgroup cycles through its own 512 KB (128 MB in all: beyond the 32 MB of L2)
    const long big = 512 * 1024;
    char* buf2;
    CHECK(hipMalloc(&buf2, (size_t)256 * big));
    CHECK(hipMemset(buf2, 1, (size_t)256 * big));
    run<0, 2>("LDS-DMA, 512 KB regions", buf2, big, 256, 512, sink);
    run<0, 4>("LDS-DMA, 512 KB regions", buf2, big, 256, 512, sink);
    run<0, 8>("LDS-DMA, 512 KB regions", buf2, big, 256, 512, sink);
    run<0, 16>("LDS-DMA, 512 KB regions", buf2, big, 256, 512, sink);
    run<1, 8>("registers, 512 KB regions", buf2, big, 256, 512, sink);
    run<1, 16>("registers, 512 KB regions", buf2, big, 256, 512, sink);
    run<1, 32>("registers, 512 KB regions", buf2, big, 256, 512, sink);
    CHECK(hipFree(buf2));
  }
  // GEMM-like panels: 512 rows (A 256 + B 256 of a 256 x 256 tile) x 4352 bytes (K = 2176 bf16), shared by every workgroup
  run_panel<64, 2>(buf, 512, 4352, 4352, sink);
  run_panel<64, 4>(buf, 512, 4352, 4352, sink);
  run_panel<128, 2>(buf, 512, 4352, 4352, sink);
  run_panel<128, 1>(buf, 512, 4352, 4352, sink);
  run_panel_pair<4>(buf, 512, 4352, 4352, sink);
  // 256 rows (a 128 x 128 tile), K = 4096 bf16
  run_panel<64, 4>(buf, 256, 8192, 8192, sink);
  run_panel<128, 4>(buf, 256, 8192, 8192, sink);
  run_panel<128, 2>(buf, 256, 8192, 8192, sink);
  return 0;
}
