// Shared host/device helpers for the evc HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/evc.h"

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef uint16_t f16_t;   // raw IEEE binary16 bits
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// thread-local last-error text (evc_last_error)
void evc_set_error(const char* fmt, ...);

#define EVC_CHECK_HIP(expr)                                                        \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) {                                                        \
      evc_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return EVC_ERR_HIP;                                                          \
    }                                                                              \
  } while (0)

#define EVC_REQUIRE(cond, code, ...)   \
  do {                                 \
    if (!(cond)) {                     \
      evc_set_error(__VA_ARGS__);      \
      return (code);                   \
    }                                  \
  } while (0)

#define EVC_LAUNCH_CHECK() EVC_CHECK_HIP(hipGetLastError())

// ---- device helpers -------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even; NaN stays NaN (plain cast semantics, see guide "Correctness boundaries")
// f32 -> bf16, round to nearest even: gfx950's v_cvt_pk_bf16_f32 (one instruction; a software
// add-and-shift rounding costs ~6 VALU instructions per value in the epilogues)
typedef __bf16 evc_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float evc_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2_hw(float lo, float hi) {   // lo in bits 0-15
  const evc_f32x2_t v = {lo, hi};
  const evc_bf16x2_t r = __builtin_convertvector(v, evc_bf16x2_t);
  return *(const uint32_t*)&r;
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2_hw(f, 0.0f) & 0xffffu); }
// f32 -> IEEE f16, round to nearest even (v_cvt_pk_f16_f32 / v_cvt_f16_f32); lo in bits 0-15
typedef _Float16 evc_f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_f16x2_hw(float lo, float hi) {
  const evc_f32x2_t v = {lo, hi};
  const evc_f16x2_t r = __builtin_convertvector(v, evc_f16x2_t);
  return *(const uint32_t*)&r;
}
__device__ __forceinline__ f16_t f32_to_f16(float f) { return (f16_t)(pack_f16x2_hw(f, 0.0f) & 0xffffu); }
__device__ __forceinline__ float f16_to_f32(f16_t v) { return (float)__builtin_bit_cast(_Float16, v); }

// One MFMA depth of the 16-bit loops: v_mfma_f32_16x16x32_bf16, or - F16 - v_mfma_f32_16x16x32_f16 on the same 16-byte
// fragments (same rate, 11 significand bits instead of 8: the "high" precision forward of the L1 levels).
template <bool F16>
__device__ __forceinline__ f32x4 mfma16(const bf16x8 a, const bf16x8 b, const f32x4 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division sequence hipcc emits for `/` without fast-math:
// the gate tails evaluate 5 of these per (row, unit) and were VALU-bound on them.
__device__ __forceinline__ float rcpf_(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sqrtf_(float x) { return __builtin_amdgcn_sqrtf(x); }   // v_sqrt_f32 (1 ulp)
// the Adam step p - lr_t * m / (sqrt(v) + eps) of both optimizer kernels (identical rounding in the fused and
// the plain path); 1-ulp hardware sqrt / rcp: ~1e-7 relative on an update that is itself ~1e-3 of the weight
__device__ __forceinline__ float adam_step_(float p, float m, float v, float lr_t, float eps) {
  return p - lr_t * m * rcpf_(sqrtf_(v) + eps);
}
__device__ __forceinline__ float sigmoidf_(float x) { return rcpf_(1.0f + __expf(-x)); }
// tanh via exp: exact to ~2e-7 relative on the range the LSTM uses
__device__ __forceinline__ float tanhf_(float x) {
  float ax = fabsf(x);
  float e = __expf(-2.0f * ax);
  float t = (1.0f - e) * rcpf_(1.0f + e);
  return copysignf(t, x);
}


// Stores of an epilogue by uniform base + 32-bit lane byte offset, with a cache policy (template POLICY):
//   0  plain (the line stays dirty in the XCD's L2 and leaves at the kernel boundary: + bytes / 6 TB/s on the boundary, MI355X_MICROARCH.md "boundary")
//   1  sc1 = write-through: the bytes leave L2 as they are stored, under the rest of the epilogue
//   2  nt
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
// (store16: a store of more than 8 bytes reads its data registers late - a VALU write of them needs 2 wait states behind it on gfx940+, which the compiler
//  inserts for its own stores but not behind an asm statement: the s_nop 1.  Found in round 6 with one s_or_b64 between the store and a v_or_b32 into its
//  first data register: the tape's dword 0 of every chunk came out as that v_or's result, scripts/dbof_nt_diag.py.)
template <int POLICY>
__device__ __forceinline__ void store16(void* base, uint32_t off, u32x4_t v) {
  if constexpr (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(off), "v"(v), "s"(base) : "memory");
  else if constexpr (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(off), "v"(v), "s"(base) : "memory");
  else *(u32x4_t*)((char*)base + off) = v;
}
template <int POLICY>
__device__ __forceinline__ void store8(void* base, uint32_t off, u32x2_t v) {
  if constexpr (POLICY == 1) asm volatile("global_store_dwordx2 %0, %1, %2 sc1" ::"v"(off), "v"(v), "s"(base) : "memory");
  else if constexpr (POLICY == 2) asm volatile("global_store_dwordx2 %0, %1, %2 nt" ::"v"(off), "v"(v), "s"(base) : "memory");
  else *(u32x2_t*)((char*)base + off) = v;
}
template <int POLICY>
__device__ __forceinline__ void store4(void* base, uint32_t off, uint32_t v) {
  if constexpr (POLICY == 1) asm volatile("global_store_dword %0, %1, %2 sc1" ::"v"(off), "v"(v), "s"(base) : "memory");
  else if constexpr (POLICY == 2) asm volatile("global_store_dword %0, %1, %2 nt" ::"v"(off), "v"(v), "s"(base) : "memory");
  else *(uint32_t*)((char*)base + off) = v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Dynamic e4m3 range (round 6): the 64 partial |x| maxima evc_absmax_partials left in `ws` -> the number of bits d >= 0 by which an e4m3 image
// scaled by 2^hi_exp must be shifted DOWN so that its largest element stays representable (max|x| 2^(hi_exp - d) <= 448).  The writer of the image
// (evc_cast_f32_to_f16_fp8x_dyn) and its reader (evc_gemm_nt_f16_fp8_dyn, whose product scale grows by the same d) both call this on the same 64
// floats: d = 0 - every bit identical to the fixed-scale entries - as long as max|x| <= 448 2^-hi_exp.  Whole wave must call it (shuffles).
#define EVC_AMAX_SLOTS 64
__device__ __forceinline__ int fp8_range_drop(const float* __restrict__ ws, int hi_exp) {
  const float amax = wave_max(ws[threadIdx.x & 63]);
  if (!(amax > 0.f) || !(amax < 3.0e38f)) return 0;            // zeros / inf / NaN: nothing to rescue, keep the fixed scale
  int k;
  const float m = frexpf(amax, &k);                              // amax = m 2^k, m in [0.5, 1)
  // largest e with amax 2^e <= 448 = 0.875 2^9:  e = 9 - k if m <= 0.875 else 8 - k
  const int e = (m <= 0.875f ? 9 : 8) - k;
  const int d = hi_exp - e;
  return d > 0 ? (d < 40 ? d : 40) : 0;
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// EVC_DETERMINISTIC=1 (read once per process): no floating-point atomics on the training path - split-K joins are not used (one
// workgroup per output tile), reductions that end in same-address atomics run on one block or through plain partial sums in a
// fixed order.  Identical inputs then give identical bits run to run and box to box; what it costs is in DESIGN.md 7.
#include <stdlib.h>
static inline bool evc_deterministic() {
  // the same rule as ops.DETERMINISTIC in Python: set, not empty and not "0" (so "true" / "yes" switch BOTH sides on, never one of them)
  static const bool on = [] { const char* e = getenv("EVC_DETERMINISTIC"); return e != nullptr && e[0] != '\0' && !(e[0] == '0' && e[1] == '\0'); }();
  return on;
}
