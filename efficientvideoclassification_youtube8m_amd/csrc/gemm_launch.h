// Launch helpers shared by the GEMM-shaped translation units (evc_gemm.hip, evc_dbof.hip).
#pragma once
#include "gemm_core_tn.h"
#include "gemm_core_v3.h"
#include <mutex>
#include <vector>
#include <stdlib.h>

// A kernel is instantiated either on a v1 tile (TileCfg: static 2-stage LDS, K steps of 64)
// or a v2 tile (TileCfg2: dynamic 4-stage LDS ring, K steps of 32).
template <class Cfg> struct is_v2 { static constexpr bool value = false; };
template <int a, int b, int c, int d, int e, int f, bool g> struct is_v2<TileCfg2<a, b, c, d, e, f, g>> { static constexpr bool value = true; };
// v3 tiles (TileCfg3: the ring with 64-wide K stages, gemm_core_v3.h) are ring tiles too (dynamic LDS, same epilogues)
template <int a, int b, int c, int d, int e, int f, int g> struct is_v2<TileCfg3<a, b, c, d, e, f, g>> { static constexpr bool value = true; };
template <class Cfg> struct is_v3 { static constexpr bool value = false; };
template <int a, int b, int c, int d, int e, int f, int g> struct is_v3<TileCfg3<a, b, c, d, e, f, g>> { static constexpr bool value = true; };

extern __shared__ __attribute__((aligned(16))) char lds_dyn[];

// The ring loops address every LDS-DMA as base + 32-bit byte offset built from a 24-bit row index and a 24-bit row stride
// (gemm_core_v2.h): an operand of `rows` rows with leading dimension `ld` (bf16 elements) must fit that.
// the forward step's epilogue addresses its stores as base + 32-bit byte offset: gate records [rows][H] x 8 B, state rows (row_map is a
// permutation of [0, rows)) of ld_state floats, h rows of up to 3H bytes / 2H halfwords (the wide images)
static inline bool fwd_tail_ok(long rows, long H, long ld_state) {
  return rows * H * 8 + 32 < (1L << 32) && rows * ld_state * 4 < (1L << 32);
}
static inline bool ring_operand_ok(long rows, long ld) {
  return rows < (1L << 24) && ld * 2 < (1L << 24) && rows * ld * 2 < (1L << 32);
}

// v2 tiles use more dynamic LDS than the 64 KiB default: raise the limit once per kernel (keyed by the
// kernel's address - two kernels of one signature share this template instantiation).
static inline void allow_big_lds(const void* kern, int bytes) {
  static std::mutex mu;
  static std::vector<const void*> done;
  std::lock_guard<std::mutex> lk(mu);
  for (const void* k : done) if (k == kern) return;
  (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  done.push_back(kern);
}

template <class Cfg, class Kern, class... Args>
static inline void launch_cfg(Kern kern, int grid, hipStream_t st, Args... args) {
  if (is_v2<Cfg>::value) {
    allow_big_lds((const void*)kern, Cfg::LDS_BYTES);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, st, args...);
  } else {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::NT), 0, st, args...);
  }
}

// debug/benchmark override of the tile choice: EVC_FORCE_TILE = 1 (v2) | 2 (v1 128x128) | 3 (v1 64x64)
static inline int forced_tile() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("EVC_FORCE_TILE"); v = e ? atoi(e) : 0; }
  return v;
}
