// Shared device/host helpers for libmi_oov.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mi_oov.h"

namespace mi_oov {

constexpr int kWave = 64;       // CDNA wavefront
constexpr int kGroup = 16;      // lanes that share one lookup = one DPP row
constexpr int kBlock = 256;     // 4 waves, one per SIMD
constexpr int kMaxGrid = 2048;  // 256 CUs x 8 blocks: grid-stride beyond that

extern thread_local int g_last_hip_error;

inline int check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_last_hip_error = static_cast<int>(e);
    return MI_OOV_ERR_LAUNCH;
  }
  return MI_OOV_OK;
}

// Developer knobs ("Developer knobs"): an integer from the environment, read where it is first used
// and CLAMPED at parse time -- a value that does not parse or lies outside [lo, hi] is ignored (the default applies), so
// no knob can push a launch parameter out of the range its kernel was written for.  Results never depend on a knob.
inline int64_t env_knob(const char* name, int64_t dflt, int64_t lo, int64_t hi) {
  const char* e = getenv(name);
  if (!e || !*e) return dflt;
  char* end = nullptr;
  const long long v = strtoll(e, &end, 10);
  if (end == e || *end != '\0' || v < lo || v > hi) return dflt;
  return static_cast<int64_t>(v);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- DPP row (16-lane) all-reduce -----------------------------------------------------------
// Stride-halving tree over the 16 lanes of a DPP row; every lane receives the sum.
//   level 1: q_l = p_l + p_(l+8)   (l = 0..7)      row_ror:8
//   level 2: r_l = q_l + q_(l+4)   (l = 0..3)      row_ror:4   (q has period 8 by then)
//   level 3: s_l = r_l + r_(l+2)   (l = 0,1)       row_ror:2
//   level 4: s_0 + s_1                             row_ror:1
// After each level the values are periodic in the lane index, so a rotation by the stride
// delivers the xor-partner's value in both directions; IEEE add is commutative, hence all 16
// lanes hold bit-identical results, equal to oracle/oov_oracle.c::tree16.  This order (largest
// stride first) is what lets a kernel keep only ONE lane's share of H sums alive per level with
// bank-masked DPP adds (rows of 16 split into banks of 4 lanes).
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}

__device__ __forceinline__ float row16_sum(float v) {
  v = v + dpp_f32<0x128>(v);  // row_ror:8
  v = v + dpp_f32<0x124>(v);  // row_ror:4
  v = v + dpp_f32<0x122>(v);  // row_ror:2
  v = v + dpp_f32<0x121>(v);  // row_ror:1
  return v;
}

__device__ __forceinline__ float qnan() { return __int_as_float(0x7FC00000); }

// 4-element slice of a canonical-order dot product: continues the lane's fmaf chain.
__device__ __forceinline__ float dot4_fma(float4 x, float4 w, float p) {
  p = __builtin_fmaf(x.x, w.x, p);
  p = __builtin_fmaf(x.y, w.y, p);
  p = __builtin_fmaf(x.z, w.z, p);
  p = __builtin_fmaf(x.w, w.w, p);
  return p;
}

// multiply, then add (torch.mul(a, b).sum(1) rounds the product before the sum)
__device__ __forceinline__ float dot4_muladd(float4 x, float4 w, float p) {
  p = p + x.x * w.x;
  p = p + x.y * w.y;
  p = p + x.z * w.z;
  p = p + x.w * w.w;
  return p;
}

// Guarded 4-float load of elements [e, e+4) of a row of length L (scalar path: any alignment).
__device__ __forceinline__ float4 load4_guard(const float* row, int64_t e, int64_t L) {
  float4 v;
  v.x = (e + 0 < L) ? row[e + 0] : 0.f;
  v.y = (e + 1 < L) ? row[e + 1] : 0.f;
  v.z = (e + 2 < L) ? row[e + 2] : 0.f;
  v.w = (e + 3 < L) ? row[e + 3] : 0.f;
  return v;
}

__device__ __forceinline__ void store4_guard(float* row, int64_t e, int64_t L, float4 v) {
  if (e + 0 < L) row[e + 0] = v.x;
  if (e + 1 < L) row[e + 1] = v.y;
  if (e + 2 < L) row[e + 2] = v.z;
  if (e + 3 < L) row[e + 3] = v.w;
}

template <bool VEC>
__device__ __forceinline__ float4 load4(const float* row, int64_t e, int64_t L) {
  if constexpr (VEC) {
    return (e < L) ? *reinterpret_cast<const float4*>(row + e) : make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    return load4_guard(row, e, L);
  }
}

template <bool VEC>
__device__ __forceinline__ void store4(float* row, int64_t e, int64_t L, float4 v) {
  if constexpr (VEC) {
    if (e < L) *reinterpret_cast<float4*>(row + e) = v;
  } else {
    store4_guard(row, e, L, v);
  }
}

// Dynamic LDS above 64 KiB needs an explicit opt-in (up to the CU's 160 KiB).
// (The attribute sticks to the function: it is set when a kernel is first launched with a size, and again only when a
//  larger one is asked for -- the call costs ~1.5 us of host time, which an isolated launch from an idle stream would
//  otherwise pay inside the region its caller times.  One slot per kernel instantiation and thread.)
template <typename K>
inline int set_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) {
    thread_local K last_kernel = nullptr;
    thread_local size_t last_bytes = 0;
    thread_local int last_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (last_kernel == kernel && last_dev == dev && bytes <= last_bytes) return MI_OOV_OK;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
    if (e != hipSuccess) {
      g_last_hip_error = static_cast<int>(e);
      return MI_OOV_ERR_LAUNCH;
    }
    last_kernel = kernel;
    last_bytes = bytes;
    last_dev = dev;
  }
  return MI_OOV_OK;
}

constexpr int64_t kLdsLimit = 160 * 1024 - 512;

inline int grid_for(int64_t work_items, int64_t items_per_block) {
  int64_t blocks = (work_items + items_per_block - 1) / items_per_block;
  if (blocks < 1) blocks = 1;
  if (blocks > kMaxGrid) blocks = kMaxGrid;
  return static_cast<int>(blocks);
}

}  // namespace mi_oov
