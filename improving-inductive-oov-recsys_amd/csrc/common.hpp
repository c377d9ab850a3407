// Shared device/host helpers for libmi_oov.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- DPP row (16-lane) all-reduce -----------------------------------------------------------
// Balanced adjacent-pair tree over the 16 lanes of a DPP row; every lane receives the sum.
// Steps: xor 1, xor 2 (quad_perm), then row_half_mirror and row_mirror, which at that point
// pair equal-valued quads/octets, i.e. they are the xor-4 and xor-8 butterflies.  IEEE add is
// commutative, so all 16 lanes hold bit-identical results and the value equals
//   ((p0+p1)+(p2+p3)) + ((p4+p5)+(p6+p7)) + ... summed pairwise in natural order,
// which is what oracle/oov_oracle.c::tree16 computes.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}

__device__ __forceinline__ float row16_sum(float v) {
  v = v + dpp_f32<0xB1>(v);   // quad_perm [1,0,3,2]
  v = v + dpp_f32<0x4E>(v);   // quad_perm [2,3,0,1]
  v = v + dpp_f32<0x141>(v);  // row_half_mirror
  v = v + dpp_f32<0x140>(v);  // row_mirror
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
template <typename K>
inline int set_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
    if (e != hipSuccess) {
      g_last_hip_error = static_cast<int>(e);
      return MI_OOV_ERR_LAUNCH;
    }
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
