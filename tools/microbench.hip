// Developer microbenchmark (not part of the product): what does a 65536-row random gather of
// 256-B rows cost on MI355X as a function of loads in flight, store policy and access pattern?
//   hipcc -O3 --offload-arch=gfx950 tools/microbench.hip -o tools/microbench && tools/microbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <chrono>
#include <vector>

#include "mi_oov.h"
#include "../improving-inductive-oov-recsys_amd/csrc/common.hpp"
using namespace mi_oov;
namespace mi_oov { thread_local int g_last_hip_error = 0; }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }

// MODE 0: copy rows to out (plain stores)   1: copy with nontemporal stores
// MODE 2: read only (reduce each row to one float per row)  3: read only, no id indirection (row = b)
template <int R, int MODE>
__global__ __launch_bounds__(256) void gather_kernel(const int64_t* __restrict__ ids, int64_t B,
                                                     const float* __restrict__ feat, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, l16 = lane & 15, grp = lane >> 4, wv = threadIdx.x >> 6;
  const int64_t ntiles = (B + 4 * R - 1) / (4 * R);
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wv; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    int64_t row[R], id[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      row[r] = tile * (4 * R) + r * 4 + grp;
      if (row[r] >= B) row[r] = B - 1;
      id[r] = (MODE == 3) ? row[r] * 151 % 10000000 : ids[row[r]];
    }
    float4 x[R];
#pragma unroll
    for (int r = 0; r < R; ++r) x[r] = *reinterpret_cast<const float4*>(feat + id[r] * 64 + l16 * 4);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (MODE == 0) *reinterpret_cast<float4*>(out + row[r] * 64 + l16 * 4) = x[r];
      if (MODE == 1) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f v = {x[r].x, x[r].y, x[r].z, x[r].w};
        __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(out + row[r] * 64 + l16 * 4));
      }
      if (MODE >= 2) {
        float s = x[r].x + x[r].y + x[r].z + x[r].w;
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
        if (l16 == 0) out[row[r]] = s;
      }
    }
  }
}

// copy + WORK x 4 dependent-chain FMAs per round: how much VALU work hides under the gather?
template <int R, int WORK, bool DPP>
__global__ __launch_bounds__(256) void work_kernel(const int64_t* __restrict__ ids, int64_t B,
                                                   const float* __restrict__ feat, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, l16 = lane & 15, grp = lane >> 4, wv = threadIdx.x >> 6;
  const int64_t ntiles = (B + 4 * R - 1) / (4 * R);
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wv; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    int64_t row[R], id[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      row[r] = tile * (4 * R) + r * 4 + grp;
      if (row[r] >= B) row[r] = B - 1;
      id[r] = ids[row[r]];
    }
    float4 x[R];
#pragma unroll
    for (int r = 0; r < R; ++r) x[r] = *reinterpret_cast<const float4*>(feat + id[r] * 64 + l16 * 4);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float4 a = x[r];
#pragma unroll
      for (int i = 0; i < WORK; ++i) {
        a.x = __builtin_fmaf(a.x, 1.0001f, a.y);
        a.y = __builtin_fmaf(a.y, 0.9999f, a.z);
        a.z = __builtin_fmaf(a.z, 1.0002f, a.w);
        if (DPP) a.w = a.w + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a.x), 0x128, 0xF, 0xF, false));
        else a.w = __builtin_fmaf(a.w, 0.9998f, a.x);
      }
      *reinterpret_cast<float4*>(out + row[r] * 64 + l16 * 4) = a;
    }
  }
}

// Staged replica of the hot lsh kernel (score variant): which stage costs what?
//  STAGE 0 loads (x,u) + one DPP reduce   1 + 8 projections   2 + aggregate   3 + division   4 + score
template <int STAGE, int R, int ORDER = 0>
__global__ __launch_bounds__(256) void staged_kernel(const int64_t* __restrict__ ids, int64_t B,
                                                     const float* __restrict__ feat, const float* __restrict__ planes,
                                                     const float* __restrict__ buckets, const float* __restrict__ other,
                                                     float* __restrict__ score) {
  constexpr int H = 8;
  const int lane = threadIdx.x & 63, l16 = lane & 15, grp = lane >> 4, wv = threadIdx.x >> 6;
  const int64_t ntiles = (B + 4 * R - 1) / (4 * R);
  float4 pw[H], bw[H];
  int64_t id0[R];
  if (ORDER == 1) {  // the tile's ids go out before anything else
    const int64_t tile = (int64_t)blockIdx.x * 4 + wv;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int64_t rw = tile * (4 * R) + r * 4 + grp;
      if (rw >= B) rw = B - 1;
      id0[r] = ids[rw];
    }
  }
  __shared__ __attribute__((aligned(16))) float sw_raw[ORDER == 2 ? 2 * H * 64 : 4];
  if (ORDER == 2) {  // weights: 4 KB per workgroup through LDS instead of 16 KB per wave through L1
    const float* src = (threadIdx.x < H * 16) ? planes + threadIdx.x * 4 : buckets + (threadIdx.x - H * 16) * 4;
    const float4 v = *reinterpret_cast<const float4*>(src);
    *reinterpret_cast<float4*>(sw_raw + threadIdx.x * 4) = v;
    __syncthreads();
#pragma unroll
    for (int h = 0; h < H; ++h) {
      pw[h] = *reinterpret_cast<const float4*>(sw_raw + (h * 16 + l16) * 4);
      bw[h] = *reinterpret_cast<const float4*>(sw_raw + (H * 16 + h * 16 + l16) * 4);
    }
  } else {
#pragma unroll
    for (int h = 0; h < H; ++h) {
      pw[h] = *reinterpret_cast<const float4*>(planes + h * 64 + l16 * 4);
      bw[h] = *reinterpret_cast<const float4*>(buckets + h * 64 + l16 * 4);
    }
  }
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wv; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    int64_t row[R], id[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      row[r] = tile * (4 * R) + r * 4 + grp;
      if (row[r] >= B) row[r] = B - 1;
      id[r] = (ORDER == 1 && tile == (int64_t)blockIdx.x * 4 + wv) ? id0[r] : ids[row[r]];
    }
    float4 u[R], x[R];
    if (ORDER != 1) {
#pragma unroll
      for (int r = 0; r < R; ++r) u[r] = *reinterpret_cast<const float4*>(other + row[r] * 64 + l16 * 4);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) x[r] = *reinterpret_cast<const float4*>(feat + id[r] * 64 + l16 * 4);
    if (ORDER == 1) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        int64_t ro = row[r];
        int lo = (int)id[r];
        asm volatile("" : "+v"(ro) : "v"(lo));  // u's address "depends" on the id: cannot be hoisted above the id wait
        u[r] = *reinterpret_cast<const float4*>(other + ro * 64 + l16 * 4);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      float cnt = 0.f;
      if (STAGE == 0) {
        acc = make_float4(x[r].x + u[r].x, x[r].y + u[r].y, x[r].z + u[r].z, x[r].w + u[r].w);
      } else {
#pragma unroll
        for (int h = 0; h < H; ++h) {
          const float s = row16_sum(dot4_fma(x[r], pw[h], 0.f));
          const float bit = (s < 0.f) ? 0.f : 1.f;
          cnt = cnt + bit;
          if (STAGE >= 2) {
            acc.x = __builtin_fmaf(bit, bw[h].x, acc.x); acc.y = __builtin_fmaf(bit, bw[h].y, acc.y);
            acc.z = __builtin_fmaf(bit, bw[h].z, acc.z); acc.w = __builtin_fmaf(bit, bw[h].w, acc.w);
          }
        }
        if (STAGE == 1) acc = make_float4(cnt + u[r].x, u[r].y, u[r].z, u[r].w);
      }
      if (STAGE >= 3 && ORDER != 3) { acc.x /= cnt; acc.y /= cnt; acc.z /= cnt; acc.w /= cnt; }
      if (STAGE >= 3 && ORDER == 3) {  // Markstein: r = RN(1/c); q = a r; e = fma(-q, c, a); q' = fma(e, r, q)
        const float rc = 1.0f / cnt;
        float q;
        q = acc.x * rc; acc.x = __builtin_fmaf(__builtin_fmaf(-q, cnt, acc.x), rc, q);
        q = acc.y * rc; acc.y = __builtin_fmaf(__builtin_fmaf(-q, cnt, acc.y), rc, q);
        q = acc.z * rc; acc.z = __builtin_fmaf(__builtin_fmaf(-q, cnt, acc.z), rc, q);
        q = acc.w * rc; acc.w = __builtin_fmaf(__builtin_fmaf(-q, cnt, acc.w), rc, q);
      }
      float sp;
      if (STAGE >= 4) sp = dot4_muladd(u[r], acc, 0.f);
      else sp = (acc.x + acc.y) + (acc.z + acc.w) + ((STAGE == 2 || STAGE == 3) ? u[r].x + u[r].y + u[r].z + u[r].w : 0.f);
      const float s = row16_sum(sp);
      if (l16 == 0) score[row[r]] = s;
    }
  }
}

// Full kernel (stage 4) with per-wave wall-clock stamps (100 MHz s_memrealtime): where does a wave
// spend its life?  stamps: 0 start, 1 ids landed, 2 x[0] landed, 3 x[R-1] landed, 4 done.
template <int R>
__global__ __launch_bounds__(256) void stamped_kernel(const int64_t* __restrict__ ids, int64_t B,
                                                      const float* __restrict__ feat, const float* __restrict__ planes,
                                                      const float* __restrict__ buckets, const float* __restrict__ other,
                                                      float* __restrict__ score, unsigned long long* __restrict__ stamps) {
  constexpr int H = 8;
  const int lane = threadIdx.x & 63, l16 = lane & 15, grp = lane >> 4, wv = threadIdx.x >> 6;
  unsigned long long t0 = wall_clock64(), t1 = 0, t2 = 0, t3 = 0;
  float4 pw[H], bw[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    pw[h] = *reinterpret_cast<const float4*>(planes + h * 64 + l16 * 4);
    bw[h] = *reinterpret_cast<const float4*>(buckets + h * 64 + l16 * 4);
  }
  const int64_t tile = (int64_t)blockIdx.x * 4 + wv;
  int64_t row[R], id[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    row[r] = tile * (4 * R) + r * 4 + grp;
    if (row[r] >= B) row[r] = B - 1;
    id[r] = ids[row[r]];
  }
  float4 u[R], x[R];
#pragma unroll
  for (int r = 0; r < R; ++r) u[r] = *reinterpret_cast<const float4*>(other + row[r] * 64 + l16 * 4);
  // force the ids to have landed, then stamp
  long long idsum = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) idsum += id[r];
  asm volatile("" ::"v"(idsum));
  t1 = wall_clock64();
#pragma unroll
  for (int r = 0; r < R; ++r) x[r] = *reinterpret_cast<const float4*>(feat + id[r] * 64 + l16 * 4);
  float tot = 0.f;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    asm volatile("" ::"v"(x[r].x));
    if (r == 0) t2 = wall_clock64();
    if (r == R - 1) t3 = wall_clock64();
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float cnt = 0.f;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const float s = row16_sum(dot4_fma(x[r], pw[h], 0.f));
      const float bit = (s < 0.f) ? 0.f : 1.f;
      cnt = cnt + bit;
      acc.x = __builtin_fmaf(bit, bw[h].x, acc.x); acc.y = __builtin_fmaf(bit, bw[h].y, acc.y);
      acc.z = __builtin_fmaf(bit, bw[h].z, acc.z); acc.w = __builtin_fmaf(bit, bw[h].w, acc.w);
    }
    acc.x /= cnt; acc.y /= cnt; acc.z /= cnt; acc.w /= cnt;
    const float s = row16_sum(dot4_muladd(u[r], acc, 0.f));
    if (l16 == 0) score[row[r]] = s;
    tot += s;
  }
  asm volatile("" ::"v"(tot));
  unsigned long long t4 = wall_clock64();
  if (lane == 0) {
    unsigned long long* o = stamps + 5 * ((size_t)blockIdx.x * 4 + wv);
    o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = t4;
  }
}

template <typename F>
static float time_it(F launch, int iters) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 10; ++i) launch(i);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < iters; ++i) launch(10 + i);
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e3f / iters;
}


// stamps of the PRODUCT kernel's structure (weights through LDS + barrier).  ORDER 0: weights first (as shipped),
// ORDER 1: ids + user rows issued before the weights are staged.
// stamps: 0 start, 1 weights in VGPRs, 2 ids landed, 3 x[0] landed, 4 x[R-1] landed, 5 done.
template <int R, int ORDER>
__global__ __launch_bounds__(256) void stamped2_kernel(const int64_t* __restrict__ ids, int64_t B,
                                                       const float* __restrict__ feat, const float* __restrict__ planes,
                                                       const float* __restrict__ buckets, const float* __restrict__ other,
                                                       float* __restrict__ score, unsigned long long* __restrict__ stamps) {
  constexpr int H = 8;
  __shared__ __attribute__((aligned(16))) float sw[2 * H * 64];
  const int lane = threadIdx.x & 63, l16 = lane & 15, grp = lane >> 4, wv = threadIdx.x >> 6;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wv;
  unsigned long long* o = stamps + 6 * ((size_t)blockIdx.x * 4 + wv);
#define STAMP(k) do { if (lane == 0) __builtin_nontemporal_store(wall_clock64(), o + (k)); } while (0)
  STAMP(0);
  int64_t row[R], id[R];
  float4 u[R], x[R];
  if (ORDER >= 1) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      row[r] = tile * (4 * R) + r * 4 + grp;
      if (row[r] >= B) row[r] = B - 1;
      id[r] = ids[row[r]];
    }
    if (ORDER == 1) {
#pragma unroll
      for (int r = 0; r < R; ++r) u[r] = *reinterpret_cast<const float4*>(other + row[r] * 64 + l16 * 4);
    }
  }
  for (int i = threadIdx.x; i < 2 * H * 16; i += 256) {
    const float* src = (i < H * 16) ? planes + i * 4 : buckets + (i - H * 16) * 4;
    *reinterpret_cast<float4*>(sw + i * 4) = *reinterpret_cast<const float4*>(src);
  }
  __syncthreads();
  float4 pw[H], bw[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    pw[h] = *reinterpret_cast<const float4*>(sw + (h * 16 + l16) * 4);
    bw[h] = *reinterpret_cast<const float4*>(sw + (H * 16 + h * 16 + l16) * 4);
  }
  asm volatile("" ::"v"(pw[0].x), "v"(bw[H - 1].w));
  STAMP(1);
  if (ORDER == 0) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      row[r] = tile * (4 * R) + r * 4 + grp;
      if (row[r] >= B) row[r] = B - 1;
      id[r] = ids[row[r]];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) u[r] = *reinterpret_cast<const float4*>(other + row[r] * 64 + l16 * 4);
  }
  long long idsum = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) idsum += id[r];
  asm volatile("" ::"v"(idsum));
  STAMP(2);
#pragma unroll
  for (int r = 0; r < R; ++r) x[r] = *reinterpret_cast<const float4*>(feat + id[r] * 64 + l16 * 4);
  if (ORDER == 2) {
#pragma unroll
    for (int r = 0; r < R; ++r) u[r] = *reinterpret_cast<const float4*>(other + row[r] * 64 + l16 * 4);
  }
  float tot = 0.f;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    asm volatile("" ::"v"(x[r].x));
    if (r == 0) STAMP(3);
    if (r == R - 1) STAMP(4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float cnt = 0.f;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const float s = row16_sum(dot4_fma(x[r], pw[h], 0.f));
      const float bit = (s < 0.f) ? 0.f : 1.f;
      cnt = cnt + bit;
      acc.x = __builtin_fmaf(bit, bw[h].x, acc.x); acc.y = __builtin_fmaf(bit, bw[h].y, acc.y);
      acc.z = __builtin_fmaf(bit, bw[h].z, acc.z); acc.w = __builtin_fmaf(bit, bw[h].w, acc.w);
    }
    acc.x /= cnt; acc.y /= cnt; acc.z /= cnt; acc.w /= cnt;
    const float s = row16_sum(dot4_muladd(u[r], acc, 0.f));
    if (l16 == 0) score[row[r]] = s;
    tot += s;
  }
  asm volatile("" ::"v"(tot));
  STAMP(5);
#undef STAMP
}


// product-ordered probe with the stamps held in registers (no stores until the end) + where the wave ran
__global__ __launch_bounds__(256, 4) void stamped3_kernel(const int64_t* __restrict__ ids, int64_t B,
                                                         const float* __restrict__ feat, const float* __restrict__ planes,
                                                         const float* __restrict__ buckets, const float* __restrict__ other,
                                                         float* __restrict__ score, unsigned long long* __restrict__ stamps) {
  constexpr int H = 8, R = 4;
  __shared__ __attribute__((aligned(16))) float sw[2 * H * 64];
  const int lane = threadIdx.x & 63, l16 = lane & 15, grp = lane >> 4, wv = threadIdx.x >> 6;
  const unsigned long long t0 = wall_clock64();
  const int64_t tile = (int64_t)blockIdx.x * 4 + wv;
  int64_t row[R], id[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    row[r] = tile * (4 * R) + r * 4 + grp;
    if (row[r] >= B) row[r] = B - 1;
    id[r] = ids[row[r]];
  }
  for (int i = threadIdx.x; i < 2 * H * 16; i += 256) {
    const float* src = (i < H * 16) ? planes + i * 4 : buckets + (i - H * 16) * 4;
    *reinterpret_cast<float4*>(sw + i * 4) = *reinterpret_cast<const float4*>(src);
  }
  __syncthreads();
  float4 pw[H], bw[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    pw[h] = *reinterpret_cast<const float4*>(sw + (h * 16 + l16) * 4);
    bw[h] = *reinterpret_cast<const float4*>(sw + (H * 16 + h * 16 + l16) * 4);
  }
  long long idsum = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) idsum += id[r];
  asm volatile("" ::"v"(idsum));
  const unsigned long long t1 = wall_clock64();
  float4 x[R], u[R];
#pragma unroll
  for (int r = 0; r < R; ++r) x[r] = *reinterpret_cast<const float4*>(feat + id[r] * 64 + l16 * 4);
  asm volatile("" ::: "memory");
#pragma unroll
  for (int r = 0; r < R; ++r) u[r] = *reinterpret_cast<const float4*>(other + row[r] * 64 + l16 * 4);
  unsigned long long t2 = 0;
  float sc = 0.f;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    asm volatile("" ::"v"(x[r].x));
    if (r == 0) t2 = wall_clock64();
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float cnt = 0.f;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const float s = row16_sum(dot4_fma(x[r], pw[h], 0.f));
      const float bit = (s < 0.f) ? 0.f : 1.f;
      cnt = cnt + bit;
      acc.x = __builtin_fmaf(bit, bw[h].x, acc.x); acc.y = __builtin_fmaf(bit, bw[h].y, acc.y);
      acc.z = __builtin_fmaf(bit, bw[h].z, acc.z); acc.w = __builtin_fmaf(bit, bw[h].w, acc.w);
    }
    const float rc = __builtin_amdgcn_rcpf(cnt);
    acc.x *= rc; acc.y *= rc; acc.z *= rc; acc.w *= rc;
    const float s = row16_sum(dot4_muladd(u[r], acc, 0.f));
    if (l16 == r) sc = s;
  }
  if (l16 < 4) score[tile * 16 + l16 * 4 + grp] = sc;
  asm volatile("" ::"v"(sc));
  const unsigned long long t3 = wall_clock64();
  if (lane == 0) {
    unsigned long long* o = stamps + 6 * ((size_t)blockIdx.x * 4 + wv);
    o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3;
    o[4] = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);   // HW_ID bits [15:0]
    o[5] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // XCC_ID bits [3:0]
  }
}

static void run_stamps3(const int64_t* ids, int64_t B, const float* feat, const float* planes2, const float* buckets2,
                        const float* users2, float* score2) {
  const int grid = (int)(B / 64), nw = grid * 4;
  unsigned long long* st;
  CK(hipMalloc(&st, (size_t)nw * 6 * 8));
  for (int i = 0; i < 50; ++i)
    hipLaunchKernelGGL(stamped3_kernel, dim3(grid), dim3(256), 0, 0, ids + (int64_t)i * B, B, feat, planes2, buckets2,
                       users2 + (int64_t)(i % 8) * B * 64, score2, st);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> hs((size_t)nw * 6);
  CK(hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long base = ~0ull, end = 0;
  for (int w = 0; w < nw; ++w) { if (hs[6 * w] < base) base = hs[6 * w]; if (hs[6 * w + 3] > end) end = hs[6 * w + 3]; }
  printf("stamped3 (product order, stamps in registers): first start -> last end = %.2f us\n", (end - base) / 100.0);
  const char* names[4] = {"start", "ids landed", "x[0] landed", "done"};
  for (int k = 0; k < 4; ++k) {
    std::vector<double> v(nw);
    for (int w = 0; w < nw; ++w) v[w] = (hs[6 * w + k] - base) / 100.0;
    std::sort(v.begin(), v.end());
    printf("  %-12s  min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  p99 %6.2f  max %6.2f us\n", names[k], v[0], v[nw / 10], v[nw / 2],
           v[nw * 9 / 10], v[nw * 99 / 100], v[nw - 1]);
  }
  // where do the slow waves sit?  group by (xcc, se, sh, cu) and report waves per CU + done time per CU
  std::map<unsigned, std::vector<double>> percu;
  for (int w = 0; w < nw; ++w) {
    const unsigned hw = (unsigned)hs[6 * w + 4], xcc = (unsigned)hs[6 * w + 5];
    const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    percu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back((hs[6 * w + 3] - base) / 100.0);
  }
  std::map<int, int> hist;
  double worst = 0; unsigned worst_key = 0;
  for (auto& kv : percu) {
    hist[(int)kv.second.size()]++;
    double m = *std::max_element(kv.second.begin(), kv.second.end());
    if (m > worst) { worst = m; worst_key = kv.first; }
  }
  printf("  distinct CUs seen: %zu; waves per CU histogram:", percu.size());
  for (auto& kv : hist) printf("  %d waves x %d CUs", kv.first, kv.second);
  printf("\n  slowest CU key %05x: %zu waves, last done %.2f us\n", worst_key, percu[worst_key].size(), worst);
  // done time as a function of waves per CU
  std::map<int, std::pair<double, int>> bycount;
  for (auto& kv : percu) { auto& e = bycount[(int)kv.second.size()]; e.first += *std::max_element(kv.second.begin(), kv.second.end()); e.second++; }
  for (auto& kv : bycount) printf("  CUs with %2d waves: mean last-done %.2f us\n", kv.first, kv.second.first / kv.second.second);
  CK(hipFree(st));
}

template <int ORDER>
static void run_stamps2(const int64_t* ids, int64_t B, const float* feat, const float* planes2, const float* buckets2,
                        const float* users2, float* score2) {
  const int grid = (int)(B / 64), nw = grid * 4;
  unsigned long long* st;
  CK(hipMalloc(&st, (size_t)nw * 6 * 8));
  for (int i = 0; i < 50; ++i)
    hipLaunchKernelGGL((stamped2_kernel<4, ORDER>), dim3(grid), dim3(256), 0, 0, ids + (int64_t)i * B, B, feat, planes2, buckets2,
                       users2 + (int64_t)(i % 8) * B * 64, score2, st);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> hs((size_t)nw * 6);
  CK(hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long base = ~0ull, end = 0;
  for (int w = 0; w < nw; ++w) { if (hs[6 * w] < base) base = hs[6 * w]; if (hs[6 * w + 5] > end) end = hs[6 * w + 5]; }
  printf("stamped2 ORDER=%d: first start -> last end = %.2f us\n", ORDER, (end - base) / 100.0);
  const char* names[6] = {"start", "weights ready", "ids landed", "x[0] landed", "x[3] landed", "done"};
  for (int k = 0; k < 6; ++k) {
    std::vector<double> v(nw);
    for (int w = 0; w < nw; ++w) v[w] = (hs[6 * w + k] - base) / 100.0;
    std::sort(v.begin(), v.end());
    printf("  %-14s  min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us\n", names[k], v[0], v[nw / 10], v[nw / 2], v[nw * 9 / 10], v[nw - 1]);
  }
  // per-XCD view of the start stamps: blockIdx % 8 is the XCD
  for (int x = 0; x < 8; x += 7) {
    std::vector<double> v;
    for (int w = 0; w < nw; ++w) if ((w / 4) % 8 == x) v.push_back((hs[6 * w + 5] - base) / 100.0);
    std::sort(v.begin(), v.end());
    printf("  XCD %d done: p50 %.2f max %.2f\n", x, v[v.size() / 2], v.back());
  }
  CK(hipFree(st));
}

int main(int argc, char** argv) {
  const int64_t N = 10000000, B = argc > 1 ? atoll(argv[1]) : 65536;
  const int iters = 200, nb = iters + 10;
  float *feat, *out; int64_t* ids;
  CK(hipMalloc(&feat, N * 64 * 4)); CK(hipMalloc(&out, B * 64 * 4)); CK(hipMalloc(&ids, nb * B * 8));
  CK(hipMemset(feat, 0, N * 64 * 4));
  std::vector<int64_t> h(nb * B);
  uint64_t s = 88172645463325252ULL;
  for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (int64_t)(s % N); }
  CK(hipMemcpy(ids, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  printf("B=%lld rows of 256 B, table %.2f GB\n", (long long)B, N * 256 / 1e9);
  if (getenv("MB_LIB_ONLY")) goto lib_cases;
  if (getenv("MB_STAMPS2")) goto stamps2;
  printf("%-44s %8.2f us\n", "empty kernel, 1024 blocks", time_it([&](int) { hipLaunchKernelGGL(empty_kernel, dim3(1024), dim3(256), 0, 0, nullptr); }, iters));
#define RUN(R, MODE, NAME, BLOCKS)                                                                        \
  {                                                                                                       \
    int64_t tiles = (B + 4 * R - 1) / (4 * R);                                                            \
    int grid = (int)((tiles + 3) / 4); if (BLOCKS > 0 && grid > BLOCKS) grid = BLOCKS;                    \
    float us = time_it([&](int i) { hipLaunchKernelGGL((gather_kernel<R, MODE>), dim3(grid), dim3(256), 0, 0, ids + (int64_t)i * B, B, feat, out); }, iters); \
    double bytes = (MODE >= 2) ? B * (8.0 + 256 + 4) : B * (8.0 + 512);                                   \
    printf("%-44s %8.2f us  %7.1f GB/s  grid %d\n", NAME, us, bytes / us / 1e3, grid);                    \
  }
  RUN(1, 0, "copy R=1", 0) RUN(2, 0, "copy R=2", 0) RUN(4, 0, "copy R=4", 0) RUN(8, 0, "copy R=8", 0) RUN(16, 0, "copy R=16", 0)
  RUN(4, 1, "copy nt-store R=4", 0) RUN(8, 1, "copy nt-store R=8", 0)
  RUN(1, 2, "read-only R=1", 0) RUN(2, 2, "read-only R=2", 0) RUN(4, 2, "read-only R=4", 0) RUN(8, 2, "read-only R=8", 0) RUN(16, 2, "read-only R=16", 0)
  RUN(4, 3, "read-only no-indirection R=4", 0) RUN(8, 3, "read-only no-indirection R=8", 0)
  RUN(4, 2, "read-only R=4 grid<=512", 512) RUN(4, 2, "read-only R=4 grid<=256", 256) RUN(8, 2, "read-only R=8 grid<=256", 256)
  RUN(4, 0, "copy R=4 grid<=512", 512)
#define RUNW(R, WORK, DPP, NAME)                                                                         \
  {                                                                                                       \
    int64_t tiles = (B + 4 * R - 1) / (4 * R);                                                            \
    int grid = (int)((tiles + 3) / 4);                                                                    \
    float us = time_it([&](int i) { hipLaunchKernelGGL((work_kernel<R, WORK, DPP>), dim3(grid), dim3(256), 0, 0, ids + (int64_t)i * B, B, feat, out); }, iters); \
    printf("%-44s %8.2f us  (%d VALU/round)\n", NAME, us, WORK * 4);                                       \
  }
  RUNW(4, 0, false, "copy+work R=4 W=0") RUNW(4, 8, false, "copy+work R=4 W=8") RUNW(4, 16, false, "copy+work R=4 W=16")
  RUNW(4, 32, false, "copy+work R=4 W=32") RUNW(4, 64, false, "copy+work R=4 W=64")
  RUNW(4, 16, true, "copy+work(dpp) R=4 W=16") RUNW(4, 32, true, "copy+work(dpp) R=4 W=32")
  RUNW(8, 32, false, "copy+work R=8 W=32") RUNW(2, 32, false, "copy+work R=2 W=32")
  {
    float *planes2, *buckets2, *users2, *score2;
    CK(hipMalloc(&planes2, 8 * 64 * 4)); CK(hipMalloc(&buckets2, 8 * 64 * 4));
    CK(hipMalloc(&users2, 8 * B * 64 * 4)); CK(hipMalloc(&score2, B * 4));
    CK(hipMemset(planes2, 0, 2048)); CK(hipMemset(buckets2, 0, 2048)); CK(hipMemset(users2, 0, 8 * B * 64 * 4));
#define RUNS(STAGE, R, NAME) RUNSO(STAGE, R, 0, NAME)
#define RUNSO(STAGE, R, ORDER, NAME)                                                                      \
    {                                                                                                     \
      int64_t tiles = (B + 4 * R - 1) / (4 * R);                                                          \
      int grid = (int)((tiles + 3) / 4);                                                                  \
      float us = time_it([&](int i) { hipLaunchKernelGGL((staged_kernel<STAGE, R, ORDER>), dim3(grid), dim3(256), 0, 0, ids + (int64_t)i * B, B, feat, planes2, buckets2, users2 + (int64_t)(i % 8) * B * 64, score2); }, iters); \
      printf("%-44s %8.2f us  %7.1f GB/s\n", NAME, us, B * 532.0 / us / 1e3);                              \
    }
    RUNS(0, 4, "staged 0: loads x,u + reduce   R=4") RUNS(1, 4, "staged 1: + 8 projections       R=4")
    RUNS(2, 4, "staged 2: + aggregate           R=4") RUNS(3, 4, "staged 3: + division            R=4")
    RUNS(4, 4, "staged 4: + score (full)        R=4")
    RUNSO(0, 4, 1, "staged 0 ORDER=1 (ids, w, x, u) R=4") RUNSO(4, 4, 1, "staged 4 (full) ORDER=1 R=4")
    RUNSO(4, 2, 1, "staged 4 (full) ORDER=1 R=2") RUNSO(4, 8, 1, "staged 4 (full) ORDER=1 R=8")
    RUNSO(4, 4, 2, "staged 4 (full) weights via LDS R=4") RUNSO(4, 2, 2, "staged 4 (full) weights via LDS R=2")
    RUNSO(4, 8, 2, "staged 4 (full) weights via LDS R=8") RUNSO(1, 4, 2, "staged 1 weights via LDS R=4")
    RUNSO(4, 4, 3, "staged 4 (full) Markstein division R=4") RUNSO(4, 4, 0, "staged 4 (full) IEEE division R=4 (again)")
    RUNSO(4, 4, 3, "staged 4 (full) Markstein division R=4 (again)")
    RUNS(0, 2, "staged 0 R=2") RUNS(4, 2, "staged 4 (full) R=2") RUNS(4, 1, "staged 4 (full) R=1") RUNS(4, 8, "staged 4 (full) R=8")
  }
  stamps2:
  if (getenv("MB_STAMPS2")) {
    float *planes2, *buckets2, *users2, *score2;
    CK(hipMalloc(&planes2, 2048)); CK(hipMalloc(&buckets2, 2048)); CK(hipMalloc(&users2, 8 * B * 64 * 4)); CK(hipMalloc(&score2, B * 4));
    CK(hipMemset(planes2, 0, 2048)); CK(hipMemset(buckets2, 0, 2048)); CK(hipMemset(users2, 0, 8 * B * 64 * 4));
    run_stamps3(ids, B, feat, planes2, buckets2, users2, score2);
    if (getenv("MB_STAMPS3_ONLY")) return 0;
    run_stamps2<0>(ids, B, feat, planes2, buckets2, users2, score2);
    run_stamps2<1>(ids, B, feat, planes2, buckets2, users2, score2);
    run_stamps2<2>(ids, B, feat, planes2, buckets2, users2, score2);
    return 0;
  }
  if (getenv("MB_STAMPS")) {
    float *planes2, *buckets2, *users2, *score2; unsigned long long* st;
    const int grid = (int)(B / 64), nw = grid * 4;
    CK(hipMalloc(&planes2, 2048)); CK(hipMalloc(&buckets2, 2048)); CK(hipMalloc(&users2, B * 64 * 4)); CK(hipMalloc(&score2, B * 4));
    CK(hipMalloc(&st, (size_t)nw * 5 * 8));
    CK(hipMemset(planes2, 0, 2048)); CK(hipMemset(buckets2, 0, 2048)); CK(hipMemset(users2, 0, B * 64 * 4));
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((stamped_kernel<4>), dim3(grid), dim3(256), 0, 0, ids + (int64_t)i * B, B, feat, planes2, buckets2, users2, score2, st);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> hs((size_t)nw * 5);
    CK(hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long base = ~0ull, end = 0;
    for (int w = 0; w < nw; ++w) { if (hs[5 * w] < base) base = hs[5 * w]; if (hs[5 * w + 4] > end) end = hs[5 * w + 4]; }
    printf("stamped kernel: first start -> last end = %.2f us (100 MHz ticks)\n", (end - base) / 100.0);
    const char* names[5] = {"start", "ids landed", "x[0] landed", "x[3] landed", "done"};
    for (int k = 0; k < 5; ++k) {
      std::vector<double> v(nw);
      for (int w = 0; w < nw; ++w) v[w] = (hs[5 * w + k] - base) / 100.0;
      std::sort(v.begin(), v.end());
      printf("  %-12s  min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us\n", names[k], v[0], v[nw / 10], v[nw / 2], v[nw * 9 / 10], v[nw - 1]);
    }
    std::vector<double> d(nw);
    for (int w = 0; w < nw; ++w) d[w] = (hs[5 * w + 4] - hs[5 * w + 3]) / 100.0;
    std::sort(d.begin(), d.end());
    printf("  x[3] landed -> done (compute tail): p10 %.2f p50 %.2f p90 %.2f us\n", d[nw / 10], d[nw / 2], d[nw * 9 / 10]);
    for (int w = 0; w < nw; ++w) d[w] = (hs[5 * w + 3] - hs[5 * w + 2]) / 100.0;
    std::sort(d.begin(), d.end());
    printf("  x[0] landed -> x[3] landed (incl. 3 rounds of compute): p10 %.2f p50 %.2f p90 %.2f us\n", d[nw / 10], d[nw / 2], d[nw * 9 / 10]);
    return 0;
  }
  lib_cases:
  {
    // the product kernels through the C ABI (variant chosen by MI_OOV_LSH64_VARIANT)
    float *planes, *buckets, *users, *score;
    CK(hipMalloc(&planes, 8 * 64 * 4)); CK(hipMalloc(&buckets, 8 * 64 * 4));
    CK(hipMalloc(&users, 8 * B * 64 * 4)); CK(hipMalloc(&score, B * 4));
    std::vector<float> hp(8 * 64), hf(1 << 20);
    for (auto& v : hp) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (float)((int64_t)(s % 2001) - 1000) / 1000.f; }
    CK(hipMemcpy(planes, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(buckets, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    for (auto& v : hf) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (float)((int64_t)(s % 2001) - 1000) / 1000.f; }
    for (int64_t off = 0; off < N * 64; off += (1 << 20)) {
      int64_t n = N * 64 - off < (1 << 20) ? N * 64 - off : (1 << 20);
      CK(hipMemcpy(feat + off, hf.data() + (off / (1 << 20)) % 7, (n - 8) * 4, hipMemcpyHostToDevice));
    }
    const char* var = getenv("MI_OOV_LSH64_VARIANT");
    float us = time_it([&](int i) { mi_oov_lsh_embed(ids + (int64_t)i * B, B, feat, N, 64, planes, 8, buckets, 64, out, nullptr, nullptr); }, iters);
    printf("variant %s: %-32s %8.2f us  %7.1f GB/s\n", var ? var : "0", "mi_oov_lsh_embed", us, B * 520.0 / us / 1e3);
    us = time_it([&](int i) { mi_oov_lsh_embed_score(ids + (int64_t)i * B, B, feat, N, 64, planes, 8, buckets, 64, users + (int64_t)(i % 8) * B * 64, score, nullptr, nullptr); }, iters);
    printf("variant %s: %-32s %8.2f us  %7.1f GB/s\n", var ? var : "0", "mi_oov_lsh_embed_score", us, B * 532.0 / us / 1e3);
    {  // cache-resident inputs: 1024-row table, one id batch, one user buffer -> what is left is launch + VALU
      std::vector<int64_t> hsmall(B);
      for (auto& v : hsmall) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (int64_t)(s % 1024); }
      int64_t* ids_small; CK(hipMalloc(&ids_small, B * 8));
      CK(hipMemcpy(ids_small, hsmall.data(), B * 8, hipMemcpyHostToDevice));
      for (int Hh : {8, 4, 2}) {
        us = time_it([&](int) { mi_oov_lsh_embed_score(ids_small, B, feat, 1024, 64, planes, Hh, buckets, 64, users, score, nullptr, nullptr); }, iters);
        printf("cache-resident inputs, H=%d: mi_oov_lsh_embed_score %8.2f us\n", Hh, us);
      }
      for (int Hh : {8, 4, 2}) {
        us = time_it([&](int i) { mi_oov_lsh_embed_score(ids + (int64_t)i * B, B, feat, N, 64, planes, Hh, buckets, 64, users + (int64_t)(i % 8) * B * 64, score, nullptr, nullptr); }, iters);
        printf("HBM inputs,            H=%d: mi_oov_lsh_embed_score %8.2f us\n", Hh, us);
      }
    }
    {  // how much does cross-launch overlap buy?  same launches, alternating between NS streams
      for (int NS : {2, 3, 4}) {
        hipStream_t st[4];
        for (int k = 0; k < NS; ++k) CK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
        float* sc[4];
        for (int k = 0; k < NS; ++k) CK(hipMalloc(&sc[k], B * 4));
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int i = 0; i < 20; ++i)
          mi_oov_lsh_embed_score(ids + (int64_t)i * B, B, feat, N, 64, planes, 8, buckets, 64, users + (int64_t)(i % 8) * B * 64, sc[i % NS], nullptr, st[i % NS]);
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < iters; ++i)
          mi_oov_lsh_embed_score(ids + (int64_t)(10 + i) * B, B, feat, N, 64, planes, 8, buckets, 64, users + (int64_t)(i % 8) * B * 64, sc[i % NS], nullptr, st[i % NS]);
        CK(hipDeviceSynchronize());
        auto t1 = std::chrono::steady_clock::now();
        double usl = std::chrono::duration<double, std::micro>(t1 - t0).count() / iters;
        printf("%d streams: mi_oov_lsh_embed_score  %8.2f us per launch (wall)  %7.1f GB/s\n", NS, usl, B * 532.0 / usl / 1e3);
      }
    }
    us = time_it([&](int i) { mi_oov_rowdot(users + (int64_t)(i % 8) * B * 64, out, B, 64, score, nullptr); }, iters);
    printf("%-44s %8.2f us  %7.1f GB/s\n", "mi_oov_rowdot", us, B * 516.0 / us / 1e3);
  }
  return 0;
}
