// Row movers and small reductions of the path (all HBM-bound, float4 per lane, 16 lanes per row):
//   gather_rows    nn.Embedding forward                     bpr.py:77-81, single_lsh_embedder.py:100,108
//   splice_rows    in-vocab U OOV scatter of BPR lookups    bpr.py:62-76,108-123
//   gather_mean    knn aggregate                            knn_embedder.py:125-126,146-147
//   col_mean       MeanEmbedder cached mean                 mean_embedder.py:54-56,76-78
//   broadcast_rows mean.repeat(B,1) / zeros(D).repeat(B,1)  mean_embedder.py:56,78; zero_embedder.py:36-60
//   rowdot         BPR.predict                              bpr.py:145-149
#include <stdlib.h>

#include "common.hpp"

namespace mi_oov {

// Rows written once and not read by the launch that writes them leave with NON-TEMPORAL stores: kept out of L2 /
// Infinity Cache they leave those to the gathers (lsh rows, same shape of traffic: 9.8 -> 8.2 us per 65536 lookups;
// a consumer that reads the rows right away pays ~1 us of that back: tools/pair_time.py).
typedef float v4f_nt __attribute__((ext_vector_type(4)));
template <bool VEC>
__device__ __forceinline__ void store4_stream(float* row, int64_t e, int64_t L, float4 v) {
  if constexpr (VEC) {
    if (e < L) __builtin_nontemporal_store(v4f_nt{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f_nt*>(row + e));
  } else {
    store4_guard(row, e, L, v);
  }
}

// One 16-lane group per output row, R rows in flight per group.
// TAB (gather only): ids_src / out_src are DEVICE arrays of K pointers, one per queued batch of B rows each
// (mi_oov_gather_rows_multi); tile t of the launch belongs to batch t / tiles_per_batch.
template <bool VEC, int MODE, bool TAB = false>  // MODE 0: gather_rows, 1: splice_rows
__global__ __launch_bounds__(kBlock) void row_copy_kernel(const void* __restrict__ ids_src,
                                                          const int64_t* __restrict__ rank, int64_t B,
                                                          const float* __restrict__ W, int64_t N,
                                                          const float* __restrict__ oov_rows, int64_t n_oov,
                                                          int64_t D, void* __restrict__ out_src, int64_t K) {
  constexpr int R = 4;
  const int lane = threadIdx.x & 63, l16 = lane & 15, grp = lane >> 4, wv = threadIdx.x >> 6;
  const int64_t tpb = (B + 4 * R - 1) / (4 * R);
  const int64_t ntiles = tpb * (TAB ? K : 1);
  const int dchunks = static_cast<int>((D + 63) / 64);
  for (int64_t gtile = static_cast<int64_t>(blockIdx.x) * 4 + wv; gtile < ntiles;
       gtile += static_cast<int64_t>(gridDim.x) * 4) {
    const int64_t batch = TAB ? gtile / tpb : 0;
    const int64_t tile = TAB ? gtile - batch * tpb : gtile;
    const int64_t* ids = TAB ? reinterpret_cast<const int64_t* const*>(ids_src)[batch] : static_cast<const int64_t*>(ids_src);
    float* out = TAB ? reinterpret_cast<float* const*>(out_src)[batch] : static_cast<float*>(out_src);
    const float* src[R];
    int64_t row[R];
    bool live[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      row[r] = tile * (4 * R) + r * 4 + grp;
      live[r] = row[r] < B;
      src[r] = nullptr;
      if (live[r]) {
        const int64_t id = ids[row[r]];
        if (MODE == 0 || id < N) {
          if (static_cast<uint64_t>(id) < static_cast<uint64_t>(N)) src[r] = W + id * D;
        } else {
          const int64_t k = rank[row[r]];
          if (static_cast<uint64_t>(k) < static_cast<uint64_t>(n_oov)) src[r] = oov_rows + k * D;
        }
      }
    }
    for (int c = 0; c < dchunks; ++c) {
      const int e = (c * 16 + l16) * 4;
      float4 v[R];
#pragma unroll
      for (int r = 0; r < R; ++r)
        v[r] = src[r] ? load4<VEC>(src[r], e, D) : make_float4(qnan(), qnan(), qnan(), qnan());
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (live[r]) store4_stream<VEC>(out + row[r] * D, e, D, v[r]);
    }
  }
}

// out[o,:] = mean_{t in [o*g, min(M,(o+1)*g))} W[idx[t],:], summed in increasing t.
// G2: the reference's g = 2 (knn_embedder.py:126,147) with both rows of an output, and the pairs of R outputs per
// 16-lane group, requested together (2 R gathers in flight per lane instead of one at a time) -- same sums, same order.
// TAB: idx_src / out_src are DEVICE arrays of K pointers, one per queued batch of M indices (mi_oov_gather_mean_multi).
#ifndef MI_GM_R
#define MI_GM_R 2  // outputs per 16-lane group of a SINGLE launch: 65536 outputs in 11.8 us with 2 (and with 1), 12.8 with 4 -- a lone launch wants its 2048 workgroups more than deeper gathers per lane; the queued form keeps 4
#endif
template <bool VEC, bool G2, bool TAB = false>
__global__ __launch_bounds__(kBlock) void gather_mean_kernel(const void* __restrict__ idx_src, int64_t M, int64_t g,
                                                             const float* __restrict__ W, int64_t N, int64_t D,
                                                             void* __restrict__ out_src, int64_t K) {
  const int lane = threadIdx.x & 63, l16 = lane & 15, grp = lane >> 4, wv = threadIdx.x >> 6;
  const int64_t nout = (M + g - 1) / g;
  constexpr int R = G2 ? (TAB ? 4 : MI_GM_R) : 1;  // outputs per 16-lane group and tile
  const int64_t tpb = (nout + 4 * R - 1) / (4 * R);
  const int64_t ntiles = tpb * (TAB ? K : 1);
  const int dchunks = static_cast<int>((D + 63) / 64);
  for (int64_t gtile = static_cast<int64_t>(blockIdx.x) * 4 + wv; gtile < ntiles;
       gtile += static_cast<int64_t>(gridDim.x) * 4) {
    const int64_t batch = TAB ? gtile / tpb : 0;
    const int64_t tile = TAB ? gtile - batch * tpb : gtile;
    const int64_t* idx = TAB ? reinterpret_cast<const int64_t* const*>(idx_src)[batch] : static_cast<const int64_t*>(idx_src);
    float* out = TAB ? reinterpret_cast<float* const*>(out_src)[batch] : static_cast<float*>(out_src);
    if constexpr (G2) {
      int64_t o[R], ia[R], ib[R];
      bool live[R], two[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        o[r] = tile * (4 * R) + r * 4 + grp;
        live[r] = o[r] < nout;
        two[r] = live[r] && 2 * o[r] + 1 < M;  // (the last group of an odd M holds one row: averaged over its own length)
        ia[r] = live[r] ? idx[2 * o[r]] : 0;
        ib[r] = two[r] ? idx[2 * o[r] + 1] : 0;
      }
      for (int c = 0; c < dchunks; ++c) {
        const int e = (c * 16 + l16) * 4;
        float4 va[R], vb[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const bool ina = static_cast<uint64_t>(ia[r]) < static_cast<uint64_t>(N);
          const bool inb = static_cast<uint64_t>(ib[r]) < static_cast<uint64_t>(N);
          va[r] = (live[r] && ina) ? load4<VEC>(W + ia[r] * D, e, D) : make_float4(0.f, 0.f, 0.f, 0.f);
          vb[r] = (two[r] && inb) ? load4<VEC>(W + ib[r] * D, e, D) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (!live[r]) continue;
          const bool ok = static_cast<uint64_t>(ia[r]) < static_cast<uint64_t>(N) &&
                          (!two[r] || static_cast<uint64_t>(ib[r]) < static_cast<uint64_t>(N));
          float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
          acc.x += va[r].x; acc.y += va[r].y; acc.z += va[r].z; acc.w += va[r].w;
          if (two[r]) { acc.x += vb[r].x; acc.y += vb[r].y; acc.z += vb[r].z; acc.w += vb[r].w; }
          const float n = two[r] ? 2.f : 1.f;
          acc.x /= n; acc.y /= n; acc.z /= n; acc.w /= n;
          if (!ok) acc = make_float4(qnan(), qnan(), qnan(), qnan());
          store4_stream<VEC>(out + o[r] * D, e, D, acc);
        }
      }
    } else {
      const int64_t o = tile * 4 + grp;
      if (o >= nout) continue;
      const int64_t t0 = o * g;
      const int64_t t1 = (t0 + g < M) ? t0 + g : M;
      const float inv_n = static_cast<float>(t1 - t0);
      for (int c = 0; c < dchunks; ++c) {
        const int e = (c * 16 + l16) * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        bool ok = true;
        for (int64_t t = t0; t < t1; ++t) {
          const int64_t id = idx[t];
          const bool in = static_cast<uint64_t>(id) < static_cast<uint64_t>(N);
          ok = ok && in;
          const float4 v = in ? load4<VEC>(W + id * D, e, D) : make_float4(0.f, 0.f, 0.f, 0.f);
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        acc.x /= inv_n; acc.y /= inv_n; acc.z /= inv_n; acc.w /= inv_n;
        if (!ok) acc = make_float4(qnan(), qnan(), qnan(), qnan());
        store4_stream<VEC>(out + o * D, e, D, acc);
      }
    }
  }
}

// ---- column mean -------------------------------------------------------------------------------
// Deterministic two-pass order, mirrored by oracle/oov_oracle.c::oov_col_mean:
//   rows are cut in P <= 1024 partitions of RP = col_part_rows(N) rows; inside a partition row-group
//   gr (0..15) sums rows gr, gr+16, ... sequentially and the 16 group sums are added in order;
//   the P partials of a column are then dealt to 64 lanes (p mod 64), each lane adds its partials in
//   increasing p, and the 64 lane sums are combined by a stride-halving tree (32,16,...,1); / N.
__host__ __device__ inline int64_t col_part_rows(int64_t N) {
  int64_t rp = (N + 1023) / 1024;
  rp = (rp + 15) / 16 * 16;
  return rp < 16 ? 16 : rp;
}

template <bool VEC>
__global__ __launch_bounds__(kBlock) void col_partial_kernel(const float* __restrict__ W, int64_t N, int64_t D,
                                                             float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [16][DP]
  const int gr = threadIdx.x >> 4, l16 = threadIdx.x & 15;
  const int dchunks = static_cast<int>((D + 63) / 64);
  const int DP = dchunks * 64;
  const int64_t RP = col_part_rows(N);
  const int64_t P = (N + RP - 1) / RP;
  for (int64_t part = blockIdx.x; part < P; part += gridDim.x) {
    const int64_t r0 = part * RP;
    const int64_t r1 = (r0 + RP < N) ? r0 + RP : N;
    for (int c = 0; c < dchunks; ++c) {
      const int e = (c * 16 + l16) * 4;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int64_t r = r0 + gr; r < r1; r += 16) {
        const float4 v = load4<VEC>(W + r * D, e, D);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      *reinterpret_cast<float4*>(red + gr * DP + e) = acc;
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += kBlock) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) s += red[q * DP + d];
      partial[part * D + d] = s;
    }
    __syncthreads();
  }
}

// one wave per column
__global__ __launch_bounds__(kBlock) void col_final_kernel(const float* __restrict__ partial, int64_t P, int64_t N,
                                                           int64_t D, float* __restrict__ mean) {
  const int lane = threadIdx.x & 63;
  const int64_t d = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6);
  if (d >= D) return;
  float s = 0.f;
  for (int64_t p = lane; p < P; p += 64) s += partial[p * D + d];
#pragma unroll
  for (int stride = 32; stride >= 1; stride >>= 1) s = s + __shfl_xor(s, stride, 64);
  if (lane == 0) mean[d] = s / static_cast<float>(N);
}

template <bool VEC>
__global__ __launch_bounds__(kBlock) void broadcast_kernel(const float* __restrict__ vec, int64_t B, int64_t D,
                                                           float* __restrict__ out) {
  const int lane = threadIdx.x & 63, l16 = lane & 15, grp = lane >> 4, wv = threadIdx.x >> 6;
  const int dchunks = static_cast<int>((D + 63) / 64);
  const int64_t ntiles = (B + 3) / 4;
  for (int c = 0; c < dchunks; ++c) {
    const int e = (c * 16 + l16) * 4;
    const float4 v = vec ? load4<false>(vec, e, D) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t tile = static_cast<int64_t>(blockIdx.x) * 4 + wv; tile < ntiles;
         tile += static_cast<int64_t>(gridDim.x) * 4) {
      const int64_t row = tile * 4 + grp;
      if (row < B) store4_stream<VEC>(out + row * D, e, D, v);
    }
  }
}

template <bool VEC>
__global__ __launch_bounds__(kBlock) void rowdot_kernel(const float* __restrict__ U, const float* __restrict__ E,
                                                        int64_t B, int64_t D, float* __restrict__ score) {
  constexpr int R = 4;
  const int lane = threadIdx.x & 63, l16 = lane & 15, grp = lane >> 4, wv = threadIdx.x >> 6;
  const int64_t ntiles = (B + 4 * R - 1) / (4 * R);
  const int dchunks = static_cast<int>((D + 63) / 64);
  for (int64_t tile = static_cast<int64_t>(blockIdx.x) * 4 + wv; tile < ntiles;
       tile += static_cast<int64_t>(gridDim.x) * 4) {
    float part[R];
    int64_t row[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      row[r] = tile * (4 * R) + r * 4 + grp;
      part[r] = 0.f;
    }
    for (int c = 0; c < dchunks; ++c) {
      const int e = (c * 16 + l16) * 4;
      float4 u[R], v[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const bool live = row[r] < B;
        u[r] = live ? load4<VEC>(U + row[r] * D, e, D) : make_float4(0.f, 0.f, 0.f, 0.f);
        v[r] = live ? load4<VEC>(E + row[r] * D, e, D) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) part[r] = dot4_muladd(u[r], v[r], part[r]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float s = row16_sum(part[r]);
      if (l16 == 0 && row[r] < B) score[row[r]] = s;
    }
  }
}

// ---- lsh backward: gradient of (bits @ W) / popcount with respect to the bucket table W ----------
// grad_W[h, d] = sum_b bits[b,h] * (g[b,d] / cnt[b])  -- what autograd derives for lsh_embedder.py:158,178
// (a cnt = 0 row contributes 0 * (g/0) = NaN, as in the reference, whose trainer then aborts on the NaN
// loss).  Deterministic two-pass reduction over the batch with the row partitioning of col_mean:
// P <= 1024 partitions, 16 row-groups each (rows gr, gr+16, ... in order, then the 16 group sums in
// order), the P partials dealt to 64 lanes and combined by a stride-halving tree.
constexpr int kBwdH = 8;  // planes per pass over the batch

// ONEHOT: the weights are (idx[b] == h) instead of bits[b,h]/cnt[b] -- the slsh backward (one bucket row
// per lookup, single_lsh_embedder.py:87,109) when the bucket table is small.
// g / cnt for a small integer count, correctly rounded: the shared-reciprocal quotient of lsh64.hip's masked_mean
// (q = a rc; q' = fma(fma(-q, cnt, a), rc, q) with rc = RN(1 / cnt) by one Newton step; verified exhaustively against
// IEEE division for cnt = 1..32 and every a except -0 and |a| < 2^-120: tools/check_division.c).  Zeros (the gradient
// rows of in-vocabulary lookups), tiny and infinite values, and counts outside 1..32 take the IEEE division.
__device__ __forceinline__ float4 div4_by_count(float4 a, float cnt) {
  const float y0 = __builtin_amdgcn_rcpf(cnt);
  const float rc = __builtin_fmaf(__builtin_fmaf(-cnt, y0, 1.0f), y0, y0);
  float4 r;
  float q;
  q = a.x * rc; r.x = __builtin_fmaf(__builtin_fmaf(-q, cnt, a.x), rc, q);
  q = a.y * rc; r.y = __builtin_fmaf(__builtin_fmaf(-q, cnt, a.y), rc, q);
  q = a.z * rc; r.z = __builtin_fmaf(__builtin_fmaf(-q, cnt, a.z), rc, q);
  q = a.w * rc; r.w = __builtin_fmaf(__builtin_fmaf(-q, cnt, a.w), rc, q);
  const float amin = fminf(fminf(fabsf(a.x), fabsf(a.y)), fminf(fabsf(a.z), fabsf(a.w)));
  const float amax = fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w)));
  if (!(amin >= 0x1p-100f) || !(amax < __builtin_inff()) || !(cnt >= 1.f) || !(cnt <= 32.f)) {
    r.x = a.x / cnt;
    r.y = a.y / cnt;
    r.z = a.z / cnt;
    r.w = a.w / cnt;
  }
  return r;
}

// FAST8 (H == 8, bits 8-byte aligned, !ONEHOT): the row's code is one 8-byte load, its count the byte sum, the
// division div4_by_count, and the (up to 4) rows of a thread are requested together -- same values, same order of
// additions as the generic loop (22 -> 17 us per backward at B = 65536).
template <bool VEC, bool ONEHOT, bool FAST8 = false>
__global__ __launch_bounds__(kBlock) void lsh_bwd_partial_kernel(const uint8_t* __restrict__ bits,
                                                                 const int64_t* __restrict__ idx,
                                                                 const float* __restrict__ g, int64_t B, int64_t H,
                                                                 int64_t D, int64_t h0, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [16][kBwdH][DP]
  const int gr = threadIdx.x >> 4, l16 = threadIdx.x & 15;
  const int dchunks = static_cast<int>((D + 63) / 64);
  const int DP = dchunks * 64;
  const int64_t RP = col_part_rows(B);
  const int64_t P = (B + RP - 1) / RP;
  const int nh = static_cast<int>((H - h0) < kBwdH ? (H - h0) : kBwdH);
  for (int64_t part = blockIdx.x; part < P; part += gridDim.x) {
    const int64_t r0 = part * RP;
    const int64_t r1 = (r0 + RP < B) ? r0 + RP : B;
    for (int c = 0; c < dchunks; ++c) {
      const int e = (c * 16 + l16) * 4;
      float4 acc[kBwdH];
#pragma unroll
      for (int j = 0; j < kBwdH; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (FAST8) {
        for (int64_t rb = r0 + gr; rb < r1; rb += 64) {  // 4 rows of this thread per round, loads first
          float4 tv[4];
          uint64_t wv[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int64_t r = rb + 16 * q;
            const bool live = r < r1;
            tv[q] = live ? load4<VEC>(g + r * D, e, D) : make_float4(0.f, 0.f, 0.f, 0.f);
            wv[q] = live ? *reinterpret_cast<const uint64_t*>(bits + r * 8) : 0ull;
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (rb + 16 * q >= r1) break;
            uint64_t sum = (wv[q] & 0x00FF00FF00FF00FFull) + ((wv[q] >> 8) & 0x00FF00FF00FF00FFull);
            sum = (sum & 0x0000FFFF0000FFFFull) + ((sum >> 16) & 0x0000FFFF0000FFFFull);
            const float cnt = static_cast<float>(static_cast<uint32_t>(sum) + static_cast<uint32_t>(sum >> 32));
            const float4 t = div4_by_count(tv[q], cnt);
#pragma unroll
            for (int j = 0; j < kBwdH; ++j) {
              const float bit = static_cast<float>(static_cast<uint32_t>(wv[q] >> (8 * j)) & 0xFFu);
              acc[j].x = __builtin_fmaf(bit, t.x, acc[j].x);
              acc[j].y = __builtin_fmaf(bit, t.y, acc[j].y);
              acc[j].z = __builtin_fmaf(bit, t.z, acc[j].z);
              acc[j].w = __builtin_fmaf(bit, t.w, acc[j].w);
            }
          }
        }
      } else
      for (int64_t r = r0 + gr; r < r1; r += 16) {
        float4 t = load4<VEC>(g + r * D, e, D);
        int64_t which = -1;
        if constexpr (ONEHOT) {
          which = idx[r] - h0;
        } else {
          float cnt = 0.f;
          for (int64_t h = 0; h < H; ++h) cnt = cnt + static_cast<float>(bits[r * H + h]);
          t.x = t.x / cnt; t.y = t.y / cnt; t.z = t.z / cnt; t.w = t.w / cnt;
        }
#pragma unroll
        for (int j = 0; j < kBwdH; ++j) {
          if (j < nh) {
            float bit;
            if constexpr (ONEHOT) bit = (which == j) ? 1.f : 0.f;
            else bit = static_cast<float>(bits[r * H + h0 + j]);
            acc[j].x = __builtin_fmaf(bit, t.x, acc[j].x);
            acc[j].y = __builtin_fmaf(bit, t.y, acc[j].y);
            acc[j].z = __builtin_fmaf(bit, t.z, acc[j].z);
            acc[j].w = __builtin_fmaf(bit, t.w, acc[j].w);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < kBwdH; ++j) *reinterpret_cast<float4*>(red + (gr * kBwdH + j) * DP + e) = acc[j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nh * D; i += kBlock) {
      const int j = i / static_cast<int>(D), d = i % static_cast<int>(D);
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) s += red[(q * kBwdH + j) * DP + d];
      partial[(part * kBwdH + j) * D + d] = s;
    }
    __syncthreads();
  }
}

// one wave per (plane, column)
__global__ __launch_bounds__(kBlock) void lsh_bwd_final_kernel(const float* __restrict__ partial, int64_t P, int nh,
                                                               int64_t D, int64_t h0, float* __restrict__ gradW) {
  const int lane = threadIdx.x & 63;
  const int64_t col = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6);
  if (col >= nh * D) return;
  const int64_t j = col / D, d = col % D;
  float s = 0.f;
  for (int64_t p = lane; p < P; p += 64) s += partial[(p * kBwdH + j) * D + d];
#pragma unroll
  for (int stride = 32; stride >= 1; stride >>= 1) s = s + __shfl_xor(s, stride, 64);
  if (lane == 0) gradW[(h0 + j) * D + d] = s;
}

// ---- round 4: the same reduction in ONE launch ----------------------------------------------------------
// The two launches above cost 9.6 + 7.2 us per 65536 lookups (slsh with 9 buckets: two passes over the batch = four
// launches, 28.8 us) for 17 MB of gradient rows: launch latency, and a final pass whose lanes read the partials 2 KB apart.
// Here a workgroup writes the partials of its partitions for ALL planes (a second group of eight planes re-reads the
// partition's 16 KB from L1 / L2), publishes them (release fence + ticket), and the LAST `nfin` workgroups to arrive
// each finish a share of the columns once every partial is there: 16 consecutive columns per chunk (64 contiguous bytes
// of every partial row), thread (lsub, c) adding the partials p = l, l + 64, ... of sum-lanes l = lsub, lsub + 16,
// lsub + 32, lsub + 48 in increasing p, then the oracle's stride-halving tree over the 64 lane sums in LDS.  Same values
// added in the same order as the two-launch form (and as oracle/oov_oracle.c::oov_lsh_embed_backward): bit-identical.
// Partials cross XCDs through device-coherent stores and loads (no fence: a release fence here writes back the whole L2).
// A finisher that is done early spins on the root counter; at most nfin <= 64 workgroups ever wait, each for workgroups
// that are running or still to be dispatched, so the grid always drains.  `counters` (u32[mi_oov_lsh_backward_fused_counters()],
// caller-owned) must be zero at launch and is zero again when the kernel ends: the last finisher resets it.
constexpr int kFinCols = 16, kFinMax = 64;
constexpr int kFinSub = 16, kCtrStride = 64;  // arrival counters, words between them; + the finishers' own counter behind them

// (four workgroups per CU -- 128 registers -- so that the 1024 partitions of a 65536-row batch are ONE round: left to itself
//  the compiler gave the finishers' 64 loads in flight 64-bit addresses each, 340 registers, one workgroup per CU, four rounds)
template <bool VEC, bool ONEHOT, bool FAST8>
__global__ __launch_bounds__(kBlock, 4) void lsh_bwd_fused_kernel(const uint8_t* __restrict__ bits, const int64_t* __restrict__ idx,
                                                               const float* __restrict__ g, int64_t B, int64_t H, int64_t D,
                                                               float* partial, unsigned* counters, int nfin,
                                                               float* __restrict__ gradW) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [16][kBwdH][DP]; the finishers' [64][kFinCols] lane sums
  const int gr = threadIdx.x >> 4, l16 = threadIdx.x & 15;
  const int dchunks = static_cast<int>((D + 63) / 64);
  const int DP = dchunks * 64;
  const int64_t RP = col_part_rows(B);
  const int64_t P = (B + RP - 1) / RP;
  const int64_t HP = (H + kBwdH - 1) / kBwdH * kBwdH;  // planes per partial row, padded to whole groups of eight
  for (int64_t part = blockIdx.x; part < P; part += gridDim.x) {
    const int64_t r0 = part * RP;
    const int64_t r1 = (r0 + RP < B) ? r0 + RP : B;
    for (int64_t h0 = 0; h0 < H; h0 += kBwdH) {
      const int nh = static_cast<int>((H - h0) < kBwdH ? (H - h0) : kBwdH);
      for (int c = 0; c < dchunks; ++c) {
        const int e = (c * 16 + l16) * 4;
        float4 acc[kBwdH];
#pragma unroll
        for (int j = 0; j < kBwdH; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int64_t rb = r0 + gr; rb < r1; rb += 64) {  // 4 rows of this thread per round, every load first
          float4 tv[4];
          uint64_t wv[4];   // FAST8: the row's eight bits; ONEHOT: its bucket index relative to h0
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int64_t r = rb + 16 * q;
            const bool live = r < r1;
            tv[q] = live ? load4<VEC>(g + r * D, e, D) : make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (FAST8) wv[q] = live ? *reinterpret_cast<const uint64_t*>(bits + r * 8) : 0ull;
            else if constexpr (ONEHOT) wv[q] = live ? static_cast<uint64_t>(idx[r] - h0) : ~0ull;
            else wv[q] = 0ull;
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int64_t r = rb + 16 * q;
            if (r >= r1) break;
            float4 t = tv[q];
            if constexpr (FAST8) {
              uint64_t sum = (wv[q] & 0x00FF00FF00FF00FFull) + ((wv[q] >> 8) & 0x00FF00FF00FF00FFull);
              sum = (sum & 0x0000FFFF0000FFFFull) + ((sum >> 16) & 0x0000FFFF0000FFFFull);
              const float cnt = static_cast<float>(static_cast<uint32_t>(sum) + static_cast<uint32_t>(sum >> 32));
              t = div4_by_count(t, cnt);
            } else if constexpr (!ONEHOT) {
              float cnt = 0.f;
              for (int64_t h = 0; h < H; ++h) cnt = cnt + static_cast<float>(bits[r * H + h]);
              t.x = t.x / cnt; t.y = t.y / cnt; t.z = t.z / cnt; t.w = t.w / cnt;
            }
#pragma unroll
            for (int j = 0; j < kBwdH; ++j) {
              if (j < nh) {
                float bit;
                if constexpr (FAST8) bit = static_cast<float>(static_cast<uint32_t>(wv[q] >> (8 * j)) & 0xFFu);
                else if constexpr (ONEHOT) bit = (wv[q] == static_cast<uint64_t>(j)) ? 1.f : 0.f;
                else bit = static_cast<float>(bits[r * H + h0 + j]);
                acc[j].x = __builtin_fmaf(bit, t.x, acc[j].x);
                acc[j].y = __builtin_fmaf(bit, t.y, acc[j].y);
                acc[j].z = __builtin_fmaf(bit, t.z, acc[j].z);
                acc[j].w = __builtin_fmaf(bit, t.w, acc[j].w);
              }
            }
          }
        }
#pragma unroll
        for (int j = 0; j < kBwdH; ++j) *reinterpret_cast<float4*>(red + (gr * kBwdH + j) * DP + e) = acc[j];
      }
      __syncthreads();
      for (int i = threadIdx.x; i < nh * D; i += kBlock) {
        const int j = i / static_cast<int>(D), d = i % static_cast<int>(D);
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += red[(q * kBwdH + j) * DP + d];
        // device-coherent (write-through) store: a finisher on another XCD reads it, and a release FENCE on this chip is a
        // write-back of the whole L2 (122 us per launch when every workgroup issued one) -- no fence anywhere below
        if (nfin > 0) __hip_atomic_store(&partial[(part * HP + h0 + j) * D + d], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else partial[(part * HP + h0 + j) * D + d] = s;  // (two-launch form: the final reduction is the next kernel)
      }
      __syncthreads();
    }
  }
  // publish: the partial stores of every thread are acknowledged (vmcnt) before the workgroup reports in.  Arrivals are
  // counted on kFinSub counters (workgroup b on counter b mod kFinSub, 256 bytes apart): one address takes ~90 atomics per us.
  if (nfin <= 0) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned G = gridDim.x;
  unsigned* root = counters + kFinSub * kCtrStride;
  if (threadIdx.x == 0)
    __hip_atomic_fetch_add(counters + (blockIdx.x % kFinSub) * kCtrStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (blockIdx.x + static_cast<unsigned>(nfin) < G) return;  // the finishers are the nfin workgroups dispatched last
  const int f = static_cast<int>(blockIdx.x - (G - static_cast<unsigned>(nfin)));
  // thread r < kFinSub of a finisher waits for counter r to hold all of its members (the workgroups b < G with b mod kFinSub
  // == r): the finishers read the arrival counters themselves -- a root counter bumped by each counter's last arrival was
  // one more memory round trip on the critical path
  if (threadIdx.x < kFinSub && threadIdx.x < G) {
    const unsigned members = (G - 1u - threadIdx.x) / kFinSub + 1u;
    while (__hip_atomic_load(counters + threadIdx.x * kCtrStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < members)
      __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
  const int64_t cols = H * D;  // flattened (plane, column): plane h, column d at h * D + d of a partial row of HP * D floats
  const int64_t nchunks = (cols + kFinCols - 1) / kFinCols;
  float* lanes = red;  // [64][kFinCols]
  const int cl = threadIdx.x & (kFinCols - 1), lsub = threadIdx.x >> 4;
  for (int64_t ch = f; ch < nchunks; ch += nfin) {
    const int64_t col = ch * kFinCols + cl;
    // address = (wave-uniform row of the partials) + (a 32-bit per-thread offset): one register per load in flight, not three
    const uint32_t toff = static_cast<uint32_t>(static_cast<int64_t>(lsub) * HP * D + (col < cols ? col : 0));
    // device-coherent loads (never a line this XCD's L2 kept from an earlier launch), 16 partials of each of the thread's
    // four sum-lanes requested together -- issued one by one behind their additions they were a chain of 64 memory round
    // trips per thread (39 us per launch) -- then added in increasing p
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t pb = 0; pb < P; pb += 64 * 16) {
      float x[4][16];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int64_t prow = pb + static_cast<int64_t>(i) * 64 + 16 * q;  // uniform; the thread's partial is prow + lsub
          const float* rowbase = partial + (prow < P ? prow : 0) * HP * D;
          x[q][i] = (prow + lsub < P) ? __hip_atomic_load(rowbase + toff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
        }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (pb + static_cast<int64_t>(i) * 64 + lsub + 16 * q < P) s4[q] += x[q][i];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) lanes[(lsub + 16 * q) * kFinCols + cl] = s4[q];
    __syncthreads();
    for (int stride = 32; stride >= 1; stride >>= 1) {
      for (int i = threadIdx.x; i < stride * kFinCols; i += kBlock) lanes[i] = lanes[i] + lanes[i + stride * kFinCols];
      __syncthreads();
    }
    if (threadIdx.x < kFinCols && col < cols) gradW[col] = lanes[cl];
    __syncthreads();
  }
  if (threadIdx.x == 0) {  // the last finisher leaves the counters at zero for the next launch
    const unsigned done = __hip_atomic_fetch_add(root, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // finishers that are done
    if (done + 1u == static_cast<unsigned>(nfin)) {  // every finisher is past its wait: zero for the next launch
      // (read-modify-writes, not stores: performed where every XCD's atomics are; a plain store could sit in this XCD's L2)
      for (int i = 0; i < kFinSub + 1; ++i) __hip_atomic_fetch_and(counters + i * kCtrStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// The final reduction as a launch of its own, in the finishers' layout (16 consecutive columns per workgroup, 64 partials
// per thread requested together, the tree in LDS): what follows `lsh_bwd_fused_kernel` launched with nfin = 0, which then
// only writes the partials of every plane (plain stores: a kernel boundary lies between the two).
__global__ __launch_bounds__(kBlock) void lsh_bwd_final_cols_kernel(const float* __restrict__ partial, int64_t P, int64_t HPD, int64_t cols,
                                                                    float* __restrict__ gradW) {
  __shared__ float lanes[64 * kFinCols];
  const int cl = threadIdx.x & (kFinCols - 1), lsub = threadIdx.x >> 4;
  const int64_t col = static_cast<int64_t>(blockIdx.x) * kFinCols + cl;
  const uint32_t toff = static_cast<uint32_t>(static_cast<int64_t>(lsub) * HPD + (col < cols ? col : 0));
  float s4[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t pb = 0; pb < P; pb += 64 * 16) {
    float x[4][16];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int64_t prow = pb + static_cast<int64_t>(i) * 64 + 16 * q;
        const float* rowbase = partial + (prow < P ? prow : 0) * HPD;
        x[q][i] = (prow + lsub < P) ? rowbase[toff] : 0.f;
      }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (pb + static_cast<int64_t>(i) * 64 + lsub + 16 * q < P) s4[q] += x[q][i];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) lanes[(lsub + 16 * q) * kFinCols + cl] = s4[q];
  __syncthreads();
  for (int stride = 32; stride >= 1; stride >>= 1) {
    for (int i = threadIdx.x; i < stride * kFinCols; i += kBlock) lanes[i] = lanes[i] + lanes[i + stride * kFinCols];
    __syncthreads();
  }
  if (threadIdx.x < kFinCols && col < cols) gradW[col] = lanes[cl];
}

// ---- context models: fused-table token gather with the user/item OOV splice ---------------------
// InductiveContextRecommender.embed_token_fields (abstract_recommender.py:794-842) and
// InductiveFMFirstOrderLinear.embed_token_fields (layers.py:1634-1693): row (b, f) is
// table[tokens[b,f] + offsets[f]] except that field 0 / 1 take the plugin (or OOV bucket) row of an
// out-of-vocabulary user / item.  The reference zeroes those ids, gathers, and overwrites.
struct TokenArgs {
  const int64_t* tokens;  // [B, nf]
  int64_t B, nf;
  const int64_t* offsets;  // [nf]
  const float* table;      // [T, D]
  int64_t T, D;
  int64_t n_users, n_items;
  const float* oov_u;  // [n_oov_u, D] rows of the OOV users in order of appearance
  const int64_t* rank_u;  // [B] rank of row b among the OOV users
  int64_t n_oov_u;
  const float* oov_i;
  const int64_t* rank_i;
  int64_t n_oov_i;
  float* out;  // [B, nf, D], or [B, D] when summing over the fields
};

__device__ __forceinline__ const float* token_row(const TokenArgs& a, int64_t b, int64_t f) {
  const int64_t tok = a.tokens[b * a.nf + f];
  if (f == 0 && tok >= a.n_users) {
    const int64_t k = a.rank_u[b];
    return (static_cast<uint64_t>(k) < static_cast<uint64_t>(a.n_oov_u)) ? a.oov_u + k * a.D : nullptr;
  }
  if (f == 1 && tok >= a.n_items) {
    const int64_t k = a.rank_i[b];
    return (static_cast<uint64_t>(k) < static_cast<uint64_t>(a.n_oov_i)) ? a.oov_i + k * a.D : nullptr;
  }
  const int64_t r = tok + a.offsets[f];
  return (tok >= 0 && static_cast<uint64_t>(r) < static_cast<uint64_t>(a.T)) ? a.table + r * a.D : nullptr;
}

// second order: one 16-lane group per (b, f) row, float4 per lane, R rows in flight
template <bool VEC>
__global__ __launch_bounds__(kBlock) void token_fields_kernel(TokenArgs a) {
  constexpr int R = 4;
  const int lane = threadIdx.x & 63, l16 = lane & 15, grp = lane >> 4, wv = threadIdx.x >> 6;
  const int64_t rows = a.B * a.nf;
  const int64_t ntiles = (rows + 4 * R - 1) / (4 * R);
  const int dchunks = static_cast<int>((a.D + 63) / 64);
  for (int64_t tile = static_cast<int64_t>(blockIdx.x) * 4 + wv; tile < ntiles;
       tile += static_cast<int64_t>(gridDim.x) * 4) {
    const float* src[R];
    int64_t row[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      row[r] = tile * (4 * R) + r * 4 + grp;
      src[r] = (row[r] < rows) ? token_row(a, row[r] / a.nf, row[r] % a.nf) : nullptr;
    }
    for (int c = 0; c < dchunks; ++c) {
      const int e = (c * 16 + l16) * 4;
      float4 v[R];
#pragma unroll
      for (int r = 0; r < R; ++r)
        v[r] = src[r] ? load4<VEC>(src[r], e, a.D) : make_float4(qnan(), qnan(), qnan(), qnan());
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (row[r] < rows) store4_stream<VEC>(a.out + row[r] * a.D, e, a.D, v[r]);
    }
  }
}

// first order: out[b, d] = sum over the fields, in field order (torch.sum(dim=1) of [B, nf, D]); D is
// the model's output_dim (1), so one thread per (b, d)
__global__ __launch_bounds__(kBlock) void token_fields_sum_kernel(TokenArgs a) {
  const int64_t total = a.B * a.D;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t b = i / a.D, d = i - b * a.D;
    float acc = 0.f;
    bool ok = true;
    for (int64_t f = 0; f < a.nf; ++f) {
      const float* src = token_row(a, b, f);
      ok = ok && (src != nullptr);
      acc = acc + (src ? src[d] : 0.f);
    }
    a.out[i] = ok ? acc : qnan();
  }
}

}  // namespace mi_oov

using namespace mi_oov;

extern "C" int64_t mi_oov_lsh_backward_workspace(int64_t B, int64_t H, int64_t D) {
  if (B <= 0 || H <= 0 || D <= 0) return 0;
  const int64_t RP = col_part_rows(B);
  return ((B + RP - 1) / RP) * kBwdH * D;  // floats
}

// rows of `g` added into out[idx[m]] with hardware float atomics (order not fixed); idx outside [0,N) skipped.
// A WAVE owns a row and lane l adds element l, l + 64, ...: one wave-instruction then covers 256 CONTIGUOUS bytes of
// the destination row, the shape the memory-side atomic units take at full rate (MI355X_MICROARCH.md, "Global float
// atomics": each 256-B wave-instruction leaves L2 as four 64-B atomic requests, ~1.3 TB/s of added bytes chip-wide).
// The first version gave a row to 16 lanes with 4 consecutive floats each: an instruction then touched every fourth
// dword of four rows -- sixteen 64-B requests a quarter full -- and ran at 0.28 TB/s of added bytes (60 us for 65536
// rows of 64 floats); this one: see DESIGN.md section 5.  Four rows per wave are in flight.
__global__ __launch_bounds__(kBlock) void scatter_add_kernel(const int64_t* __restrict__ idx, int64_t M,
                                                             const float* __restrict__ g, int64_t N, int64_t D,
                                                             float* __restrict__ out) {
  constexpr int R = 4;
  const int lane = threadIdx.x & 63;
  const int64_t wave = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6);
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * (kBlock / 64);
  for (int64_t m0 = wave * R; m0 < M; m0 += nwaves * R) {
    int64_t r[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int64_t m = m0 + j;
      r[j] = (m < M) ? idx[m] : -1;
      if (r[j] >= N) r[j] = -1;
    }
    for (int64_t e = lane; e < D; e += 64) {
      float v[R];
#pragma unroll
      for (int j = 0; j < R; ++j) v[j] = (r[j] >= 0) ? g[(m0 + j) * D + e] : 0.f;
#pragma unroll
      for (int j = 0; j < R; ++j)
        if (r[j] >= 0) unsafeAtomicAdd(out + r[j] * D + e, v[j]);
    }
  }
}

extern "C" int mi_oov_scatter_add_rows(const int64_t* idx, int64_t M, const float* g, int64_t N, int64_t D, float* out,
                                       void* stream) {
  if (M < 0 || N < 0 || D <= 0) return MI_OOV_ERR_SHAPE;
  if (M == 0) return MI_OOV_OK;
  if (!idx || !g || !out) return MI_OOV_ERR_NULL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned grid = grid_for(M, (kBlock / 64) * 4);
  hipLaunchKernelGGL(scatter_add_kernel, dim3(grid), dim3(kBlock), 0, st, idx, M, g, N, D, out);
  return check_launch();
}

static int run_lsh_bwd(const uint8_t* bits, const int64_t* idx, const float* grad_out, int64_t B, int64_t H, int64_t D,
                       float* grad_buckets, float* workspace, hipStream_t st);

extern "C" int mi_oov_slsh_embed_backward(const int64_t* idx, const float* grad_out, int64_t B, int64_t n_buckets,
                                          int64_t D, float* grad_buckets, float* workspace, void* stream) {
  if (B < 0 || n_buckets <= 0 || D <= 0) return MI_OOV_ERR_SHAPE;
  if (!grad_buckets) return MI_OOV_ERR_NULL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n_buckets <= 64 && D <= 256) return run_lsh_bwd(nullptr, idx, grad_out, B, n_buckets, D, grad_buckets, workspace, st);
  if (hipMemsetAsync(grad_buckets, 0, static_cast<size_t>(n_buckets * D) * sizeof(float), st) != hipSuccess) {
    check_launch();
    return MI_OOV_ERR_LAUNCH;
  }
  return mi_oov_scatter_add_rows(idx, B, grad_out, n_buckets, D, grad_buckets, stream);
}

extern "C" int mi_oov_lsh_embed_backward(const uint8_t* bits, const float* grad_out, int64_t B, int64_t H, int64_t D,
                                         float* grad_buckets, float* workspace, void* stream) {
  return run_lsh_bwd(bits, nullptr, grad_out, B, H, D, grad_buckets, workspace, static_cast<hipStream_t>(stream));
}

// one launch: partials + last-workgroups-done final reduction (lsh_bwd_fused_kernel)
extern "C" int64_t mi_oov_lsh_backward_fused_workspace(int64_t B, int64_t H, int64_t D) {
  if (B <= 0 || H <= 0 || D <= 0) return 0;
  const int64_t RP = col_part_rows(B);
  return ((B + RP - 1) / RP) * ((H + kBwdH - 1) / kBwdH * kBwdH) * D;  // floats
}

extern "C" int64_t mi_oov_lsh_backward_fused_counters(void) { return (kFinSub + 1) * kCtrStride; }  // 32-bit words

static int run_lsh_bwd_fused(const uint8_t* bits, const int64_t* idx, const float* grad_out, int64_t B, int64_t H, int64_t D,
                             float* grad_buckets, float* workspace, uint32_t* counters, hipStream_t st) {
  if (B < 0 || H <= 0 || D <= 0 || D > 256) return MI_OOV_ERR_SHAPE;
  if (!grad_buckets) return MI_OOV_ERR_NULL;
  if (B == 0) {
    if (hipMemsetAsync(grad_buckets, 0, static_cast<size_t>(H * D) * sizeof(float), st) != hipSuccess) {
      check_launch();
      return MI_OOV_ERR_LAUNCH;
    }
    return MI_OOV_OK;
  }
  if ((!bits && !idx) || !grad_out || !workspace) return MI_OOV_ERR_NULL;  // counters == NULL: partials, then a final launch
  const int64_t RP = col_part_rows(B);
  const int64_t P = (B + RP - 1) / RP;
  const int dchunks = static_cast<int>((D + 63) / 64);
  size_t lds = static_cast<size_t>(16) * kBwdH * dchunks * 64 * sizeof(float);
  if (lds < 64 * kFinCols * sizeof(float)) lds = 64 * kFinCols * sizeof(float);
  const bool vec = (D % 4 == 0) && aligned16(grad_out);
  const bool fast8 = !idx && H == 8 && vec && (reinterpret_cast<uintptr_t>(bits) & 7u) == 0;
  const int64_t nchunks = (H * D + kFinCols - 1) / kFinCols;
  int nfin = static_cast<int>(nchunks < kFinMax ? nchunks : kFinMax);
  if (nfin > P) nfin = static_cast<int>(P);
  if (!counters) nfin = 0;
  auto k = idx ? (vec ? lsh_bwd_fused_kernel<true, true, false> : lsh_bwd_fused_kernel<false, true, false>)
               : fast8 ? lsh_bwd_fused_kernel<true, false, true>
                       : (vec ? lsh_bwd_fused_kernel<true, false, false> : lsh_bwd_fused_kernel<false, false, false>);
  if (int rc = set_lds(k, lds)) return rc;
  hipLaunchKernelGGL(k, dim3(static_cast<unsigned>(P)), dim3(kBlock), lds, st, bits, idx, grad_out, B, H, D, workspace, counters, nfin,
                     grad_buckets);
  if (int rc = check_launch()) return rc;
  if (nfin > 0) return MI_OOV_OK;
  const int64_t HP = (H + kBwdH - 1) / kBwdH * kBwdH;
  hipLaunchKernelGGL(lsh_bwd_final_cols_kernel, dim3(static_cast<unsigned>(nchunks)), dim3(kBlock), 0, st, workspace, P, HP * D, H * D, grad_buckets);
  return check_launch();
}

extern "C" int mi_oov_lsh_embed_backward_fused(const uint8_t* bits, const float* grad_out, int64_t B, int64_t H, int64_t D,
                                               float* grad_buckets, float* workspace, uint32_t* counters, void* stream) {
  return run_lsh_bwd_fused(bits, nullptr, grad_out, B, H, D, grad_buckets, workspace, counters, static_cast<hipStream_t>(stream));
}

extern "C" int mi_oov_slsh_embed_backward_fused(const int64_t* idx, const float* grad_out, int64_t B, int64_t n_buckets, int64_t D,
                                                float* grad_buckets, float* workspace, uint32_t* counters, void* stream) {
  if (B < 0 || n_buckets <= 0 || D <= 0) return MI_OOV_ERR_SHAPE;
  if (!grad_buckets) return MI_OOV_ERR_NULL;
  if (n_buckets <= 64 && D <= 256)
    return run_lsh_bwd_fused(nullptr, idx, grad_out, B, n_buckets, D, grad_buckets, workspace, counters, static_cast<hipStream_t>(stream));
  return mi_oov_slsh_embed_backward(idx, grad_out, B, n_buckets, D, grad_buckets, workspace, stream);  // memset + float atomics
}

static int run_lsh_bwd(const uint8_t* bits, const int64_t* idx, const float* grad_out, int64_t B, int64_t H, int64_t D,
                       float* grad_buckets, float* workspace, hipStream_t st) {
  if (B < 0 || H <= 0 || D <= 0 || D > 256) return MI_OOV_ERR_SHAPE;
  if (!grad_buckets) return MI_OOV_ERR_NULL;
  if (B == 0) {
    if (hipMemsetAsync(grad_buckets, 0, static_cast<size_t>(H * D) * sizeof(float), st) != hipSuccess) {
      check_launch();
      return MI_OOV_ERR_LAUNCH;
    }
    return MI_OOV_OK;
  }
  if ((!bits && !idx) || !grad_out || !workspace) return MI_OOV_ERR_NULL;
  const int64_t RP = col_part_rows(B);
  const int64_t P = (B + RP - 1) / RP;
  const int dchunks = static_cast<int>((D + 63) / 64);
  const size_t lds = static_cast<size_t>(16) * kBwdH * dchunks * 64 * sizeof(float);
  const bool vec = (D % 4 == 0) && aligned16(grad_out);
  for (int64_t h0 = 0; h0 < H; h0 += kBwdH) {
    const int nh = static_cast<int>((H - h0) < kBwdH ? (H - h0) : kBwdH);
    const bool fast8 = !idx && H == 8 && vec && (reinterpret_cast<uintptr_t>(bits) & 7u) == 0;
    auto k = idx ? (vec ? lsh_bwd_partial_kernel<true, true> : lsh_bwd_partial_kernel<false, true>)
                 : fast8 ? lsh_bwd_partial_kernel<true, false, true>
                         : (vec ? lsh_bwd_partial_kernel<true, false> : lsh_bwd_partial_kernel<false, false>);
    if (int rc = set_lds(k, lds)) return rc;
    hipLaunchKernelGGL(k, dim3(static_cast<unsigned>(P)), dim3(kBlock), lds, st, bits, idx, grad_out, B, H, D, h0, workspace);
    if (int rc = check_launch()) return rc;
    const int cols = nh * static_cast<int>(D);
    hipLaunchKernelGGL(lsh_bwd_final_kernel, dim3((cols + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, st, workspace,
                       P, nh, D, h0, grad_buckets);
    if (int rc = check_launch()) return rc;
  }
  return MI_OOV_OK;
}

extern "C" int mi_oov_token_fields_embed(const int64_t* tokens, int64_t B, int64_t nf, const int64_t* offsets,
                                         const float* table, int64_t T, int64_t D, int64_t n_users, int64_t n_items,
                                         const float* oov_user_rows, const int64_t* user_rank, int64_t n_oov_users,
                                         const float* oov_item_rows, const int64_t* item_rank, int64_t n_oov_items,
                                         int sum_fields, float* out, void* stream) {
  if (B < 0 || nf <= 0 || T <= 0 || D <= 0 || n_oov_users < 0 || n_oov_items < 0) return MI_OOV_ERR_SHAPE;
  if (B == 0) return MI_OOV_OK;
  if (!tokens || !offsets || !table || !out) return MI_OOV_ERR_NULL;
  if ((n_oov_users > 0 && (!oov_user_rows || !user_rank)) || (n_oov_items > 0 && (!oov_item_rows || !item_rank)))
    return MI_OOV_ERR_NULL;
  TokenArgs a{tokens, B, nf, offsets, table, T, D, n_users, n_items, oov_user_rows, user_rank, n_oov_users,
              oov_item_rows, item_rank, n_oov_items, out};
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (sum_fields) {
    hipLaunchKernelGGL(token_fields_sum_kernel, dim3(grid_for(B * D, kBlock)), dim3(kBlock), 0, st, a);
    return check_launch();
  }
  const bool vec = (D % 4 == 0) && aligned16(table) && aligned16(out) && (!oov_user_rows || aligned16(oov_user_rows)) &&
                   (!oov_item_rows || aligned16(oov_item_rows));
  const int grid = grid_for(B * nf, 64);
  if (vec)
    hipLaunchKernelGGL(token_fields_kernel<true>, dim3(grid), dim3(kBlock), 0, st, a);
  else
    hipLaunchKernelGGL(token_fields_kernel<false>, dim3(grid), dim3(kBlock), 0, st, a);
  return check_launch();
}

namespace mi_oov {
int launch_gather_mean64_persistent(const int64_t* const* idx_tab, float* const* out_tab, int64_t K, int64_t M,
                                    const float* W, int64_t N, hipStream_t st);  // lsh64p.hip
static bool persist_movers() {
  static const bool on = env_knob("MI_OOV_PERSIST_MOVERS", 1, 0, 1) != 0;  // developer A/B knob: 0 = the grid-stride kernels for every width
  return on;
}
}  // namespace mi_oov

extern "C" int mi_oov_gather_rows(const int64_t* ids, int64_t B, const float* W, int64_t N, int64_t D, float* out,
                                  void* stream) {
  if (B < 0 || N <= 0 || D <= 0) return MI_OOV_ERR_SHAPE;
  if (B == 0) return MI_OOV_OK;
  if (!ids || !W || !out) return MI_OOV_ERR_NULL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool vec = (D % 4 == 0) && aligned16(W) && aligned16(out);
  const int grid = grid_for(B, 64);
  if (vec)
    hipLaunchKernelGGL((row_copy_kernel<true, 0>), dim3(grid), dim3(kBlock), 0, st, ids, nullptr, B, W, N, nullptr, 0, D, out, 1);
  else
    hipLaunchKernelGGL((row_copy_kernel<false, 0>), dim3(grid), dim3(kBlock), 0, st, ids, nullptr, B, W, N, nullptr, 0, D, out, 1);
  return check_launch();
}

// K queued batches of mi_oov_gather_rows in one launch (grid-stride over the tiles of all batches): ids_tab / out_tab are
// DEVICE arrays of K device pointers (int64[B] / f32[B,D], 16-byte aligned rows), every batch B rows.
extern "C" int mi_oov_gather_rows_multi(const int64_t* const* ids_tab, float* const* out_tab, int64_t K, int64_t B,
                                        const float* W, int64_t N, int64_t D, void* stream) {
  if (K < 0 || B < 0 || N <= 0 || D <= 0) return MI_OOV_ERR_SHAPE;
  if (K == 0 || B == 0) return MI_OOV_OK;
  if (!ids_tab || !out_tab || !W) return MI_OOV_ERR_NULL;
  if (D % 4 != 0) return MI_OOV_ERR_SHAPE;  // rows of whole float4 (callers fall back to K single launches)
  if (!aligned16(W) || (reinterpret_cast<uintptr_t>(ids_tab) & 7u) || (reinterpret_cast<uintptr_t>(out_tab) & 7u)) return MI_OOV_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int grid = grid_for(K * ((B + 15) / 16) * 16, 64);
  hipLaunchKernelGGL((row_copy_kernel<true, 0, true>), dim3(grid), dim3(kBlock), 0, st, static_cast<const void*>(ids_tab), nullptr, B, W, N,
                     nullptr, 0, D, static_cast<void*>(const_cast<float**>(out_tab)), K);
  return check_launch();
}

extern "C" int mi_oov_splice_rows(const int64_t* ids, const int64_t* oov_rank, int64_t B, const float* table,
                                  int64_t n_vocab, const float* oov_rows, int64_t n_oov, int64_t D, float* out,
                                  void* stream) {
  if (B < 0 || n_vocab < 0 || n_oov < 0 || D <= 0) return MI_OOV_ERR_SHAPE;
  if (B == 0) return MI_OOV_OK;
  if (!ids || !oov_rank || !out || (n_vocab > 0 && !table) || (n_oov > 0 && !oov_rows)) return MI_OOV_ERR_NULL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool vec = (D % 4 == 0) && (!table || aligned16(table)) && (!oov_rows || aligned16(oov_rows)) && aligned16(out);
  const int grid = grid_for(B, 64);
  if (vec)
    hipLaunchKernelGGL((row_copy_kernel<true, 1>), dim3(grid), dim3(kBlock), 0, st, ids, oov_rank, B, table, n_vocab, oov_rows, n_oov, D, out, 1);
  else
    hipLaunchKernelGGL((row_copy_kernel<false, 1>), dim3(grid), dim3(kBlock), 0, st, ids, oov_rank, B, table, n_vocab, oov_rows, n_oov, D, out, 1);
  return check_launch();
}

extern "C" int mi_oov_gather_mean(const int64_t* idx, int64_t M, int64_t g, const float* W, int64_t N, int64_t D,
                                  float* out, void* stream) {
  if (M < 0 || g <= 0 || N <= 0 || D <= 0) return MI_OOV_ERR_SHAPE;
  if (M == 0) return MI_OOV_OK;
  if (!idx || !W || !out) return MI_OOV_ERR_NULL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool vec = (D % 4 == 0) && aligned16(W) && aligned16(out);
  const int64_t nout = (M + g - 1) / g;
  if (g == 2) {  // the reference's group size: four outputs per 16-lane group, their 8 gathers in flight together
    const int grid = grid_for(nout, 16 * MI_GM_R);
    if (vec)
      hipLaunchKernelGGL((gather_mean_kernel<true, true>), dim3(grid), dim3(kBlock), 0, st, idx, M, g, W, N, D, out, 1);
    else
      hipLaunchKernelGGL((gather_mean_kernel<false, true>), dim3(grid), dim3(kBlock), 0, st, idx, M, g, W, N, D, out, 1);
    return check_launch();
  }
  const int grid = grid_for(nout, 16);
  if (vec)
    hipLaunchKernelGGL((gather_mean_kernel<true, false>), dim3(grid), dim3(kBlock), 0, st, idx, M, g, W, N, D, out, 1);
  else
    hipLaunchKernelGGL((gather_mean_kernel<false, false>), dim3(grid), dim3(kBlock), 0, st, idx, M, g, W, N, D, out, 1);
  return check_launch();
}

// K queued batches of mi_oov_gather_mean (every batch M indices, group size g) in one launch.
extern "C" int mi_oov_gather_mean_multi(const int64_t* const* idx_tab, float* const* out_tab, int64_t K, int64_t M, int64_t g,
                                        const float* W, int64_t N, int64_t D, void* stream) {
  if (K < 0 || M < 0 || g <= 0 || N <= 0 || D <= 0) return MI_OOV_ERR_SHAPE;
  if (K == 0 || M == 0) return MI_OOV_OK;
  if (!idx_tab || !out_tab || !W) return MI_OOV_ERR_NULL;
  if (D % 4 != 0) return MI_OOV_ERR_SHAPE;
  if (!aligned16(W) || (reinterpret_cast<uintptr_t>(idx_tab) & 7u) || (reinterpret_cast<uintptr_t>(out_tab) & 7u)) return MI_OOV_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t nout = (M + g - 1) / g;
  // 64-float rows, pairs: the software-pipelined persistent kernel of lsh64p.hip (indices two tiles ahead, rows one
  // tile ahead: 9.0 us per 65536 outputs against 10.8 for the grid-stride kernel below)
  if (g == 2 && D == 64 && M < (int64_t(1) << 24) && persist_movers()) return launch_gather_mean64_persistent(idx_tab, out_tab, K, M, W, N, st);
  if (g == 2) {
    const int grid = grid_for(K * ((nout + 15) / 16) * 16, 64);
    hipLaunchKernelGGL((gather_mean_kernel<true, true, true>), dim3(grid), dim3(kBlock), 0, st, static_cast<const void*>(idx_tab), M, g, W, N, D,
                       static_cast<void*>(const_cast<float**>(out_tab)), K);
  } else {
    const int grid = grid_for(K * ((nout + 3) / 4) * 4, 16);
    hipLaunchKernelGGL((gather_mean_kernel<true, false, true>), dim3(grid), dim3(kBlock), 0, st, static_cast<const void*>(idx_tab), M, g, W, N, D,
                       static_cast<void*>(const_cast<float**>(out_tab)), K);
  }
  return check_launch();
}

extern "C" int64_t mi_oov_col_mean_workspace(int64_t N, int64_t D) {
  if (N <= 0 || D <= 0) return 0;
  const int64_t RP = col_part_rows(N);
  return ((N + RP - 1) / RP) * D;
}

extern "C" int mi_oov_col_mean(const float* W, int64_t N, int64_t D, float* mean, float* workspace, void* stream) {
  if (N <= 0 || D <= 0) return MI_OOV_ERR_SHAPE;
  if (!W || !mean || !workspace) return MI_OOV_ERR_NULL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t RP = col_part_rows(N);
  const int64_t P = (N + RP - 1) / RP;
  const int dchunks = static_cast<int>((D + 63) / 64);
  const size_t lds = static_cast<size_t>(16) * dchunks * 64 * sizeof(float);
  if (lds > 64 * 1024) return MI_OOV_ERR_SHAPE;
  const int grid = static_cast<int>(P < kMaxGrid ? P : kMaxGrid);
  const bool vec = (D % 4 == 0) && aligned16(W);
  if (vec)
    hipLaunchKernelGGL(col_partial_kernel<true>, dim3(grid), dim3(kBlock), lds, st, W, N, D, workspace);
  else
    hipLaunchKernelGGL(col_partial_kernel<false>, dim3(grid), dim3(kBlock), lds, st, W, N, D, workspace);
  if (int rc = check_launch()) return rc;
  const int g2 = static_cast<int>((D + kBlock / 64 - 1) / (kBlock / 64));
  hipLaunchKernelGGL(col_final_kernel, dim3(g2), dim3(kBlock), 0, st, workspace, P, N, D, mean);
  return check_launch();
}

extern "C" int mi_oov_broadcast_rows(const float* vec, int64_t B, int64_t D, float* out, void* stream) {
  if (B < 0 || D <= 0) return MI_OOV_ERR_SHAPE;
  if (B == 0) return MI_OOV_OK;
  if (!out) return MI_OOV_ERR_NULL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool v = (D % 4 == 0) && aligned16(out);
  const int grid = grid_for(B, 64);
  if (v)
    hipLaunchKernelGGL(broadcast_kernel<true>, dim3(grid), dim3(kBlock), 0, st, vec, B, D, out);
  else
    hipLaunchKernelGGL(broadcast_kernel<false>, dim3(grid), dim3(kBlock), 0, st, vec, B, D, out);
  return check_launch();
}

extern "C" int mi_oov_rowdot(const float* U, const float* E, int64_t B, int64_t D, float* score, void* stream) {
  if (B < 0 || D <= 0) return MI_OOV_ERR_SHAPE;
  if (B == 0) return MI_OOV_OK;
  if (!U || !E || !score) return MI_OOV_ERR_NULL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool vec = (D % 4 == 0) && aligned16(U) && aligned16(E);
  const int grid = grid_for(B, 64);
  if (vec)
    hipLaunchKernelGGL(rowdot_kernel<true>, dim3(grid), dim3(kBlock), 0, st, U, E, B, D, score);
  else
    hipLaunchKernelGGL(rowdot_kernel<false>, dim3(grid), dim3(kBlock), 0, st, U, E, B, D, score);
  return check_launch();
}
