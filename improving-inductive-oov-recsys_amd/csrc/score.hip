// Full-catalogue scoring: BPR.full_sort_predict / ind_full_sort_predict (bpr.py:151-163) and the
// row-wise top-k that consumes it (R/evaluator/collector.py:158-167); the same pair serves as the
// exact neighbour search that stands in for ScaNN (knn_embedder.py:100-102).
//
// scores[b,n] = sum_d U[b,d]*E[n,d] is the only dense contraction of the path and runs on the
// f32 matrix cores: v_mfma_f32_32x32x2_f32 is bit-for-bit an fmaf chain in k order (no reduced
// precision), so the result equals oracle/oov_oracle.c::oov_full_sort_scores exactly.
// With K = D = 64 the GEMM is output-bound as much as MFMA-bound (32 flop per stored byte), so the
// tile is chosen for full-width coalesced stores: 128 x 128 per workgroup, 64 x 64 per wave.
#include <cstdlib>

#include "common.hpp"

namespace mi_oov {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, KC = 32, LDK = KC + 4;  // K chunk 32: 36 KB LDS -> 4 workgroups per CU (K chunk 64,
                                                          // 2 per CU, measured: full sort +12 %, dhe MLP +19 % time);
                                                          // row stride 36 floats keeps the b128 reads conflict-free

// Loads 8 consecutive k of one row (guarded), returns them split into even/odd k so that the MFMA
// lane halves (h = lane>>5 supplies k = 2s+h) read their 4 values with ONE ds_read_b128 and the
// chain still runs over increasing k.
template <bool VEC>
__device__ __forceinline__ void load8_split(const float* M, int64_t row, int64_t nrows, int64_t D, int k0,
                                            float4& ev, float4& od) {
  float v[8];
  if (row < nrows) {
    const float* src = M + row * D;
    if (VEC && k0 + 8 <= D) {
      const float4 a = *reinterpret_cast<const float4*>(src + k0);
      const float4 b = *reinterpret_cast<const float4*>(src + k0 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (k0 + i < D) ? src[k0 + i] : 0.f;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
  }
  ev = make_float4(v[0], v[2], v[4], v[6]);
  od = make_float4(v[1], v[3], v[5], v[7]);
}

// Epilogues: the same tiled product serves full-sort scoring (EPI_NONE) and the Linear layers of the
// dhe/fdhe/dnn hash nets (y = act(x W^T + b), dh_embedder.py:70-89): W is [N_out, K] row-major, exactly
// the E operand's layout.
//
// EPI_TILEMAX / EPI_FILTER are the two passes of the fused score + top-k (mi_oov_score_topk): the
// [B,N] score matrix is never written.  Pass 1 records, per row and per 64-column tile, the best
// order key of the tile; the k-th best of a row's tile maxima, tau, is a lower bound of the row's
// k-th best score (k different tiles hold a score >= tau).  Pass 2 recomputes the scores and appends
// every (key, column) with key >= tau to the row's candidate list; a last kernel rank-sorts the
// short lists.  The GEMM passes replace the 0.8 GB write plus four re-reads of the materialised path.  Pass 1 only
// visits every 8th 128-column block (the k-th best tile maximum of a SUBSET of the columns is still a lower
// bound), candidates are collected in LDS and appended in one sweep per tile into 64 interleaved per-row
// segments, tau is found by rank counting: 4096 x 50000 x 64, k = 20: 0.45 ms (two full passes + one list per
// row: 0.76 ms).  The filter compares each score with tau as a float (one v_cmp per element); keys are built only for the survivors.
enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_GELU = 2, EPI_BIAS_SIGMOID = 3, EPI_TILEMAX = 4, EPI_FILTER = 5 };

struct TopkArgs {
  uint32_t* tilemax;    // [B, NT] best order key per 64-column tile (0 = no valid column)
  int64_t NT;
  const uint32_t* tau;  // [B] lower bound of the k-th best key
  const float* tauf;    // [B] the same bound as a float threshold: a score passes iff !(score < tauf)
  int* cnt;             // [B, kSeg] candidates appended so far, per segment (see below)
  uint64_t* cand;       // [B, kSeg, kSegCap] (key << 32) | (0xFFFFFFFF - column)
  int seg_width;        // 128-column blocks per segment
  int cap;
  int64_t n_skip_low;
  int col_stride;  // TILEMAX pass: only every col_stride-th 128-column block is visited (0/1 = all)
};

// A row's candidate list is cut into kSeg segments, each fed by every kSeg-th 128-column block and with its own
// counter: the filter pass appends with one global atomic per candidate, and ~100 appends to ONE counter per row
// from hundreds of workgroups serialise in L2 (the filter pass went from 330 to 550 us when the sampled first pass
// let 160 instead of 40 candidates per row through); 64 counters per row do not.
constexpr int kSeg = 64, kSegCap = 16;  // 64 x 16 = the 1024 candidates the finalize kernel can rank
// A segment that is full spills into the row's overflow list (one more counter per row, kept behind the B x kSeg
// segment counters; its entries behind the B x kSeg x kSegCap segment slots): with ~200 candidates per row a 17th
// entry in one of 64 segments happens about once per few thousand rows, and the exact fallback for a row whose
// candidates were lost re-scores the whole catalogue in one workgroup (6.5 ms at N = 50000, 13 ms at 500 000, 0.26 s at
// 10 M).  The count of candidates of a row is heavy-tailed when k is small -- tau is the k-th best of a SAMPLE, so the
// count behaves like a Gamma(k) variable around its mean of ~1.3 k stride: with k = 2 one row in 10^5 has 7 times the
// mean -- and with 128 overflow slots about one call in twenty of the 10 M-row knn search (k = 2) hit a fallback row
// (0.39 s instead of 7 ms), as did row 1726 of the 4096 x 500 000, k = 5 sweep case (387 candidates, 13.4 ms instead
// of 0.4).  512 slots put that at e^-36 per row.
constexpr int kOvfCap = 512;
constexpr int kOvfLds = 128;  // entries of an overflow list the finalize kernel keeps in LDS (longer lists: read in place)
struct TopkArgs;
__device__ __forceinline__ void append_candidate(const TopkArgs& ta, int64_t B, int64_t row, unsigned blk, uint64_t packed);

__device__ __forceinline__ void append_candidate(const TopkArgs& ta, int64_t B, int64_t row, unsigned blk, uint64_t packed) {
  const int64_t seg = row * kSeg + blk % kSeg;  // interleaved: clustered good columns spread over the segments
  const int pos = atomicAdd(&ta.cnt[seg], 1);
  if (pos < kSegCap) {
    ta.cand[seg * kSegCap + pos] = packed;
  } else {
    const int p2 = atomicAdd(&ta.cnt[B * kSeg + row], 1);
    if (p2 < kOvfCap) ta.cand[B * kSeg * kSegCap + row * kOvfCap + p2] = packed;
  }
}

// Order: larger value first, NaN above everything (torch.topk), ties -> lower column index.
__device__ __forceinline__ uint32_t order_key(float v) {
  const uint32_t u = __float_as_uint(v);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0xFFFFFFFFu;  // NaN
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float key_to_float(uint32_t key) {
  if (key == 0xFFFFFFFFu) return qnan();
  return __uint_as_float((key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key);
}

template <int EPI>
__device__ __forceinline__ float epilogue(float v, float b) {
  if (EPI == EPI_NONE) return v;
  v = v + b;
  if (EPI == EPI_BIAS_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));  // nn.GELU(): erf form
  if (EPI == EPI_BIAS_SIGMOID) return 1.0f / (1.0f + expf(-v));
  return v;
}

#ifndef MI_FS_NT
#define MI_FS_NT 1
#endif
#ifndef MI_FS_WIDE
#define MI_FS_WIDE 1
#endif
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <bool VEC, int EPI>
__global__ __launch_bounds__(kBlock, 4) void full_sort_kernel(const float* __restrict__ U, int64_t B,
                                                           const float* __restrict__ E, int64_t N, int64_t D,
                                                           const float* __restrict__ bias,
                                                           float* __restrict__ S, int64_t ldS, TopkArgs ta) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;             // [BM][LDK]
  float* sB = smem + BM * LDK;  // [BN][LDK]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  const int i32 = lane & 31, hh = lane >> 5;
  const int64_t n0 = static_cast<int64_t>(blockIdx.x) * BN * ((EPI == EPI_TILEMAX && ta.col_stride > 1) ? ta.col_stride : 1);
  const int64_t b0 = static_cast<int64_t>(blockIdx.y) * BM;

  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // K loop.  Measured and dropped for D = 64: a "strip" kernel (user rows staged once per workgroup, the next block of
  // E prefetched into registers during the MFMAs, 2 x 34 KB LDS, 2 workgroups per CU): 358 vs 363 us for the full sort,
  // 547 vs 461 us for the fused top-k (221-256 VGPRs) -- the tile loop is not waiting for its operands; the f32 matrix
  // cores at their sustained clock are the bound.
  // A register-prefetched (software-pipelined) variant measured SLOWER on MI355X (full sort
  // 442 vs 362 us, dhe MLP 1.53 vs 1.49 ms): with 36 KB of LDS four workgroups share a CU and already
  // overlap one another's staging with MFMA work, while the extra 32 VGPRs of prefetch cost occupancy.
  for (int kc = 0; kc < D; kc += KC) {
    if (kc) __syncthreads();
    // stage: 128 rows x KC/8 units of 8 floats per operand
    constexpr int UPR = KC / 8;
#pragma unroll
    for (int j = 0; j < (BM * UPR) / kBlock; ++j) {
      const int u = tid + kBlock * j;
      const int r = u / UPR, t8 = u % UPR;
      float4 ev, od;
      load8_split<VEC>(U, b0 + r, B, D, kc + t8 * 8, ev, od);
      *reinterpret_cast<float4*>(sA + r * LDK + t8 * 8) = ev;
      *reinterpret_cast<float4*>(sA + r * LDK + t8 * 8 + 4) = od;
      load8_split<VEC>(E, n0 + r, N, D, kc + t8 * 8, ev, od);
      *reinterpret_cast<float4*>(sB + r * LDK + t8 * 8) = ev;
      *reinterpret_cast<float4*>(sB + r * LDK + t8 * 8 + 4) = od;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < KC / 8; ++t) {
      float4 a[2], b[2];
#pragma unroll
      for (int m = 0; m < 2; ++m)
        a[m] = *reinterpret_cast<const float4*>(sA + (wm * 64 + m * 32 + i32) * LDK + t * 8 + hh * 4);
#pragma unroll
      for (int n = 0; n < 2; ++n)
        b[n] = *reinterpret_cast<const float4*>(sB + (wn * 64 + n * 32 + i32) * LDK + t * 8 + hh * 4);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].x, b[n].x, acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].y, b[n].y, acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].z, b[n].z, acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].w, b[n].w, acc[m][n], 0, 0, 0);
        }
    }
  }

  // C/D map of the 32x32 shapes: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  if constexpr (EPI == EPI_TILEMAX) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = b0 + wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        uint32_t best = 0;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int64_t col = n0 + wn * 64 + n * 32 + i32;
          const uint32_t key = order_key(acc[m][n][r]);
          if (col < N && col >= ta.n_skip_low && key > best) best = key;
        }
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) {  // the 32 lanes of one half share the row
          const uint32_t o = __shfl_xor(best, off, 64);
          best = o > best ? o : best;
        }
        // compact index: the visited blocks are numbered consecutively (blockIdx.x), two 64-column tiles each;
        // a tile that lies entirely beyond N records 0 ("no valid column")
        if (i32 == 0 && row < B) ta.tilemax[row * ta.NT + blockIdx.x * 2 + wn] = (n0 + wn * 64 < N) ? best : 0u;
      }
  } else if constexpr (EPI == EPI_FILTER) {
    // Candidates (key >= tau) are first collected in LDS -- the operand tiles are dead by now -- and appended to the
    // rows' global lists in one sweep at the end: a global atomic with return stalls its wave for a full round
    // trip, and with ~0.4 candidates per row per block a wave would eat a dozen of those one after another.
    constexpr int WCAP = 1024;
    __syncthreads();  // every wave is done reading sA / sB
    int* wcnt = reinterpret_cast<int*>(smem);
    int* wrow = reinterpret_cast<int*>(smem) + 4;
    uint64_t* wbuf = reinterpret_cast<uint64_t*>(smem + 4 + WCAP);
    if (tid == 0) *wcnt = 0;
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lrow = wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const int64_t row = b0 + lrow;
        // one float compare per score instead of a key conversion + integer compare (the epilogue is a third of this
        // pass): !(s < tauf) is true for s >= tauf and for NaN scores, which rank first.  A superset of key >= tau (it
        // also lets -0 through when tau is +0): harmless, the finalize kernel ranks by key.
        const float tauf = (row < B) ? ta.tauf[row] : __builtin_inff();
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int64_t col = n0 + wn * 64 + n * 32 + i32;
          if (!(acc[m][n][r] < tauf) && row < B && col < N && col >= ta.n_skip_low) {
            const uint32_t key = order_key(acc[m][n][r]);
            const uint64_t packed = (static_cast<uint64_t>(key) << 32) | (0xFFFFFFFFu - static_cast<uint32_t>(col));
            const int p = atomicAdd(wcnt, 1);
            if (p < WCAP) {
              wrow[p] = lrow;
              wbuf[p] = packed;
            } else {  // more than WCAP candidates in one 128 x 128 tile: append directly
              append_candidate(ta, B, row, blockIdx.x, packed);
            }
          }
        }
      }
    __syncthreads();
    const int nw = *wcnt < WCAP ? *wcnt : WCAP;
    for (int i = tid; i < nw; i += kBlock) {
      append_candidate(ta, B, b0 + wrow[i], blockIdx.x, wbuf[i]);
    }
  } else {
    // Store epilogue.  A lane holds 16 ROWS of one column per accumulator, so writing the accumulators as they stand is
    // 64 dword stores per lane and tile (two 128-byte pieces per instruction).  WIDE: each wave turns its 64 x 64 sub-tile
    // through LDS -- the operand tiles are dead by now; 32 rows x 64 columns at a time in a region of its own, so no
    // workgroup barrier is needed after the first -- and stores 16 bytes per lane: 16 store instructions per lane and
    // tile, each covering four 256-byte row pieces.  Worth 2 % on the full sort (4096 x 50 000 x 64: 367 -> 360 us) and
    // 4 % on the dhe hash net (1339 -> 1281 us per 65536 lookups).  What bounds this kernel (round-3 knock-outs, same
    // shape): one store in sixteen 268 us, the stores and the staging without the matrix instructions 183 us, all of it
    // 358: the main loop runs at ~65 % of the f32 matrix rate (two K chunks per tile: a workgroup's first operand load
    // overlaps nothing of its own) and the 819 MB of stores add ~90 us on top -- HBM writes at ~4.5 TB/s need 180 us and
    // hide only half under the matrix phases of the neighbours.  Neither static wave priorities by workgroup (s_setprio
    // 0..3) nor first-round workgroups staggered over eight start times moved it (365-376 us).
    // Used when rows of S are 16-byte aligned (ldS % 4 == 0 and S aligned); otherwise the dword form below.
    const bool wide = MI_FS_WIDE && (ldS % 4 == 0) && ((reinterpret_cast<uintptr_t>(S) & 15u) == 0);
    if (wide) {
      constexpr int TLD = 68;  // floats per LDS row of the turned sub-tile: 64 + 4 (16-byte aligned rows, staggered banks)
      __syncthreads();         // every wave is done reading sA / sB
      float* tw = smem + wv * (32 * TLD);  // 4 waves x 32 x 68 floats = 34 816 B of the 36 864 the operands had
      const int l16 = lane & 15, lr4 = lane >> 4;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int64_t col = n0 + wn * 64 + n * 32 + i32;
          const float bcol = (EPI != EPI_NONE && col < N) ? bias[col] : 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r)
            tw[((r & 3) + 8 * (r >> 2) + 4 * hh) * TLD + n * 32 + i32] = epilogue<EPI>(acc[m][n][r], bcol);
        }
        __builtin_amdgcn_wave_barrier();  // (the region is this wave's own: LDS operations of one wave complete in order)
        const int64_t colq = n0 + wn * 64 + l16 * 4;
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
          const int lrow = ps * 4 + lr4;
          const int64_t row = b0 + wm * 64 + m * 32 + lrow;
          const f32x4v v = *reinterpret_cast<const f32x4v*>(tw + lrow * TLD + l16 * 4);
          if (row < B) {
            float* dst = S + row * ldS + colq;
            if (colq + 3 < N) {
              if (EPI == EPI_NONE && MI_FS_NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4v*>(dst));
              else *reinterpret_cast<f32x4v*>(dst) = v;
            } else {  // the catalogue ends inside these four columns
              if (colq + 0 < N) dst[0] = v.x;
              if (colq + 1 < N) dst[1] = v.y;
              if (colq + 2 < N) dst[2] = v.z;
            }
          }
        }
        __builtin_amdgcn_wave_barrier();  // the reads are issued before the next half overwrites the region
      }
      return;
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int64_t col = n0 + wn * 64 + n * 32 + i32;
        const float bcol = (EPI != EPI_NONE && col < N) ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = b0 + wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (row < B && col < N) {
            // the [B, N] score matrix is written once and read by a later launch, if at all, long after it has left the
            // caches (819 MB at 4096 x 50 000): non-temporal stores (MI_FS_NT=0: plain).  The layers of the hash nets
            // keep plain stores: their 134 MB outputs are the next layer's input and are served from the Infinity Cache.
            if (EPI == EPI_NONE && MI_FS_NT) __builtin_nontemporal_store(acc[m][n][r], &S[row * ldS + col]);
            else S[row * ldS + col] = epilogue<EPI>(acc[m][n][r], bcol);
          }
        }
      }
  }
}

// ---- row-wise top-k over materialised scores ----------------------------------------------------
__device__ __forceinline__ uint64_t u64_max(uint64_t a, uint64_t b) { return a > b ? a : b; }

// One workgroup per row; k selection passes, each a max-reduction of (key<<32 | ~col) over the
// candidates strictly below the previous winner.
// `mask` (optional): row r's exclusion bits at mask + r * mask_words; an excluded column is never a candidate.
__global__ __launch_bounds__(kBlock) void topk_rows_kernel(const float* __restrict__ S, int64_t rows, int64_t N,
                                                           int64_t ldS, int64_t k, int64_t n_skip_low,
                                                           float* __restrict__ vals, int64_t* __restrict__ idx,
                                                           const uint64_t* __restrict__ mask = nullptr, int64_t mask_words = 0) {
  __shared__ uint64_t red[kBlock / 64];
  __shared__ uint64_t winner;
  const int64_t row = blockIdx.x;
  if (row >= rows) return;
  const float* srow = S + row * ldS;
  const uint64_t* mrow = mask ? mask + row * mask_words : nullptr;
  uint64_t prev = ~0ULL;
  for (int64_t t = 0; t < k; ++t) {
    uint64_t best = 0;
    for (int64_t c = n_skip_low + threadIdx.x; c < N; c += kBlock) {
      if (mrow && ((mrow[c >> 6] >> (c & 63)) & 1ull)) continue;
      const uint64_t cand = (static_cast<uint64_t>(order_key(srow[c])) << 32) |
                            (0xFFFFFFFFu - static_cast<uint32_t>(c));
      if (t == 0 || cand < prev) best = u64_max(best, cand);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) best = u64_max(best, __shfl_xor(best, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint64_t b = red[0];
#pragma unroll
      for (int w = 1; w < kBlock / 64; ++w) b = u64_max(b, red[w]);
      winner = b;
      const int64_t c = 0xFFFFFFFFu - static_cast<uint32_t>(b & 0xFFFFFFFFu);
      const bool found = (b != 0);
      vals[row * k + t] = found ? srow[c] : -__builtin_inff();
      idx[row * k + t] = found ? c : -1;
    }
    __syncthreads();
    prev = winner;  // 0 when exhausted: no candidate is < 0, later passes emit (-inf, -1)
  }
}

// ---- radix-select top-k (k <= 256): 3 histogram passes + 1 collection pass instead of k passes --
// Exact: finds the 32-bit order key T of the k-th best entry of the row (11 + 11 + 10 bit digits, most
// significant first), then collects every column with key > T plus, in increasing column order, as
// many key == T columns as are still needed (ties -> lower index), and finally rank-sorts the <= k
// winners by (key desc, column asc).  One workgroup per row.  The row's keys come from a provider:
//   MatKeys  a materialised score row          DotKeys  the scores recomputed on the fly (fallback of
//   U32Keys  an array of keys (tile maxima)             the fused path when a candidate list overflows)
constexpr int kSelBins = 2048;

struct MatKeys {
  const float* p;
  __device__ __forceinline__ uint32_t operator()(int64_t c) const { return order_key(p[c]); }
};
struct U32Keys {
  const uint32_t* p;
  __device__ __forceinline__ uint32_t operator()(int64_t c) const { return p[c]; }
};
struct DotKeys {  // same fmaf chain as the MFMA kernel: increasing d from +0, d zero-padded to KC
  const float* u;
  const float* E;
  int64_t D;
  __device__ __forceinline__ uint32_t operator()(int64_t c) const {
    const float* e = E + c * D;
    float acc = 0.f;
    for (int64_t d = 0; d < D; ++d) acc = __builtin_fmaf(u[d], e[d], acc);
    if (D % KC) acc = __builtin_fmaf(0.f, 0.f, acc);  // the zero padding turns -0 into +0
    return order_key(acc);
  }
};

struct Dot64Keys {  // DotKeys for 64-float rows, 16-byte aligned: float4 loads, four in flight, then the same chain
  const float* u;   // (all 16 in flight cost the finalize kernel 135 instead of 81 registers: 13 -> 19 us on the common path)
  const float* E;
  __device__ __forceinline__ uint32_t operator()(int64_t c) const {
    const float4* e4 = reinterpret_cast<const float4*>(E + c * 64);
    float acc = 0.f;
#pragma unroll 1
    for (int g = 0; g < 4; ++g) {
      float4 ev[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) ev[q] = e4[4 * g + q];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc = __builtin_fmaf(u[16 * g + 4 * q + 0], ev[q].x, acc);
        acc = __builtin_fmaf(u[16 * g + 4 * q + 1], ev[q].y, acc);
        acc = __builtin_fmaf(u[16 * g + 4 * q + 2], ev[q].z, acc);
        acc = __builtin_fmaf(u[16 * g + 4 * q + 3], ev[q].w, acc);
      }
    }
    return order_key(acc);
  }
};

template <typename Dot>
struct MaskedKeys {  // ... with a row of exclusion bits: an excluded column gets key 0 (no real key is 0)
  Dot dot;
  const uint64_t* mrow;
  __device__ __forceinline__ uint32_t operator()(int64_t c) const {
    return ((mrow[c >> 6] >> (c & 63)) & 1ull) ? 0u : dot(c);
  }
};

// The finalize kernel's (rare) exact fallback, all four forms behind ONE type: select_topk_row keeps ~11 KB of static LDS
// per instantiation, and four of them left that kernel two workgroups per CU -- 4096 rows in two rounds of its latency chain
// instead of one (round 4: 17.2 -> see topk_finalize_exact_kernel).
struct AnyDotKeys {
  const float* u;
  const float* E;
  int64_t D;
  const uint64_t* mrow;  // or null
  __device__ __forceinline__ uint32_t operator()(int64_t c) const {
    if (mrow && ((mrow[c >> 6] >> (c & 63)) & 1ull)) return 0u;
    return D == 64 ? Dot64Keys{u, E}(c) : DotKeys{u, E, D}(c);
  }
};

__device__ __forceinline__ int block_excl_scan(int v, int* wave_tot, int& total) {
  // exclusive prefix of v over the 256 threads of the block (shuffles + LDS)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wave_tot[wv] = inc;
  __syncthreads();
  int base = 0;
  total = 0;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) {
    if (w < wv) base += wave_tot[w];
    total += wave_tot[w];
  }
  __syncthreads();
  return base + inc - v;
}

// Block-wide: key of the kk-th best of keys(c), c in [lo, N); returns it in T and, in need_eq, how many
// of the key == T entries belong to the best kk.  kk >= 1 and kk <= N - lo.
template <typename Keys>
__device__ void radix_kth(const Keys& keys, int64_t lo, int64_t N, int kk, uint32_t& T, int& need_eq) {
  __shared__ int hist[kSelBins];
  __shared__ int part[kBlock];
  __shared__ uint32_t s_prefix, s_pmask;
  __shared__ int s_krem;
  const int tid = threadIdx.x;
  if (tid == 0) { s_prefix = 0; s_pmask = 0; s_krem = kk; }
  __syncthreads();
  const int shifts[3] = {21, 10, 0};
  const int nbits[3] = {11, 11, 10};
  for (int pass = 0; pass < 3; ++pass) {
    const int shift = shifts[pass], nb = 1 << nbits[pass];
    for (int i = tid; i < kSelBins; i += kBlock) hist[i] = 0;
    __syncthreads();
    const uint32_t prefix = s_prefix, pmask = s_pmask;
    for (int64_t c = lo + tid; c < N; c += kBlock) {
      const uint32_t key = keys(c);
      if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & (nb - 1)], 1);
    }
    __syncthreads();
    const int per = nb / kBlock;  // bins per thread: 8 or 4; scanned from the top bin downwards
    int mine = 0;
    for (int j = 0; j < per; ++j) mine += hist[tid * per + j];
    part[tid] = mine;
    __syncthreads();
    if (tid == 0) {
      int krem = s_krem, above = 0, t = kBlock - 1;
      for (; t > 0; --t) {
        if (above + part[t] >= krem) break;
        above += part[t];
      }
      int b = t * per + per - 1;
      for (; b > t * per; --b) {
        if (above + hist[b] >= krem) break;
        above += hist[b];
      }
      s_krem = krem - above;
      s_prefix = prefix | (static_cast<uint32_t>(b) << shift);
      s_pmask = pmask | (static_cast<uint32_t>(nb - 1) << shift);
    }
    __syncthreads();
  }
  T = s_prefix;
  need_eq = s_krem;
  __syncthreads();
}

// rank-sort of n <= 1024 candidates ((key << 32) | ~column) held in LDS; the best kk go to the outputs
__device__ void emit_ranked(const uint64_t* cand, int n, int kk, int64_t k, float* vals_row, int64_t* idx_row) {
  for (int i = threadIdx.x; i < n; i += kBlock) {
    const uint64_t me = cand[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += (cand[j] > me) ? 1 : 0;
    if (rank < kk) {
      vals_row[rank] = key_to_float(static_cast<uint32_t>(me >> 32));
      idx_row[rank] = static_cast<int64_t>(0xFFFFFFFFu - static_cast<uint32_t>(me));
    }
  }
  for (int64_t t = kk + threadIdx.x; t < k; t += kBlock) {  // fewer than k candidate columns exist
    vals_row[t] = -__builtin_inff();
    idx_row[t] = -1;
  }
}

template <typename Keys>
__device__ void select_topk_row(const Keys& keys, int64_t N, int k, int64_t n_skip_low, float* vals_row,
                                int64_t* idx_row) {
  __shared__ int wave_tot[kBlock / 64];
  __shared__ uint64_t winners[256];
  const int tid = threadIdx.x;
  const int64_t nvalid = N > n_skip_low ? N - n_skip_low : 0;
  const int kk = static_cast<int>(nvalid < k ? nvalid : k);
  if (kk > 0) {
    uint32_t T;
    int need_eq;
    radix_kth(keys, n_skip_low, N, kk, T, need_eq);
    const int n_gt = kk - need_eq;  // every key > T column is a winner
    int base_gt = 0, base_eq = 0;
    for (int64_t c0 = n_skip_low; c0 < N; c0 += kBlock) {  // collection in increasing column order
      const int64_t c = c0 + tid;
      uint32_t key = 0;
      bool gt = false, eq = false;
      if (c < N) {
        key = keys(c);
        gt = key > T;
        eq = key == T;
      }
      int tot_gt, tot_eq;
      const int p_gt = block_excl_scan(gt ? 1 : 0, wave_tot, tot_gt);
      const int p_eq = block_excl_scan(eq ? 1 : 0, wave_tot, tot_eq);
      const uint64_t packed = (static_cast<uint64_t>(key) << 32) | (0xFFFFFFFFu - static_cast<uint32_t>(c));
      if (gt) winners[base_gt + p_gt] = packed;
      else if (eq && base_eq + p_eq < need_eq) winners[n_gt + base_eq + p_eq] = packed;
      base_gt += tot_gt;
      base_eq += tot_eq;
      if (base_gt >= n_gt && base_eq >= need_eq) break;  // uniform: all winners found
    }
    __syncthreads();
  }
  emit_ranked(winners, kk, kk, k, vals_row, idx_row);
}

__global__ __launch_bounds__(kBlock) void topk_select_kernel(const float* __restrict__ S, int64_t rows, int64_t N,
                                                             int64_t ldS, int k, int64_t n_skip_low,
                                                             float* __restrict__ vals, int64_t* __restrict__ idx) {
  const int64_t row = blockIdx.x;
  if (row >= rows) return;
  select_topk_row(MatKeys{S + row * ldS}, N, k, n_skip_low, vals + row * k, idx + row * k);
}

// ... with a row of exclusion bits (the route of last resort of mi_oov_score_topk_excl_dense): excluded columns rank below
// every real key and are blanked afterwards, as in the fused path's exact fallback
__global__ __launch_bounds__(kBlock) void topk_select_masked_kernel(const float* __restrict__ S, int64_t rows, int64_t N,
                                                                    int64_t ldS, int k, int64_t n_skip_low,
                                                                    const uint64_t* __restrict__ mask, int64_t mask_words,
                                                                    float* __restrict__ vals, int64_t* __restrict__ idx) {
  const int64_t row = blockIdx.x;
  if (row >= rows) return;
  const uint64_t* mrow = mask + row * mask_words;
  select_topk_row(MaskedKeys<MatKeys>{MatKeys{S + row * ldS}, mrow}, N, k, n_skip_low, vals + row * k, idx + row * k);
  __syncthreads();
  for (int t = threadIdx.x; t < k; t += kBlock) {
    const int64_t c = idx[row * k + t];
    if (c >= 0 && ((mrow[c >> 6] >> (c & 63)) & 1ull)) {
      idx[row * k + t] = -1;
      vals[row * k + t] = -__builtin_inff();
    }
  }
}

// fused path, between the two GEMM passes: tau[row] = k-th best of the row's tile maxima
__device__ __forceinline__ float tau_as_float(uint32_t key) {
  if (key == 0xFFFFFFFFu) return __builtin_inff();   // the k-th best is NaN: only +inf and NaN scores can matter
  if (key == 0u) return -__builtin_inff();           // no valid column seen: everything passes
  return key_to_float(key);
}

// bf16 prefilter (see bf16_tile_kernel): eps[row] = 1.05 * 2^-8 * |u_row| * max|e| (+ the subnormal term),
// thr[row] = tauf - 2 eps: tauf is the k-th best bf16 tile maximum, so the true k-th best score is >= tauf - eps, and a
// column that reaches it has a bf16 score >= tauf - 2 eps.  -inf when the bound is not finite.
constexpr int kNormGrid = 512;  // workgroups per matrix of the conversion kernel = partial maxima to reduce
struct Bf16Bound {
  const float* u2;         // [B] squared row norms of U
  const uint32_t* e2part;  // [n_e2 <= kNormGrid] per-workgroup maxima of the squared row norms of E (float bits)
  int n_e2;
  float* thr;              // [B] out
  float* eps;              // [B] out
};
// max_j |e_j|^2 as float bits: the callers pass the partial maxima they loaded (strided over the wave or the thread)
__device__ __forceinline__ void bf16_threshold(const Bf16Bound& bb, int64_t row, float tauf, uint32_t e2) {
  // + what flushing subnormal bf16 operands to zero could cost: 2^-126 * sqrt(D) * (|u| + |e|), D <= 128
  const float nu = sqrtf(bb.u2[row]), ne = sqrtf(__uint_as_float(e2));
  const float eps = 1.05f * 0x1p-8f * nu * ne + 0x1p-122f * (nu + ne) + 1e-30f;
  const float t = tauf - 2.f * eps;
  const bool ok = eps < __builtin_inff() && t == t;
  bb.thr[row] = ok ? t : -__builtin_inff();
  bb.eps[row] = ok ? eps : __builtin_inff();
}
__device__ __forceinline__ uint32_t e2_serial(const Bf16Bound& bb) {  // one thread (the workgroup-per-row kernel: rare shapes)
  uint32_t e2 = 0u;
  for (int i = 0; i < bb.n_e2; ++i) e2 = bb.e2part[i] > e2 ? bb.e2part[i] : e2;
  return e2;
}

__global__ __launch_bounds__(kBlock) void tile_kth_kernel(const uint32_t* __restrict__ tilemax, int64_t B, int64_t NT,
                                                          int k, uint32_t* __restrict__ tau, float* __restrict__ tauf,
                                                          Bf16Bound bb) {
  __shared__ uint32_t sk[1024];
  const int64_t row = blockIdx.x;
  if (row >= B) return;
  if (NT <= 1024) {  // rank by counting: the key that has exactly k-1 keys ahead of it (ties: lower index first)
    const int n = static_cast<int>(NT);
    for (int i = threadIdx.x; i < n; i += kBlock) sk[i] = tilemax[row * NT + i];
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += kBlock) {
      const uint32_t me = sk[i];
      int ahead = 0;
      for (int j = 0; j < n; ++j) ahead += (sk[j] > me || (sk[j] == me && j < i)) ? 1 : 0;
      if (ahead == k - 1) {
        tau[row] = me;
        tauf[row] = tau_as_float(me);
        if (bb.thr) bf16_threshold(bb, row, tau_as_float(me), e2_serial(bb));
      }
    }
    return;
  }
  uint32_t T;
  int need_eq;
  radix_kth(U32Keys{tilemax + row * NT}, 0, NT, k, T, need_eq);
  if (threadIdx.x == 0) {
    tau[row] = T;
    tauf[row] = tau_as_float(T);
    if (bb.thr) bf16_threshold(bb, row, tau_as_float(T), e2_serial(bb));
  }
}

// the same for NT <= 1024, one wave per row (4 rows per workgroup): the tile maxima sit in registers (up to 16 per lane)
// and the k-th best is the largest T with at least k keys >= T, found bit by bit (32 ballot steps).  The
// workgroup-per-row kernel above ranks by counting, NT^2 compares per row: 12.5 us for 4096 rows of 98 keys, nearly all
// of it launch and barrier latency, and 85 us at 262 keys.
constexpr int kKthRegs = 16;
// PAIRS: the row holds 2 NT maxima of 32-column tiles (bf16_tilemax_direct_kernel) and the key of a 64-column tile is the
// larger of a pair -- the same NT keys as the LDS-staged pass 1 leaves, read 8 bytes at a time.
template <int REGS, bool PAIRS = false>  // 64 REGS >= NT (4 for the common shapes: the unused slots of 16 doubled the kernel's time)
__global__ __launch_bounds__(kBlock) void tile_kth_wave_kernel(const uint32_t* __restrict__ tilemax, int64_t B, int NT, int k,
                                                               uint32_t* __restrict__ tau, float* __restrict__ tauf, Bf16Bound bb) {
  const int lane = threadIdx.x & 63;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= B) return;
  uint32_t key[REGS];
#pragma unroll
  for (int q = 0; q < REGS; ++q) {
    if constexpr (PAIRS) {
      const uint2 p2 = (q * 64 + lane < NT) ? reinterpret_cast<const uint2*>(tilemax + row * (2 * NT))[q * 64 + lane] : make_uint2(0u, 0u);
      key[q] = p2.x > p2.y ? p2.x : p2.y;
    } else {
      key[q] = (q * 64 + lane < NT) ? tilemax[row * NT + q * 64 + lane] : 0u;
    }
  }
  uint32_t T = 0u;  // (fewer than k keys > 0: T stays 0 = "no bound", as the rank-counting kernel returns)
  for (int bit = 31; bit >= 0; --bit) {
    const uint32_t c = T | (1u << bit);
    int have = 0;
#pragma unroll
    for (int q = 0; q < REGS; ++q)
      if (REGS <= 4 || q * 64 < NT) have += __popcll(__ballot(key[q] >= c));  // (16 registers: uniform skip of the unused ones)
    if (have >= k) T = c;
  }
  uint32_t e2 = 0u;
  if (bb.thr) {  // max |e|^2: the norm kernel's per-workgroup maxima, reduced by the wave
    for (int i = lane; i < bb.n_e2; i += 64) e2 = bb.e2part[i] > e2 ? bb.e2part[i] : e2;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const uint32_t o = __shfl_xor(e2, off, 64);
      e2 = o > e2 ? o : e2;
    }
  }
  if (lane == 0) {
    tau[row] = T;
    tauf[row] = tau_as_float(T);
    if (bb.thr) bf16_threshold(bb, row, tau_as_float(T), e2);
  }
}

// fused path, after the filter pass: rank the row's short candidate list; a list that overflowed its
// capacity (possible only with heavily duplicated scores) falls back to a radix select that recomputes
// the row's scores on the fly.
__global__ __launch_bounds__(kBlock) void topk_finalize_kernel(const float* __restrict__ U, const float* __restrict__ E,
                                                               int64_t B, int64_t N, int64_t D, int k,
                                                               int64_t n_skip_low, const int* __restrict__ cnt,
                                                               const uint64_t* __restrict__ cand,
                                                               float* __restrict__ vals, int64_t* __restrict__ idx) {
  __shared__ uint64_t lc[kSeg * kSegCap + kOvfCap];
  __shared__ int seg_off[kSeg + 2];
  __shared__ int overflow;
  const int64_t row = blockIdx.x;
  if (row >= B) return;
  if (threadIdx.x == 0) {
    int n = 0, over = 0;
    for (int g = 0; g < kSeg; ++g) {
      const int c = cnt[row * kSeg + g];
      (void)over;
      seg_off[g] = n;
      n += c < kSegCap ? c : kSegCap;
    }
    seg_off[kSeg] = n;
    const int oc = cnt[B * kSeg + row];  // the row's overflow list: only ITS overflow loses candidates
    over = (oc > kOvfCap);
    seg_off[kSeg + 1] = n + (oc < kOvfCap ? oc : kOvfCap);
    overflow = over;
  }
  __syncthreads();
  if (!overflow) {
    for (int i = threadIdx.x; i < kSeg * kSegCap; i += kBlock) {
      const int g = i / kSegCap, j = i % kSegCap;
      if (j < seg_off[g + 1] - seg_off[g]) lc[seg_off[g] + j] = cand[(row * kSeg + g) * kSegCap + j];
    }
    for (int j = threadIdx.x; j < seg_off[kSeg + 1] - seg_off[kSeg]; j += kBlock)
      lc[seg_off[kSeg] + j] = cand[B * kSeg * kSegCap + row * kOvfCap + j];
    __syncthreads();
    const int n = seg_off[kSeg + 1];
    emit_ranked(lc, n, n < k ? n : k, k, vals + row * k, idx + row * k);
  } else {
    select_topk_row(DotKeys{U + row * D, E, D}, N, k, n_skip_low, vals + row * k, idx + row * k);
  }
}

// ---- evaluation: top-k of sparse candidate lists (uni250-style sampled ranking) --------------------------
// The reference scatters the batch's sampled scores into a dense [users, items] matrix of -inf and runs
// torch.topk on it (R/inductive/evaluator.py:118-134 -> R/evaluator/collector.py:158-167).  Here the dense
// matrix never exists: segment s owns candidates [seg_ptr[s], seg_ptr[s+1]) given as (score, column); one
// workgroup ranks them (rank sort in LDS up to 1024 candidates, radix select above) and keeps the best k
// whose column lies in [col_lo, col_hi).  Missing entries (fewer than k valid candidates): (-inf, -1), where
// the reference would pick arbitrary -inf columns -- which can never be positives.
template <bool RANGE = true>  // RANGE false: every column is inside [col_lo, col_hi) -- the 8-byte columns are not read
struct SegKeys {
  const float* s;
  const int64_t* c;
  int64_t col_lo, col_hi;
  __device__ __forceinline__ uint32_t operator()(int64_t i) const {
    if (!RANGE) return order_key(s[i]);
    const int64_t col = c[i];
    return (col >= col_lo && col < col_hi) ? order_key(s[i]) : 0u;  // 0: below every real key
  }
};

__global__ __launch_bounds__(kBlock) void segment_topk_kernel(const float* __restrict__ scores,
                                                              const int64_t* __restrict__ cols,
                                                              const int64_t* __restrict__ seg_ptr, int64_t S, int k,
                                                              int64_t col_lo, int64_t col_hi, float* __restrict__ vals,
                                                              int64_t* __restrict__ idx) {
  __shared__ uint64_t lc[1024];
  const int64_t seg = blockIdx.x;
  if (seg >= S) return;
  const int64_t lo = seg_ptr[seg], n = seg_ptr[seg + 1] - lo;
  float* vrow = vals + seg * k;
  int64_t* irow = idx + seg * k;
  const SegKeys<true> keys{scores + lo, cols + lo, col_lo, col_hi};
  if (n <= 0) {
    for (int t = threadIdx.x; t < k; t += kBlock) { vrow[t] = -__builtin_inff(); irow[t] = -1; }
    return;
  }
  if (n <= 1024) {
    for (int i = threadIdx.x; i < n; i += kBlock)
      lc[i] = (static_cast<uint64_t>(keys(i)) << 32) | (0xFFFFFFFFu - static_cast<uint32_t>(i));
    __syncthreads();
    emit_ranked(lc, static_cast<int>(n), static_cast<int>(n < k ? n : k), k, vrow, irow);
  } else {
    select_topk_row(keys, n, k, 0, vrow, irow);
  }
  __syncthreads();
  for (int t = threadIdx.x; t < k; t += kBlock) {  // candidate position -> column; drop the filtered ones
    const int64_t pos = irow[t];
    if (pos < 0) continue;
    const int64_t col = cols[lo + pos];
    if (col >= col_lo && col < col_hi) {
      irow[t] = col;
    } else {
      irow[t] = -1;
      vrow[t] = -__builtin_inff();
    }
  }
}

// The same, one WAVE per segment (4 segments per workgroup, no barriers): up to 2048 keys sit in registers (32 per
// lane), the k-th best key T is the largest T with at least kk keys >= T (32 ballot steps), the winners -- every key
// > T, then the first kk - n_gt positions with key == T -- are collected in position order and rank-sorted.  Same
// outputs as segment_topk_kernel (which ranks 1506 candidates per user in 106 us for 4096 users: a radix select with
// a dozen barriers per pass; this one is bound by reading the 12 B per candidate).  Longer segments stream the keys
// from memory on every step (slow, rare).
constexpr int kSegRegs = 32;
template <bool RANGE>
__global__ __launch_bounds__(kBlock) void segment_topk_wave_kernel(const float* __restrict__ scores,
                                                                   const int64_t* __restrict__ cols,
                                                                   const int64_t* __restrict__ seg_ptr, int64_t S, int k,
                                                                   int64_t col_lo, int64_t col_hi, float* __restrict__ vals,
                                                                   int64_t* __restrict__ idx) {
  __shared__ uint64_t win_all[kBlock / 64][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t seg = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + wv;
  if (seg >= S) return;
  uint64_t* win = win_all[wv];
  const int64_t lo = seg_ptr[seg], n64 = seg_ptr[seg + 1] - lo;
  float* vrow = vals + seg * k;
  int64_t* irow = idx + seg * k;
  const SegKeys<RANGE> keys{scores + lo, cols + lo, col_lo, col_hi};
  const int n = n64 > 0x7FFFFFFF ? 0x7FFFFFFF : static_cast<int>(n64 < 0 ? 0 : n64);  // (positions are packed in 32 bits)
  const int kk = n < k ? n : k;
  const bool in_regs = n <= 64 * kSegRegs;
  uint32_t reg[kSegRegs];
  if (in_regs) {
#pragma unroll
    for (int q = 0; q < kSegRegs; ++q) reg[q] = (q * 64 + lane < n) ? keys(q * 64 + lane) : 0u;
  }
  // count of keys >= c (c > 0: absent slots never count)
  auto count_ge = [&](uint32_t c) {
    int have = 0;
    if (in_regs) {
#pragma unroll
      for (int q = 0; q < kSegRegs; ++q)
        if (q * 64 < n) have += __popcll(__ballot(reg[q] >= c));
    } else {
      for (int b = 0; b < n; b += 64) have += __popcll(__ballot(b + lane < n && keys(b + lane) >= c));
    }
    return have;
  };
  // Short cut for kk <= 64 with the keys in registers: the kk-th best of the 64 per-lane maxima is a lower bound T0 of
  // the kk-th best key (one ballot per step instead of one per register), the keys >= T0 -- kk of them at least, a few
  // dozen typically -- are compacted in position order and rank-sorted; the best kk of that order are the winners.  More
  // than 256 such keys (heavy ties): the general search below.
  bool done = false;
  if (kk > 0 && kk <= 64 && in_regs) {
    uint32_t lmax = 0u;
#pragma unroll
    for (int q = 0; q < kSegRegs; ++q) lmax = reg[q] > lmax ? reg[q] : lmax;
    uint32_t T0 = 0u;
    for (int bit = 31; bit >= 0; --bit) {
      const uint32_t c = T0 | (1u << bit);
      if (__popcll(__ballot(lmax >= c)) >= kk) T0 = c;
    }
    int m = 0;
#pragma unroll
    for (int q = 0; q < kSegRegs; ++q)
      if (q * 64 < n) m += __popcll(__ballot(q * 64 + lane < n && reg[q] >= T0));
    if (m <= 256) {
      int base = 0;
#pragma unroll
      for (int q = 0; q < kSegRegs; ++q)
        if (q * 64 < n) {
          const bool in = q * 64 + lane < n && reg[q] >= T0;
          const uint64_t mk = __ballot(in);
          if (in) win[base + __popcll(mk & ((1ull << lane) - 1ull))] = (static_cast<uint64_t>(reg[q]) << 32) | (0xFFFFFFFFu - static_cast<uint32_t>(q * 64 + lane));
          base += __popcll(mk);
        }
      __builtin_amdgcn_wave_barrier();
      for (int i = lane; i < m; i += 64) {
        const uint64_t mine = win[i];
        int rank = 0;
        for (int j = 0; j < m; ++j) rank += (win[j] > mine) ? 1 : 0;
        if (rank < kk) {
          const int64_t pos = static_cast<int64_t>(0xFFFFFFFFu - static_cast<uint32_t>(mine));
          const int64_t col = cols[lo + pos];
          const bool ok = col >= col_lo && col < col_hi;
          vrow[rank] = ok ? key_to_float(static_cast<uint32_t>(mine >> 32)) : -__builtin_inff();
          irow[rank] = ok ? col : -1;
        }
      }
      done = true;
    }
  }
  if (kk > 0 && !done) {
    uint32_t T = 0u;
    for (int bit = 31; bit >= 0; --bit) {
      const uint32_t c = T | (1u << bit);
      if (count_ge(c) >= kk) T = c;
    }
    const int n_gt = (T == 0xFFFFFFFFu) ? 0 : count_ge(T + 1u);
    const int need_eq = kk - n_gt;
    int base_gt = 0, base_eq = 0;
    auto place = [&](int i, uint32_t key) {  // all lanes call it, in position order
      const bool gt = i < n && key > T, eq = i < n && key == T;
      const uint64_t mg = __ballot(gt), me = __ballot(eq);
      const uint64_t packed = (static_cast<uint64_t>(key) << 32) | (0xFFFFFFFFu - static_cast<uint32_t>(i));
      const uint64_t below = (1ull << lane) - 1ull;
      if (gt) win[base_gt + __popcll(mg & below)] = packed;
      if (eq) {
        const int p = base_eq + __popcll(me & below);
        if (p < need_eq) win[n_gt + p] = packed;
      }
      base_gt += __popcll(mg);
      base_eq += __popcll(me);
    };
    if (in_regs) {
#pragma unroll
      for (int q = 0; q < kSegRegs; ++q)
        if (q * 64 < n) place(q * 64 + lane, reg[q]);
    } else {
      for (int b = 0; b < n; b += 64) place(b + lane, b + lane < n ? keys(b + lane) : 0u);
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < kk; i += 64) {  // rank sort of the winners; candidate position -> column, filtered ones dropped
      const uint64_t mine = win[i];
      int rank = 0;
      for (int j = 0; j < kk; ++j) rank += (win[j] > mine) ? 1 : 0;
      const int64_t pos = static_cast<int64_t>(0xFFFFFFFFu - static_cast<uint32_t>(mine));
      const int64_t col = cols[lo + pos];
      const bool ok = col >= col_lo && col < col_hi;
      vrow[rank] = ok ? key_to_float(static_cast<uint32_t>(mine >> 32)) : -__builtin_inff();
      irow[rank] = ok ? col : -1;
    }
  }
  for (int t = kk + lane; t < k; t += 64) {
    vrow[t] = -__builtin_inff();
    irow[t] = -1;
  }
}

// rec.topk of the reference collector: out[s, j] = 1 when the j-th recommended column of segment s is one of its
// positives, out[s, k] = number of positives (collector.py:161-166); positives are CSR (pos_ptr, pos_cols).
// RANGE: only positives with col_lo <= column < col_hi count (the old-item / new-item slices of the filtered collectors).
template <bool RANGE>
__global__ __launch_bounds__(kBlock) void topk_hits_kernel(const int64_t* __restrict__ idx, int64_t S, int k,
                                                           const int64_t* __restrict__ pos_ptr,
                                                           const int64_t* __restrict__ pos_cols, int64_t col_lo, int64_t col_hi,
                                                           int* __restrict__ out) {
  const int64_t total = S * (k + 1);
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t s = i / (k + 1);
    const int j = static_cast<int>(i % (k + 1));
    const int64_t p0 = pos_ptr[s], p1 = pos_ptr[s + 1];
    if (j == k) {
      if (RANGE) {
        int n = 0;
        for (int64_t q = p0; q < p1; ++q) n += (pos_cols[q] >= col_lo && pos_cols[q] < col_hi) ? 1 : 0;
        out[i] = n;
      } else {
        out[i] = static_cast<int>(p1 - p0);
      }
    } else {
      const int64_t c = idx[s * k + j];
      int hit = 0;
      for (int64_t q = p0; q < p1; ++q) hit |= (pos_cols[q] == c) ? 1 : 0;
      out[i] = (c >= 0 && (!RANGE || (c >= col_lo && c < col_hi))) ? hit : 0;
    }
  }
}

extern "C" int mi_oov_segment_topk(const float* scores, const int64_t* cols, const int64_t* seg_ptr, int64_t S,
                                   int64_t k, int64_t col_lo, int64_t col_hi, float* vals, int64_t* idx, void* stream) {
  if (S < 0 || k <= 0 || k > 256) return MI_OOV_ERR_SHAPE;
  if (S == 0) return MI_OOV_OK;
  if (!scores || !cols || !seg_ptr || !vals || !idx) return MI_OOV_ERR_NULL;
  static const bool wg_kernel = env_knob("MI_OOV_SEGMENT_TOPK_WG", 0, 0, 1) != 0;
  if (wg_kernel)
    hipLaunchKernelGGL(segment_topk_kernel, dim3(static_cast<unsigned>(S)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       scores, cols, seg_ptr, S, static_cast<int>(k), col_lo, col_hi, vals, idx);
  else if (col_lo <= 0 && col_hi >= (1LL << 62))  // the whole column range (columns are ids: >= 0): 4 instead of 12 bytes per candidate
    hipLaunchKernelGGL(segment_topk_wave_kernel<false>, dim3(static_cast<unsigned>((S + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), scores, cols, seg_ptr, S, static_cast<int>(k), col_lo, col_hi, vals, idx);
  else
    hipLaunchKernelGGL(segment_topk_wave_kernel<true>, dim3(static_cast<unsigned>((S + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), scores, cols, seg_ptr, S, static_cast<int>(k), col_lo, col_hi, vals, idx);
  return check_launch();
}

extern "C" int mi_oov_topk_hits_range(const int64_t* idx, int64_t S, int64_t k, const int64_t* pos_ptr, const int64_t* pos_cols,
                                      int64_t col_lo, int64_t col_hi, int32_t* out, void* stream) {
  if (S < 0 || k <= 0) return MI_OOV_ERR_SHAPE;
  if (S == 0) return MI_OOV_OK;
  if (!idx || !pos_ptr || !pos_cols || !out) return MI_OOV_ERR_NULL;
  if (col_lo <= 0 && col_hi >= (1LL << 62))
    hipLaunchKernelGGL(topk_hits_kernel<false>, dim3(grid_for(S * (k + 1), kBlock)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), idx,
                       S, static_cast<int>(k), pos_ptr, pos_cols, col_lo, col_hi, out);
  else
    hipLaunchKernelGGL(topk_hits_kernel<true>, dim3(grid_for(S * (k + 1), kBlock)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), idx,
                       S, static_cast<int>(k), pos_ptr, pos_cols, col_lo, col_hi, out);
  return check_launch();
}

extern "C" int mi_oov_topk_hits(const int64_t* idx, int64_t S, int64_t k, const int64_t* pos_ptr, const int64_t* pos_cols,
                                int32_t* out, void* stream) {
  return mi_oov_topk_hits_range(idx, S, k, pos_ptr, pos_cols, 0, 1LL << 62, out, stream);
}

// ---- bf16 prefilter: both GEMM passes of the fused top-k (D = 64) ---------------------------------------------------------
// The two passes only have to decide which (row, column) pairs MAY belong to the top-k: the exact f32 score of the few
// hundred survivors per row is recomputed by the finalize kernel anyway.  So they run on the bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16: 16 x the f32 MFMA rate; bf16 copies of U and E are made once per call) with a threshold
// lowered by a bound on the bf16 error:
//     |s_bf16 - s_f32| <= (2^-8 + 2^-17 + 2 D 2^-24) * sum_d |u_d e_d|  <=  1.02 * 2^-8 * |u| * |e|      (Cauchy-Schwarz)
// (each operand rounds to 8 significant bits: relative 2^-9; f32 accumulation on both sides).  With eps_row =
// 1.05 * 2^-8 * |u_row| * max_j |e_j| every column whose exact score reaches tau has a bf16 score >= tau - eps_row, so
// the candidate set is a superset of the exact one and the final top-k is bit-identical to the all-f32 path's (the
// finalize kernel recomputes every candidate's score with the oracle's fmaf chain and ranks exactly; the bound also
// carries a term for subnormal operands that the matrix cores may flush).  Non-finite norms
// give eps = inf/NaN -> threshold -inf: everything passes, the row overflows and takes the exact fallback.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
constexpr int BLD = 72;  // bf16 elements per LDS row: 64 + 8 (144 B: staggers the banks, keeps 16-B alignment)

// The bf16 copy of E is kept in FRAGMENT ORDER: rows in tiles of 32, a tile = [8 chunks of 8 k][32 rows][8 bf16] = 4 KB,
// padded with zero rows to a multiple of 128 rows.  The matrix cores want, per k-step, 16 bytes of each of 32 rows: in
// row-major order that is a load instruction over 32 cache lines using a quarter of each, four times over for the four
// steps of a row (the second pass spent 41 % of its waves' cycles in s_waitcnt that a deeper prefetch did not touch: the
// texture path's line rate, not latency); in this order the 64 lanes of a step read one contiguous KB.  The copy is
// private to the library (workspace / prepared catalogue), so only its writer and its three readers know.
__host__ __device__ constexpr int64_t eb_rows(int64_t N) { return (N + 127) / 128 * 128; }
// KH = number of 64-k halves of a row: 1 for D <= 64, 2 for 64 < D <= 128 (round 3: the same kernels with twice the
// k-steps; a tile is then [16 chunks of 8 k][32 rows][8 bf16] = 8 KB and a row of the U copy 128 bf16).
template <int KH = 1>
__device__ __forceinline__ int64_t eb_chunk(int64_t row, int chunk) {  // offset (bf16 elements) of 8 k of a row; chunk < 8 KH
  return (row >> 5) * (2048 * KH) + (static_cast<int64_t>(chunk) * 32 + (row & 31)) * 8;
}

// bf16 copies of U and E (64-float rows) + squared row norms, one launch: workgroups [0, gu) take U (norms to u2[], the
// rows' overflow counters zeroed), workgroups [gu, gu + ge) take E and leave the maximum squared norm of their rows in
// e2part[workgroup] (float bits; unsigned order = float order for non-negative floats, NaN images sort above +inf; the
// tau kernel reduces the <= 512 partials -- an atomicMax per row on one word took 140 us, one per workgroup 2.5 us, and
// the word needed zeroing by an earlier launch).  16 lanes per row, 4 rows per thread and round (independent loads).
// Separate launches for U and E were 5.2 + 9.3 us: a launch that moves 1 MB costs 5 us all the same.
template <int KH>
__global__ __launch_bounds__(kBlock) void to_bf16_norm_kernel(const float* __restrict__ U, int64_t B, __bf16* __restrict__ Ub,
                                                              float* __restrict__ u2, int* __restrict__ zero_rows, int gu,
                                                              const float* __restrict__ E, int64_t N, __bf16* __restrict__ Eb,
                                                              uint32_t* __restrict__ e2part, int D) {
  // D <= 64 KH floats per row (any alignment unless D is 64 or 128): the bf16 copies are zero-padded to 64 KH k
  __shared__ uint32_t bmax;
  const bool is_u = static_cast<int>(blockIdx.x) < gu;
  const float* M = is_u ? U : E;
  __bf16* Mb = is_u ? Ub : Eb;
  const int64_t rows = is_u ? B : N;
  const int64_t rows_out = is_u ? B : eb_rows(N);  // (E: zero rows up to the next multiple of 128)
  const int blk = is_u ? blockIdx.x : blockIdx.x - gu, nblk = is_u ? gu : gridDim.x - gu;
  if (threadIdx.x == 0) bmax = 0u;
  __syncthreads();
  const int l16 = threadIdx.x & 15;
  const int64_t per = kBlock / 16;
  const int64_t step = static_cast<int64_t>(nblk) * per;
  uint32_t mine = 0u;
  for (int64_t r0 = static_cast<int64_t>(blk) * per + (threadIdx.x >> 4); r0 < rows_out; r0 += 4 * step) {
    float4 v[KH][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t r = r0 + q * step;
#pragma unroll
      for (int h = 0; h < KH; ++h) {
        const int e = h * 64 + l16 * 4;  // this lane's four floats of the half
        v[h][q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < rows) {
          if (D == 64 * KH) {
            v[h][q] = *reinterpret_cast<const float4*>(M + r * (64 * KH) + e);
          } else {
            const float* src = M + r * D + e;
            if (e + 0 < D) v[h][q].x = src[0];
            if (e + 1 < D) v[h][q].y = src[1];
            if (e + 2 < D) v[h][q].z = src[2];
            if (e + 3 < D) v[h][q].w = src[3];
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t r = r0 + q * step;
      float s = 0.f;
#pragma unroll
      for (int h = 0; h < KH; ++h) {
        s = __builtin_fmaf(v[h][q].x, v[h][q].x, s);
        s = __builtin_fmaf(v[h][q].y, v[h][q].y, s);
        s = __builtin_fmaf(v[h][q].z, v[h][q].z, s);
        s = __builtin_fmaf(v[h][q].w, v[h][q].w, s);
      }
      s = row16_sum(s);  // (every lane takes part: no divergence around the cross-lane sum)
      if (r < rows_out) {
#pragma unroll
        for (int h = 0; h < KH; ++h) {
          uint2 pk;  // (rows past N: v = 0)
          pk.x = __builtin_bit_cast(uint32_t, bf16x2{static_cast<__bf16>(v[h][q].x), static_cast<__bf16>(v[h][q].y)});
          pk.y = __builtin_bit_cast(uint32_t, bf16x2{static_cast<__bf16>(v[h][q].z), static_cast<__bf16>(v[h][q].w)});
          *reinterpret_cast<uint2*>(Mb + (is_u ? r * (64 * KH) + h * 64 + l16 * 4
                                               : eb_chunk<KH>(r, h * 8 + (l16 >> 1)) + (l16 & 1) * 4)) = pk;
        }
      }
      if (r < rows) {
        if (l16 == 0) {
          if (is_u) {
            u2[r] = s;
            zero_rows[r] = 0;
          }
          const uint32_t bits = __float_as_uint(s);
          mine = bits > mine ? bits : mine;
        }
      }
    }
  }
  if (!is_u) {  // uniform per workgroup
    if (l16 == 0 && mine) atomicMax(&bmax, mine);
    __syncthreads();
    if (threadIdx.x == 0) e2part[blk] = bmax;
  }
}

// exclusion bitmap of a user batch from the CSR of excluded columns (histories): one workgroup per row; the words were
// zeroed by a memset on the same stream.  Columns outside [0, N) are ignored, as the oracle does.
__global__ __launch_bounds__(kBlock) void mask_build_kernel(const int64_t* __restrict__ excl_ptr, const int64_t* __restrict__ excl_cols,
                                                            int64_t B, int64_t N, unsigned long long* __restrict__ mask, int64_t words) {
  const int64_t row = blockIdx.x;
  if (row >= B) return;
  for (int64_t q = excl_ptr[row] + threadIdx.x; q < excl_ptr[row + 1]; q += kBlock) {
    const int64_t c = excl_cols[q];
    if (c >= 0 && c < N) atomicOr(&mask[row * words + (c >> 6)], 1ull << (c & 63));
  }
}

// Candidate lists of the bf16 filter pass.  Strip s of a 128-row block is walked by exactly one workgroup, so the list
// (row, s) has ONE writer: its counter lives in that workgroup's LDS and the entries go out as plain stores -- the
// filter pass does no global atomics (one per candidate, ~0.85 M per launch, cost it 50 us).  Lists that run over
// (clustered good columns, runs of equal scores) spill into the row's shared overflow list, and a row whose overflow
// list runs over as well takes the exact fallback in the finalize kernel.
struct StripLists {
  int* cnt;        // [B, ns] entries of list (row, strip), written once by the owner
  uint64_t* cand;  // [B, ns, cap] (bf16 key << 32) | (0xFFFFFFFF - column)
  int* ovf_cnt;    // [B] zeroed per launch
  uint64_t* ovf;   // [B, kOvfCap]
  int ns, cap;
  const uint64_t* mask;  // [B, mask_words] bit c of a row = column c is excluded (the user's history); null: no exclusions
  int64_t mask_words;
};

// A workgroup keeps its 128 user rows in LDS and walks a strip of column blocks (blocks s, s + ns, s + 2 ns, ...: a
// cluster of good columns is spread over the strips, and with ns a multiple of 8 strip s always runs on XCD s % 8, so
// each XCD's L2 sees an eighth of E).  With the bf16 matrix cores a 128 x 128 x 64 tile is 0.25 us of MFMA work, so
// everything else has to stay off the critical path:
//   * the next block of E is fetched into registers while the current one is multiplied and filtered (measured by
//     knocking phases out of the FILTER kernel, 56 us: no re-staging and no barriers 54 us, no MFMAs 41 us, no
//     epilogue 31 us -- the per-score compare-and-branch, ~3 instructions x 64 scores per lane, is what is left).
//     512-thread workgroups (8 waves of 64 x 32: half the accumulators, 74 registers, 6 waves per SIMD) were slower:
//     110 vs 101 us at 50 000 items, 7.4 vs 6.3 ms at 10 M -- more waves do not help, operand reads per MFMA do (1.5 vs 1);
//   * rows past B / N are clamped to the last row (no zero fill, no divergent loads): the epilogues mask them;
//   * FILTER seeds the accumulators with -thr[row] (read from LDS with the operands), so the epilogue is one compare
//     against zero and a branch per score: a threshold read per row inside the epilogue was an exposed LDS round trip
//     each, 32 per tile, and dominated the kernel.  The keys are therefore scores shifted by the row's threshold: the
//     finalize kernel only compares keys of one row with each other.  The shift's rounding (2^-22 relative) sits well
//     inside the 5% slack of eps.
constexpr int kWaveQueue = 128;  // queued candidates per wave and tile (6 B each: 36864 + 1024 + 3072 B of LDS = 4 workgroups per CU)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// (pass 1 runs ~3 workgroups per CU anyway -- 800 working workgroups -- and at 128 registers it spills: the reload's
// s_waitcnt vmcnt(0) then also waits for the block prefetch)
// F32A (round 4; pass 1 against a PREPARED catalogue, rows of exactly 64 KH aligned floats): the user rows are staged straight
// from the caller's f32 matrix -- converted while they go to LDS, the rounding of to_bf16_norm_kernel -- and the workgroups
// of strip 0 leave what the later kernels read: the bf16 copy (filter pass), the squared norms (tau kernel) and the zeroed
// overflow counters.  The call is then four launches instead of five: with the catalogue prepared the conversion launch
// had 1 MB to convert and cost its 5.4 us of kernel boundary all the same.
struct FoldU {
  const float* U;   // [B, 64 KH] f32 user rows (null: the bf16 copy Ub is read, as before)
  __bf16* Ub_out;   // [B, 64 KH]
  float* u2;        // [B]
  int* zero_rows;   // [B] overflow counters
};
template <int EPI, bool MASKED, int KH = 1, bool F32A = false>
__global__ __launch_bounds__(kBlock, KH == 2 ? 2 : (EPI == EPI_TILEMAX ? 3 : 4)) void bf16_tile_kernel(const __bf16* __restrict__ Ub, int64_t B,
                                                             const __bf16* __restrict__ Eb, int64_t N,
                                                             const float* __restrict__ thr, TopkArgs ta, StripLists sl,
                                                             int nvisit, FoldU fold = FoldU{}) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16* sA = reinterpret_cast<__bf16*>(smem);  // [KH][BM][BLD]
  __bf16* sB = sA + KH * BM * BLD;               // [KH][BN][BLD]
  float* snthr = reinterpret_cast<float*>(sB + KH * BN * BLD);  // FILTER only: -thr[row], then the rows' list counters
  int* rowcnt = reinterpret_cast<int*>(snthr + BM);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  const int i32 = lane & 31, hh = lane >> 5;
  const int stride = (EPI == EPI_TILEMAX && ta.col_stride > 1) ? ta.col_stride : 1;
  const int64_t b0 = static_cast<int64_t>(blockIdx.y) * BM;
  const int strip = blockIdx.x, nstrip = gridDim.x;
  if (strip >= nvisit) {  // idle strip (the count is padded to a multiple of 8): its lists are empty
    if (EPI == EPI_FILTER && tid < BM && b0 + tid < B) sl.cnt[(b0 + tid) * sl.ns + strip] = 0;
    return;
  }
  const int rows_here = (B - b0 < BM) ? static_cast<int>(B - b0) : BM;
  const uint32_t n_cols = static_cast<uint32_t>(N);  // N < 2^32 (checked by the caller)
  const uint32_t skip = ta.n_skip_low < N ? static_cast<uint32_t>(ta.n_skip_low) : n_cols;
  const int erow = tid >> 3, eoff = (tid & 7) * 8;  // operand staging: 128 rows x 8 units of 8 k (16 B), 4 units per thread
  const int brow = tid & 31, bchunk = tid >> 5;     // ... of E (fragment order, eb_chunk): thread = (row of a 32-row tile, chunk)
  u32x4 pf[KH][4];  // the next block of E (a native vector type: an array of HIP's uint4 struct lands in scratch)

  if constexpr (F32A) {
    // thread (erow, eoff): eight consecutive floats of a row per half, rows erow, erow + 32, ...; the eight threads of a row
    // (a DPP row's lanes 8 i .. 8 i + 7) add their squares up for the norm
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = erow + 32 * q;
      const int64_t ra = (r < rows_here) ? b0 + r : B - 1;
      float sq = 0.f;
#pragma unroll
      for (int h = 0; h < KH; ++h) {
        const float4* src = reinterpret_cast<const float4*>(fold.U + ra * (64 * KH) + h * 64 + eoff);
        const float4 x0 = src[0], x1 = src[1];
        u32x4 pk;
        pk.x = __builtin_bit_cast(uint32_t, bf16x2{static_cast<__bf16>(x0.x), static_cast<__bf16>(x0.y)});
        pk.y = __builtin_bit_cast(uint32_t, bf16x2{static_cast<__bf16>(x0.z), static_cast<__bf16>(x0.w)});
        pk.z = __builtin_bit_cast(uint32_t, bf16x2{static_cast<__bf16>(x1.x), static_cast<__bf16>(x1.y)});
        pk.w = __builtin_bit_cast(uint32_t, bf16x2{static_cast<__bf16>(x1.z), static_cast<__bf16>(x1.w)});
        *reinterpret_cast<u32x4*>(sA + (h * BM + r) * BLD + eoff) = pk;
        if (strip == 0 && r < rows_here) *reinterpret_cast<u32x4*>(fold.Ub_out + ra * (64 * KH) + h * 64 + eoff) = pk;
        sq = __builtin_fmaf(x0.x, x0.x, sq); sq = __builtin_fmaf(x0.y, x0.y, sq);
        sq = __builtin_fmaf(x0.z, x0.z, sq); sq = __builtin_fmaf(x0.w, x0.w, sq);
        sq = __builtin_fmaf(x1.x, x1.x, sq); sq = __builtin_fmaf(x1.y, x1.y, sq);
        sq = __builtin_fmaf(x1.z, x1.z, sq); sq = __builtin_fmaf(x1.w, x1.w, sq);
      }
      sq = sq + __shfl_xor(sq, 1, 64);
      sq = sq + __shfl_xor(sq, 2, 64);
      sq = sq + __shfl_xor(sq, 4, 64);
      if (strip == 0 && r < rows_here && (tid & 7) == 0) {
        fold.u2[ra] = sq;
        fold.zero_rows[ra] = 0;
      }
    }
  }
#pragma unroll
  for (int h = 0; h < KH; ++h)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = erow + 32 * q;
      const int64_t ra = (r < rows_here) ? b0 + r : B - 1;
      const int64_t n0 = static_cast<int64_t>(strip) * stride * BN;
      if constexpr (!F32A)
        *reinterpret_cast<u32x4*>(sA + (h * BM + r) * BLD + eoff) = *reinterpret_cast<const u32x4*>(Ub + ra * (64 * KH) + h * 64 + eoff);
      *reinterpret_cast<u32x4*>(sB + (h * BN + brow + 32 * q) * BLD + bchunk * 8) =
          *reinterpret_cast<const u32x4*>(Eb + eb_chunk<KH>(n0 + brow + 32 * q, h * 8 + bchunk));
    }
  if constexpr (EPI == EPI_FILTER) {
    if (tid < BM) {
      snthr[tid] = (tid < rows_here) ? 0.f - thr[b0 + tid] : -__builtin_inff();  // (0 - t: never -0)
      rowcnt[tid] = 0;
    }
  }
  __syncthreads();

  for (int j = strip; j < nvisit; j += nstrip) {
    const int64_t n0 = static_cast<int64_t>(j) * stride * BN;
    const bool more = j + nstrip < nvisit;
    {
      const int64_t n1 = static_cast<int64_t>(more ? j + nstrip : j) * stride * BN;  // (the last block re-reads itself)
#pragma unroll
      for (int h = 0; h < KH; ++h)
#pragma unroll
        for (int q = 0; q < 4; ++q) pf[h][q] = *reinterpret_cast<const u32x4*>(Eb + eb_chunk<KH>(n1 + brow + 32 * q, h * 8 + bchunk));
    }
    // the epilogue's row / column offsets are loop invariant: made opaque here so that the compiler recomputes them per
    // block instead of carrying 64+ registers of addresses (i.e. spills) across the strip loop
    int lrow0 = wm * 64 + 4 * hh, lcol0 = wn * 64 + i32;
    asm volatile("" : "+v"(lrow0), "+v"(lcol0));
    f32x16 acc[2][2];

    uint64_t mw[2] = {0ull, 0ull};  // MASKED pass 1: the exclusion bits of this lane's two user rows for its 64-column tile
    if constexpr (EPI == EPI_TILEMAX && MASKED) {
      const int64_t tw = (n0 >> 6) + wn;  // the tile's word
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int lr = wm * 64 + m * 32 + i32;
        if (lr < rows_here && tw < sl.mask_words) mw[m] = sl.mask[(b0 + lr) * sl.mask_words + tw];
      }
    }
    if constexpr (EPI == EPI_TILEMAX) {
      // pass 1: maxima of the 64-column tiles.  The product is taken transposed (E block x U^T): a lane then holds 16
      // COLUMNS of one user row per accumulator, so the tile maximum is 32 in-lane max3 operations + one exchange with
      // lane ^ 32, instead of a 32-lane reduction per row (5 DPP steps for each of 32 rows: 1000 VALU operations per
      // wave and tile, 4x the matrix-core time).  NaN sorts first (torch.topk): max3 drops NaNs, so they are tracked
      // through the largest |bits| seen.
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 4 * KH; ++kk) {  // lane (r, h) supplies k = 16 kk + 8 h + j
        const int kh = kk >> 2, ks = kk & 3;
        bf16x8 a[2], b[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
          a[m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(sA + (kh * BM + wm * 64 + m * 32 + i32) * BLD + ks * 16 + hh * 8));
#pragma unroll
        for (int n = 0; n < 2; ++n)
          b[n] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(sB + (kh * BN + wn * 64 + n * 32 + i32) * BLD + ks * 16 + hh * 8));
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[n], a[m], acc[m][n], 0, 0, 0);
      }
      // acc[m][n][r]: column n0 + wn 64 + n 32 + (r & 3) + 8 (r >> 2) + 4 hh, user row wm 64 + m 32 + i32
      const int64_t c_lo = n0 + wn * 64, c_hi = c_lo + 64;
      const bool all_cols = c_hi <= N && c_lo >= ta.n_skip_low;  // uniform; false only at the two ends of the catalogue
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float best = -__builtin_inff();
        uint32_t amax = 0u;
        bool any = false;
        if (all_cols) {
          const uint64_t wsh = mw[m] >> (4 * hh);  // bit n 32 + (r & 3) + 8 (r >> 2) = this lane's column (n, r)
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            const uint32_t wn32 = static_cast<uint32_t>(wsh >> (32 * n));
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              float v0 = acc[m][n][r], v1 = acc[m][n][r + 1];
              if constexpr (MASKED) {  // an excluded column counts as -inf: sign-extended bit, bit-field insert
                const int b0_ = (r & 3) + 8 * (r >> 2), b1_ = ((r + 1) & 3) + 8 * ((r + 1) >> 2);
                const uint32_t e0 = static_cast<uint32_t>(static_cast<int>(wn32 << (31 - b0_)) >> 31);
                const uint32_t e1 = static_cast<uint32_t>(static_cast<int>(wn32 << (31 - b1_)) >> 31);
                v0 = __uint_as_float((__float_as_uint(v0) & ~e0) | (0xFF800000u & e0));
                v1 = __uint_as_float((__float_as_uint(v1) & ~e1) | (0xFF800000u & e1));
              }
              best = __builtin_fmaxf(__builtin_fmaxf(best, v0), v1);
              const uint32_t a0 = __float_as_uint(v0) & 0x7FFFFFFFu, a1 = __float_as_uint(v1) & 0x7FFFFFFFu;
              amax = a0 > amax ? a0 : amax;
              amax = a1 > amax ? a1 : amax;
            }
          }
          any = true;
        } else {
          uint32_t cbase = static_cast<uint32_t>(c_lo) + 4 * hh;
          asm volatile("" : "+v"(cbase));  // (not hoisted out of the rare branch)
#pragma unroll
          for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const uint32_t col = cbase + n * 32 + (r & 3) + 8 * (r >> 2);
              const bool excluded = MASKED && ((mw[m] >> (col & 63u)) & 1ull);
              if (col < n_cols && col >= skip && !excluded) {
                best = __builtin_fmaxf(best, acc[m][n][r]);
                const uint32_t a0 = __float_as_uint(acc[m][n][r]) & 0x7FFFFFFFu;
                amax = a0 > amax ? a0 : amax;
                any = true;
              }
            }
        }
        uint32_t key = !any ? 0u : (amax > 0x7F800000u ? 0xFFFFFFFFu : order_key(best));
        const uint32_t other = __shfl_xor(key, 32, 64);
        key = other > key ? other : key;
        int lrow = wm * 64 + m * 32 + i32;
        asm volatile("" : "+v"(lrow));  // (address recomputed per block: carried across the loop it spills)
        if (hh == 0 && lrow < rows_here) (ta.tilemax + b0 * ta.NT + j * 2 + wn)[lrow * static_cast<int>(ta.NT)] = key;
      }
    } else {
      // pass 2: every score that is not below the row's threshold -> the (row, strip) list
      f32x16 seed[2];  // -thr of the 16 rows a lane holds per 32 x 32 tile: the C operand of the first MFMA
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // accumulator r holds row (r & 3) + 8 (r >> 2) + 4 hh of the 32 x 32 tile
          const f32x4 t4 = *reinterpret_cast<const f32x4*>(snthr + lrow0 + m * 32 + 8 * g);
          seed[m][4 * g + 0] = t4.x;
          seed[m][4 * g + 1] = t4.y;
          seed[m][4 * g + 2] = t4.z;
          seed[m][4 * g + 3] = t4.w;
        }
#pragma unroll
      for (int kk = 0; kk < 4 * KH; ++kk) {
        const int kh = kk >> 2, ks = kk & 3;
        bf16x8 a[2], b[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
          a[m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(sA + (kh * BM + wm * 64 + m * 32 + i32) * BLD + ks * 16 + hh * 8));
#pragma unroll
        for (int n = 0; n < 2; ++n)
          b[n] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(sB + (kh * BN + wn * 64 + n * 32 + i32) * BLD + ks * 16 + hh * 8));
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[n], kk == 0 ? seed[m] : acc[m][n], 0, 0, 0);
      }
      // A passing score costs one queue push here (no waiting): the wave's queue is drained once per tile, one LDS
      // atomic on the row's list counter and one store per entry, all entries in parallel.  (Taking the list slot inside
      // the branch was an LDS round trip per taken branch, ~14 per wave and tile, and doubled the kernel's time.)
      uint64_t* cand_wg = sl.cand + (b0 * sl.ns + strip) * sl.cap;  // + local row * (ns cap) + slot: 32-bit offsets
      const int row_stride = sl.ns * sl.cap;
      auto emit = [&](int lrow, uint32_t lcol, float v) {
        const uint32_t col = static_cast<uint32_t>(n0) + lcol;
        bool allowed = lrow < rows_here && col < n_cols && col >= skip;
        if constexpr (MASKED) {
          if (allowed) allowed = !((sl.mask[(b0 + lrow) * sl.mask_words + (col >> 6)] >> (col & 63u)) & 1ull);
        }
        if (allowed) {
          const uint64_t packed = (static_cast<uint64_t>(order_key(v)) << 32) | (0xFFFFFFFFu - col);
          const int p = atomicAdd(&rowcnt[lrow], 1);
          if (p < sl.cap) {
            cand_wg[lrow * row_stride + p] = packed;
          } else if (__atomic_load_n(&sl.ovf_cnt[b0 + lrow], __ATOMIC_RELAXED) <= kOvfCap) {
            // (a row that has already run over is lost to the exact fallback: no point in counting on -- a row of equal
            // scores would otherwise do N atomics on one word; a stale low read only costs an extra atomic)
            const int p2 = atomicAdd(&sl.ovf_cnt[b0 + lrow], 1);
            if (p2 < kOvfCap) sl.ovf[(b0 + lrow) * kOvfCap + p2] = packed;
          }
        }
      };
      uint32_t* qv = reinterpret_cast<uint32_t*>(rowcnt + BM) + wv * kWaveQueue;                        // score bits
      uint16_t* qc = reinterpret_cast<uint16_t*>(reinterpret_cast<uint32_t*>(rowcnt + BM) + 4 * kWaveQueue) + wv * kWaveQueue;  // row << 8 | col
      int code0 = (lrow0 << 8) | lcol0;
      asm volatile("" : "+v"(code0));
      int qpos = 0;  // wave-uniform
      auto test_one = [&](int v) {  // v = m 32 + r 2 + n, a compile-time constant after unrolling
        const int m = v >> 5, r = (v >> 1) & 15, n = v & 1;
        const bool pass = !(acc[m][n][r] < 0.f);  // (a NaN passes too: emit() checks the row and the column)
        const uint64_t mask = __ballot(pass);
        if (mask) {
          if (pass) {
            const int slot = qpos + static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                                                         __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u)));
            const int code = code0 + (((m * 32 + (r & 3) + 8 * (r >> 2)) << 8) | (n * 32));
            if (slot < kWaveQueue) {
              qv[slot] = __float_as_uint(acc[m][n][r]);
              qc[slot] = static_cast<uint16_t>(code);
            } else {  // queue full (runs of equal scores): take the slot here
              emit(code >> 8, static_cast<uint32_t>(code & 255), acc[m][n][r]);
            }
          }
          qpos = __builtin_amdgcn_readfirstlane(qpos + __popcll(mask));
        }
      };
      // Three scores per wave-level branch: as signed integers the float images that FAIL (x < 0) are exactly those
      // <= 0xFF800000 (-inf) -- negative numbers; NaNs of either sign and everything >= +0 lie above; -0 cannot occur,
      // the seed is never -0 and a sum that cancels rounds to +0 -- so "some score passes" is one v_max3_i32 and one
      // compare.  (About one wave-level test in eight finds a passing score; per-score branches were 3 instructions
      // each on the path where none does.  A coarser first test -- one per 32 scores -- for catalogues so big that
      // hardly any tile holds a candidate changed nothing: 10 M rows, k = 2: 6.42 vs 6.40 ms.)
      auto bits = [&](int v) { return static_cast<int>(__float_as_uint(acc[v >> 5][v & 1][(v >> 1) & 15])); };
#pragma unroll
      for (int v = 0; v < 64; v += 3) {
        int mx = bits(v);
        if (v + 1 < 64) mx = mx > bits(v + 1) ? mx : bits(v + 1);
        if (v + 2 < 64) mx = mx > bits(v + 2) ? mx : bits(v + 2);
        if (__ballot(mx > static_cast<int>(0xFF800000u))) {
          test_one(v);
          if (v + 1 < 64) test_one(v + 1);
          if (v + 2 < 64) test_one(v + 2);
        }
      }
      __builtin_amdgcn_wave_barrier();
      const int nq = qpos < kWaveQueue ? qpos : kWaveQueue;
      for (int i = lane; i < nq; i += 64) {
        const int code = qc[i];
        emit(code >> 8, static_cast<uint32_t>(code & 255), __uint_as_float(qv[i]));
      }
    }
    if (more) {
      __syncthreads();  // every wave is done with sB
#pragma unroll
      for (int h = 0; h < KH; ++h)
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x4*>(sB + (h * BN + brow + 32 * q) * BLD + bchunk * 8) = pf[h][q];
      __syncthreads();
    }
  }
  if constexpr (EPI == EPI_FILTER) {
    __syncthreads();
    if (tid < rows_here) sl.cnt[(b0 + tid) * sl.ns + strip] = rowcnt[tid] < sl.cap ? rowcnt[tid] : sl.cap;
  }
}

// ---- pass 2 without LDS operands ---------------------------------------------------------------------------------------
// The FILTER pass again, laid out so that a wave never waits for another: the four waves of a workgroup share the 128
// user rows and split each 128-column block into four 32-column slices.
//   * A wave's A operand -- its 128 rows x 64 k in the matrix cores' fragment layout -- is loaded ONCE, straight from
//     global memory (a fragment is 16 contiguous bytes of a row), and stays in 64 registers for the whole strip.
//   * Its B operand is 32 columns x 64 k per block = four 16-byte loads per lane, also straight in fragment layout, and
//     requested one block ahead.  No staging through LDS, no barrier inside the strip loop (bf16_tile_kernel: two per
//     block, and 1.5 LDS operand reads per MFMA).
//   * The row thresholds ride in the K dimension: -thr[row] = hi + mid + lo, three bf16 pieces that add up to the f32
//     value, are three extra k of A against ones in B (one v_mfma_f32_32x32x8_bf16_1k per 32-row tile: a fifth
//     of a full step), so the accumulators come out as score - thr without 64 seed registers or LDS reads per block.
//     Non-finite thresholds keep one piece (inf - inf would poison the sum); a NaN threshold lets everything pass, as
//     !(score < NaN) does.
// Same keys as bf16_tile_kernel<EPI_FILTER> up to the rounding of the shifted sum (both inside the slack of eps), same
// lists and overflow rules.  Measured at 4096 x 50 000, k = 20 (rocprofv3; bf16_tile_kernel<EPI_FILTER>: 56.4 us):
//   MFMAs alone (B loaded once)   25.8 us   978 MFMAs per SIMD: ~1.5 PFLOP/s chip-wide with the threshold step counted
//   + the block loads             34.6      (one block ahead: exposed only while nothing else fills the block)
//   + the 22 group tests          34.0      (free: independent vector work in the MFMAs' shadow)
//   + pushes, drain               49.1      ~9 records per wave and block in ~7 of the 22 groups, 13 instructions each
// Tried on the way: one ballot, branch and push per passing SCORE 70 us; a drain per block 68 (three dependent LDS round
// trips for 9 active lanes); a compare of the slot against the queue's end inside the push 54 (a vector compare feeding
// a branch: ~100 cycles, and a wave has one partner to hide behind); E laid out in fragment order (coalesced 1 KB
// loads) 31 instead of 35 for MFMAs + loads; warming the XCD's L2 with the strip's lines first: no change;
// 256 / 768 / 1024 workgroups instead of 512 (two per CU): 72 / 61 / 58 us; two B register sets refilled right behind
// their MFMAs (loads ~2 blocks ahead, exact s_waitcnt counts): 52 -- SQ_WAIT_ANY is 41 % of the waves' cycles, but it is
// not load latency that a deeper prefetch would hide (profiles/r02_pmc_mfma.txt: matrix cores busy 25 % of the kernel);
// E in fragment order (eb_chunk: contiguous 1 KB per load instruction instead of 32 quarter-used lines) 49.2 -> 48.7,
// kept; s_setprio 1 while testing and pushing 48.4 -> 47.8, kept; s_setprio on the MFMA phase instead: no change.
typedef short s16x4 __attribute__((ext_vector_type(4)));
// Developer knock-outs of this kernel (tools/build_variant.sh <name> "-DMI_FD_KNOCK=1" score.hip; tools/topk_chain_libs.sh), round 4,
// 4096 x 50 000, k = 20, D = 64, kernel time in the launch chain (41.4 us as built; an EMPTY kernel of this grid: 4.8 us):
//   MI_FD_KNOCK=2  no epilogue, no loads in the loop   24.3 us  (28 k matrix-pipe cycles = 11.3 us are the instructions themselves)
//   MI_FD_KNOCK=1  no epilogue                          32.0     (the loads, with no epilogue to hide behind)
//   as built                                            41.4     (the epilogue: ~60 vector + ~50 scalar instructions per wave and block)
//   MI_FD_KNOCK=2 -DMI_FD_PAD=40 / 80                    + 3.7 / + 7.9 us: independent vector instructions do NOT hide behind
//                   the matrix pipe on this chip -- each costs ~2.4 cycles of the SIMD whether or not the pipe is busy -- so
//                   the kernel's time is (matrix instructions) + (everything else), and "matrix cores busy" can only rise by
//                   issuing fewer other instructions per matrix instruction.  The sampling stride of pass 1 (2 / 3 / 4 / 6:
//                   40.1 / 39.9 / 41.4 / 43.7 us here) moves this kernel little: the tests of all 32 accumulators of a lane, not
//                   the pushes of the few that pass, are most of the epilogue.
#ifndef MI_FD_KNOCK
#define MI_FD_KNOCK 0
#endif
#ifndef MI_FD_PRIO
#define MI_FD_PRIO 1  // issue priority of a wave while it tests and pushes (its partner is then mostly issuing MFMAs): 48.4 -> 47.8 us
#endif
constexpr int kDirectQueue = 256;  // records (16 B) a wave of the direct filter kernel can hold between drains
constexpr int kNegInfBits = static_cast<int>(0xFF800000u);
constexpr int kRowLost = 1 << 24;  // flag in a row's LDS list counter: some of its candidates never reached a list

__device__ __forceinline__ uint32_t bf16_bits_rne(float v) {  // finite v
  const uint32_t u = __float_as_uint(v);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

// RARE (round 4): for searches in which a wave's block of 64 x 32 scores seldom holds a single passing one -- small k over a
// big catalogue: the 10 M-row knn search lets ~8 in a million through -- ONE test of the lane's whole set of accumulators
// comes first (a max3 tree: 16 vector instructions and a ballot) and the per-group tests, ballots and branches run only if
// some lane passes.  Vector and scalar instructions do not hide behind the matrix pipe in this kernel (MI_FD_PAD above), so
// ~40 fewer of them per block is time: 6.46 -> 6.33 ms for 4096 x 10 M, k = 2 with 64-row workgroups -- little: that search is bound by the L2s feeding B, see the rows-per-workgroup choice in score_topk_impl.  Chosen by the host from the expected number of
// passing scores per wave and block.
// Round 4: the user fragments of a 64-row workgroup (MT = 2) live in LDS, ONE copy per workgroup, instead of 32 (D <= 64) or 64
// (D <= 128) registers in every lane of every wave -- all four waves hold the same rows.  A matrix instruction's A operand is
// then a conflict-free ds_read_b128 (consecutive lanes, consecutive 16 bytes), the kernel needs 117 registers instead of 173 at
// D = 128 and runs four waves per SIMD instead of two, which is what hides the B loads: filter pass at 4096 x 50 000, k = 20,
// D = 128: 75.8 -> 64.8 us (the call 136.6 -> 129.0); D = 64 (already four waves): 41.4 -> 39.3 (82.4 -> 80.0).  The 128-row
// workgroups of the large-catalogue searches (MT = 4) keep their fragments in registers: there the extra waves only add
// pressure on the L2s that feed B (4096 x 10 M, k = 2: 6.3 ms against 7.7 with the LDS copy).  Half of the fragments in registers
// and half in LDS (D = 128, three waves per SIMD): 83.6 us -- it is the occupancy, not the LDS operand reads, that this kernel lives
// on.  MI_FD_ALDS=0: rounds 2-3.
#ifndef MI_FD_ALDS
#define MI_FD_ALDS 1
#endif
template <bool MASKED, int MT, int KH = 1, bool RARE = false>  // MT 32-row tiles = the workgroup's user rows (every wave holds all of them)
__global__ __launch_bounds__(kBlock, (MI_FD_ALDS && MT == 2) ? 4 : ((MT == 4 || KH == 2) ? 2 : 4)) void bf16_filter_direct_kernel(const __bf16* __restrict__ Ub, int64_t B,
                                                                       const __bf16* __restrict__ Eb, int64_t N,
                                                                       const float* __restrict__ thr, TopkArgs ta, StripLists sl,
                                                                       int nvisit) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int RM = MT * 32, NV = MT * 16, NG = (NV + 2) / 3;  // rows; accumulators per lane; groups of three
  int* rowcnt = reinterpret_cast<int*>(smem);  // [RM] entries of the (row, strip) lists so far
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int i32 = lane & 31, hh = lane >> 5;
  const int64_t b0 = static_cast<int64_t>(blockIdx.y) * RM;
  const int strip = blockIdx.x, nstrip = gridDim.x;
  if (strip >= nvisit) {
    if (tid < RM && b0 + tid < B) sl.cnt[(b0 + tid) * sl.ns + strip] = 0;
    return;
  }
  const int rows_here = (B - b0 < RM) ? static_cast<int>(B - b0) : RM;
  const uint32_t n_cols = static_cast<uint32_t>(N);
  const uint32_t skip = ta.n_skip_low < N ? static_cast<uint32_t>(ta.n_skip_low) : n_cols;
  if (tid < RM) rowcnt[tid] = 0;

  // the wave's B slice of the strip's first block, then A
  const int ccol = wv * 32 + i32;  // this lane's column inside a block
  constexpr int NKS = 4 * KH;  // k-steps of 16 per row
  u32x4 bq[NKS], bn[NKS];
  {
    const int64_t n0 = static_cast<int64_t>(strip) * BN;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) bq[ks] = *reinterpret_cast<const u32x4*>(Eb + eb_chunk<KH>(n0 + ccol, ks * 2 + hh));
  }
  constexpr bool kALds = MI_FD_ALDS != 0 && MT == 2;
  u32x4 aq[kALds ? 1 : MT][NKS];
  u32x4* sA = reinterpret_cast<u32x4*>(rowcnt + RM) + 4 * (kDirectQueue + 1);  // (kALds) [MT][NKS][64 lanes] behind the four queues
  s16x4 at[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int r = m * 32 + i32;
    const int64_t ra = (r < rows_here) ? b0 + r : B - 1;
    if constexpr (kALds) {  // every wave would hold the same fragments: wave w stages k-steps w, w + 4, ...
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
        if ((ks & 3) == wv) sA[(m * NKS + ks) * 64 + lane] = *reinterpret_cast<const u32x4*>(Ub + ra * (64 * KH) + ks * 16 + hh * 8);
    } else {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) aq[m][ks] = *reinterpret_cast<const u32x4*>(Ub + ra * (64 * KH) + ks * 16 + hh * 8);
    }
    float nt = (r < rows_here) ? 0.f - thr[b0 + r] : -__builtin_inff();
    if (nt != nt) nt = __builtin_inff();
    uint32_t p0, p1 = 0u, p2 = 0u;
    if (__builtin_fabsf(nt) == __builtin_inff()) {
      p0 = __float_as_uint(nt) >> 16;
    } else {
      p0 = bf16_bits_rne(nt);
      const float r1 = nt - __uint_as_float(p0 << 16);  // exact
      p1 = bf16_bits_rne(r1);
      const float r2 = r1 - __uint_as_float(p1 << 16);  // exact
      p2 = bf16_bits_rne(r2);
      if ((p0 & 0x7FFFu) == 0x7F80u) p1 = p2 = 0u;      // (rounded up to inf: |nt| within half a bf16 ulp of FLT_MAX)
    }
    at[m] = hh == 0 ? s16x4{static_cast<short>(p0), static_cast<short>(p1), static_cast<short>(p2), 0} : s16x4{0, 0, 0, 0};
  }
  const s16x4 ones = hh == 0 ? s16x4{0x3F80, 0x3F80, 0x3F80, 0} : s16x4{0, 0, 0, 0};
  __syncthreads();  // rowcnt zeroed

  uint64_t* cand_wg = sl.cand + (b0 * sl.ns + strip) * sl.cap;
  const int row_stride = sl.ns * sl.cap;
  u32x4* queue = reinterpret_cast<u32x4*>(rowcnt + RM) + wv * (kDirectQueue + 1);  // {block, group, lane; three score images}; + a dump slot
  int qpos = 0;  // records in the queue (wave-uniform)
  // Drain: 64 records at a time, a lane per record: list slots for its (up to three) passing scores -- the three LDS
  // atomics in flight together -- then keys and stores.
  auto drain = [&]() {
    const int nq = qpos < kDirectQueue ? qpos : kDirectQueue;
    for (int i = lane; i < nq; i += 64) {
      const u32x4 rec = queue[i];
      const int v0 = static_cast<int>((rec.x >> 6) & 63u), src = static_cast<int>(rec.x & 63u);
      const uint32_t col = (rec.x >> 12) * static_cast<uint32_t>(BN) + static_cast<uint32_t>(wv * 32 + (src & 31));
      const bool col_ok = col < n_cols && col >= skip;
      const uint32_t image[3] = {rec.y, rec.z, rec.w};
      int lrow[3], p[3];
      bool ok[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int v = v0 + t;
        lrow[t] = (v >> 4) * 32 + (v & 3) + 8 * ((v & 15) >> 2) + 4 * (src >> 5);
        ok[t] = static_cast<int>(image[t]) > kNegInfBits && col_ok && lrow[t] < rows_here;  // (v past the last accumulator: image -inf)
        if constexpr (MASKED) {
          if (ok[t]) ok[t] = !((sl.mask[(b0 + lrow[t]) * sl.mask_words + (col >> 6)] >> (col & 63u)) & 1ull);
        }
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) p[t] = ok[t] ? atomicAdd(&rowcnt[lrow[t]], 1) : 0;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        if (ok[t]) {
          const uint64_t packed = (static_cast<uint64_t>(order_key(__uint_as_float(image[t]))) << 32) | (0xFFFFFFFFu - col);
          if (p[t] < sl.cap) {
            cand_wg[lrow[t] * row_stride + p[t]] = packed;
          } else if (__atomic_load_n(&sl.ovf_cnt[b0 + lrow[t]], __ATOMIC_RELAXED) <= kOvfCap) {
            // (a row that has already run over is lost to the exact fallback: no point in counting on)
            const int p2 = atomicAdd(&sl.ovf_cnt[b0 + lrow[t]], 1);
            if (p2 < kOvfCap) sl.ovf[(b0 + lrow[t]) * kOvfCap + p2] = packed;
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    qpos = 0;
  };

  for (int j = strip; j < nvisit; j += nstrip) {
    {
      const int64_t n1 = static_cast<int64_t>(j + nstrip < nvisit ? j + nstrip : j) * BN;  // (the last block re-reads itself)
#if MI_FD_KNOCK == 2  // developer knock-out: no loads in the loop either
      (void)n1;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) bn[ks] = bq[ks];
#else
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) bn[ks] = *reinterpret_cast<const u32x4*>(Eb + eb_chunk<KH>(n1 + ccol, ks * 2 + hh));
#endif
    }
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      f32x16 c;
#pragma unroll
      for (int r = 0; r < 16; ++r) c[r] = 0.f;
      c = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(at[m], ones, c, 0, 0, 0);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const u32x4 af = kALds ? sA[(m * NKS + ks) * 64 + lane] : aq[kALds ? 0 : m][ks];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bq[ks]), c, 0, 0, 0);
      }
      acc[m] = c;
    }
    // acc[m][r]: row m 32 + (r & 3) + 8 (r >> 2) + 4 hh, column n0 + ccol
    // The scores are tested three at a time: as signed integers the float images that FAIL (x < 0) are exactly those
    // <= 0xFF800000 (-inf) -- NaNs of either sign and everything >= +0 lie above; -0 cannot occur (a sum that cancels
    // rounds to +0) -- so "one of my three passes" is one v_max3_i32 and one compare.  All 22 group ballots are taken
    // first (independent VALU work, the masks stay in scalar registers); the branches that follow depend on scalar
    // registers only.  A lane with a passing score pushes ONE record -- its three score images and where they sit --
    // into the wave's queue (slot = rank among the pushing lanes).  The queue lives across blocks and is drained when
    // it may not hold another block's records (a block leaves ~9): what matters in this kernel is the NUMBER of
    // vector instructions next to the 20 MFMAs of a block (they share the SIMD's issue: knocking the epilogue out
    // took 70 us to 34), and a drain costs ~100 of them whether 9 lanes have a record or 64.
#if MI_FD_KNOCK == 1 || MI_FD_KNOCK == 2  // developer knock-out: no epilogue (the accumulators stay alive through an impossible store)
    {
      int keepi = 0;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; r += 3) keepi |= __float_as_uint(acc[m][r]) == 0x12345678u;
      if (keepi) rowcnt[0] = 1;
#ifdef MI_FD_PAD  // developer probe: MI_FD_PAD independent vector instructions per block -- do they hide behind the matrix pipe?
      {
        int pad0 = lane, pad1 = lane + 1;
#pragma unroll
        for (int q = 0; q < MI_FD_PAD / 2; ++q) {
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(pad0) : "v"(j));
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(pad1) : "v"(j));
        }
        if ((pad0 ^ pad1) == 0x7FFFFFFF) rowcnt[1] = 1;
      }
#endif
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) bq[ks] = bn[ks];
      continue;
    }
#endif
    if (qpos > kDirectQueue - 128) drain();
#if MI_FD_PRIO
    __builtin_amdgcn_s_setprio(MI_FD_PRIO);
#endif
    const uint32_t tag0 = (static_cast<uint32_t>(j) << 12) | static_cast<uint32_t>(lane);
    auto bits = [&](int v) { return v < NV ? static_cast<int>(__float_as_uint(acc[v >> 4][v & 15])) : kNegInfBits; };
    if constexpr (RARE) {
      // the largest of the lane's NV images (as signed integers: see below) by a tree of three-input maxima
      int t[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int x0 = bits(3 * g), x1 = bits(3 * g + 1), x2 = bits(3 * g + 2);
        int mx = x0 > x1 ? x0 : x1;
        t[g] = mx > x2 ? mx : x2;
      }
      int n = NG;
#pragma unroll
      for (int level = 0; level < 3; ++level) {  // 11 (22) -> 4 (8) -> 2 (3) -> 1
        const int m3 = (n + 2) / 3;
#pragma unroll
        for (int q = 0; q < NG; ++q)
          if (q < m3) {
            const int a = t[3 * q], b = 3 * q + 1 < n ? t[3 * q + 1] : kNegInfBits, c = 3 * q + 2 < n ? t[3 * q + 2] : kNegInfBits;
            int mx = a > b ? a : b;
            t[q] = mx > c ? mx : c;
          }
        n = m3;
      }
      if (__ballot(t[0] > kNegInfBits) == 0) {
#if MI_FD_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) bq[ks] = bn[ks];
        continue;
      }
    }
    uint64_t gm[NG];
    uint64_t any = 0;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int x0 = bits(3 * g), x1 = bits(3 * g + 1), x2 = bits(3 * g + 2);
      int mx = x0 > x1 ? x0 : x1;
      mx = mx > x2 ? mx : x2;
      gm[g] = __ballot(mx > kNegInfBits);
      any |= gm[g];
    }
    if (any) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (gm[g]) {
          const int x0 = bits(3 * g), x1 = bits(3 * g + 1), x2 = bits(3 * g + 2);
          int mx = x0 > x1 ? x0 : x1;
          mx = mx > x2 ? mx : x2;
          if (mx > kNegInfBits) {  // (the lanes of gm[g])
            // No test of the slot here (a vector compare feeding a branch is a ~100-cycle chain, seven times per
            // block, and a wave has one other wave to hide it behind): records past the end land in a dump slot and
            // the count tells afterwards.
            int slot = static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(gm[g] >> 32),
                                            __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(gm[g]), static_cast<uint32_t>(qpos))));
            slot = slot < kDirectQueue ? slot : kDirectQueue;
            queue[slot] = u32x4{tag0 | static_cast<uint32_t>(3 * g << 6), static_cast<uint32_t>(x0), static_cast<uint32_t>(x1),
                                static_cast<uint32_t>(x2)};
          }
          qpos += __popcll(gm[g]);
        }
      }
      if (qpos > kDirectQueue) {
        // more records than the queue holds (a dozen rows of equal scores in this block): which were dropped is not
        // known any more, so all rows of the workgroup take the exact fallback
        for (int r = lane; r < RM; r += 64) atomicOr(&rowcnt[r], kRowLost);
        qpos = kDirectQueue;
      }
    }
#if MI_FD_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) bq[ks] = bn[ks];
  }
  drain();
  __syncthreads();
  if (tid < rows_here) {
    const int n = rowcnt[tid];
    sl.cnt[(b0 + tid) * sl.ns + strip] = n < sl.cap ? n : sl.cap;
    if (n & kRowLost) atomicAdd(&sl.ovf_cnt[b0 + tid], kOvfCap + 1);  // candidates of this row were dropped: exact fallback
  }
}

// ---- pass 1 in the direct layout (round 4) -------------------------------------------------------------------------------
// The sampled tile maxima on the structure of bf16_filter_direct_kernel instead of the LDS-staged bf16_tile_kernel: a
// 64-row workgroup keeps ONE copy of its user fragments in LDS, each of its four waves takes a 32-column slice of a
// 128-column block with its E fragments straight from global memory one block ahead, no barrier inside the strip loop.
// The product is taken transposed (E slice x U^T), so a lane holds 16 columns of ONE user row per accumulator and the
// maximum of a (row, 32-column tile) is 8 in-lane operations and one exchange with lane ^ 32 -- the tiles are 32 columns
// wide here (NT = 4 per visited block; bf16_tile_kernel: 2), which also makes the k-th best tile maximum a tighter bound.
// Per full product this structure costs ~27 us at D = 64 where the LDS-staged kernel costs ~48: pass 1 (every fourth
// block) 16.8 -> see DESIGN.md section 5d.1.  F32A / FoldU as in bf16_tile_kernel: the user rows are converted while they
// are staged, and the workgroups of strip 0 leave the bf16 copy, the squared norms and the zeroed overflow counters.
template <bool MASKED, int KH, bool F32A>
__global__ __launch_bounds__(kBlock, 4) void bf16_tilemax_direct_kernel(const __bf16* __restrict__ Ub, int64_t B,
                                                                        const __bf16* __restrict__ Eb, int64_t N, TopkArgs ta,
                                                                        StripLists sl, int nvisit, FoldU fold) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int RM = 64, NKS = 4 * KH;
  u32x4* sA = reinterpret_cast<u32x4*>(smem);  // [2 tiles of 32 rows][NKS][64 lanes]: the A operand's fragments
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int i32 = lane & 31, hh = lane >> 5;
  const int stride = ta.col_stride > 1 ? ta.col_stride : 1;
  const int64_t b0 = static_cast<int64_t>(blockIdx.y) * RM;
  const int strip = blockIdx.x, nstrip = gridDim.x;
  if (strip >= nvisit) return;
  const int rows_here = (B - b0 < RM) ? static_cast<int>(B - b0) : RM;
  const uint32_t n_cols = static_cast<uint32_t>(N);
  const uint32_t skip = ta.n_skip_low < N ? static_cast<uint32_t>(ta.n_skip_low) : n_cols;

  const int ccol = wv * 32 + i32;  // this lane's column inside a block
  u32x4 bq[NKS], bn[NKS];
  {
    const int64_t n0 = static_cast<int64_t>(strip) * stride * BN;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) bq[ks] = *reinterpret_cast<const u32x4*>(Eb + eb_chunk<KH>(n0 + ccol, ks * 2 + hh));
  }
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int r = m * 32 + i32;
    const int64_t ra = (r < rows_here) ? b0 + r : B - 1;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
      if ((ks & 3) == wv) {  // every wave would hold the same fragments: wave w stages k-steps w, w + 4
        u32x4 pk;
        if constexpr (F32A) {
          const float4* src = reinterpret_cast<const float4*>(fold.U + ra * (64 * KH) + ks * 16 + hh * 8);
          const float4 x0 = src[0], x1 = src[1];
          pk.x = __builtin_bit_cast(uint32_t, bf16x2{static_cast<__bf16>(x0.x), static_cast<__bf16>(x0.y)});
          pk.y = __builtin_bit_cast(uint32_t, bf16x2{static_cast<__bf16>(x0.z), static_cast<__bf16>(x0.w)});
          pk.z = __builtin_bit_cast(uint32_t, bf16x2{static_cast<__bf16>(x1.x), static_cast<__bf16>(x1.y)});
          pk.w = __builtin_bit_cast(uint32_t, bf16x2{static_cast<__bf16>(x1.z), static_cast<__bf16>(x1.w)});
        } else {
          pk = *reinterpret_cast<const u32x4*>(Ub + ra * (64 * KH) + ks * 16 + hh * 8);
        }
        sA[(m * NKS + ks) * 64 + lane] = pk;
      }
  }
  if constexpr (F32A) {
    if (strip == 0) {
      // what the later kernels read, exactly as bf16_tile_kernel<..., F32A> leaves it: thread (erow, eoff) takes eight
      // consecutive floats of a row per half, the eight threads of a row add their squares up for the norm
      const int erow = tid >> 3, eoff = (tid & 7) * 8;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int r = erow + 32 * q;
        const int64_t ra = (r < rows_here) ? b0 + r : B - 1;
        float sq = 0.f;
#pragma unroll
        for (int h = 0; h < KH; ++h) {
          const float4* src = reinterpret_cast<const float4*>(fold.U + ra * (64 * KH) + h * 64 + eoff);
          const float4 x0 = src[0], x1 = src[1];
          u32x4 pk;
          pk.x = __builtin_bit_cast(uint32_t, bf16x2{static_cast<__bf16>(x0.x), static_cast<__bf16>(x0.y)});
          pk.y = __builtin_bit_cast(uint32_t, bf16x2{static_cast<__bf16>(x0.z), static_cast<__bf16>(x0.w)});
          pk.z = __builtin_bit_cast(uint32_t, bf16x2{static_cast<__bf16>(x1.x), static_cast<__bf16>(x1.y)});
          pk.w = __builtin_bit_cast(uint32_t, bf16x2{static_cast<__bf16>(x1.z), static_cast<__bf16>(x1.w)});
          if (r < rows_here) *reinterpret_cast<u32x4*>(fold.Ub_out + ra * (64 * KH) + h * 64 + eoff) = pk;
          sq = __builtin_fmaf(x0.x, x0.x, sq); sq = __builtin_fmaf(x0.y, x0.y, sq);
          sq = __builtin_fmaf(x0.z, x0.z, sq); sq = __builtin_fmaf(x0.w, x0.w, sq);
          sq = __builtin_fmaf(x1.x, x1.x, sq); sq = __builtin_fmaf(x1.y, x1.y, sq);
          sq = __builtin_fmaf(x1.z, x1.z, sq); sq = __builtin_fmaf(x1.w, x1.w, sq);
        }
        sq = sq + __shfl_xor(sq, 1, 64);
        sq = sq + __shfl_xor(sq, 2, 64);
        sq = sq + __shfl_xor(sq, 4, 64);
        if (r < rows_here && (tid & 7) == 0) {
          fold.u2[ra] = sq;
          fold.zero_rows[ra] = 0;
        }
      }
    }
  }
  __syncthreads();

  for (int j = strip; j < nvisit; j += nstrip) {
    const int64_t n0 = static_cast<int64_t>(j) * stride * BN;
    {
      const int64_t n1 = static_cast<int64_t>(j + nstrip < nvisit ? j + nstrip : j) * stride * BN;  // (the last block re-reads itself)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) bn[ks] = *reinterpret_cast<const u32x4*>(Eb + eb_chunk<KH>(n1 + ccol, ks * 2 + hh));
    }
    uint64_t mw[2] = {0ull, 0ull};  // MASKED: the exclusion bits of this lane's two user rows for the 64 columns its slice lies in
    if constexpr (MASKED) {
      const int64_t tw = (n0 >> 6) + (wv >> 1);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int lr = m * 32 + i32;
        if (lr < rows_here && tw < sl.mask_words) mw[m] = sl.mask[(b0 + lr) * sl.mask_words + tw];
      }
    }
    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      f32x16 c;
#pragma unroll
      for (int r = 0; r < 16; ++r) c[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bq[ks]), __builtin_bit_cast(bf16x8, sA[(m * NKS + ks) * 64 + lane]), c, 0, 0, 0);
      acc[m] = c;
    }
    // acc[m][r]: column n0 + wv 32 + (r & 3) + 8 (r >> 2) + 4 hh, user row m 32 + i32.  NaN sorts first (torch.topk): max
    // drops NaNs, so they are tracked through the largest |bits| seen (as bf16_tile_kernel does).
    const int64_t c_lo = n0 + wv * 32, c_hi = c_lo + 32;
    const bool all_cols = c_hi <= N && c_lo >= ta.n_skip_low;  // uniform; false only at the two ends of the catalogue
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      float best = -__builtin_inff();
      uint32_t amax = 0u;
      bool any = false;
      if (all_cols) {
        const uint32_t w32 = static_cast<uint32_t>(mw[m] >> ((wv & 1) * 32 + 4 * hh));  // bit (r & 3) + 8 (r >> 2) = this lane's column r
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          float v0 = acc[m][r], v1 = acc[m][r + 1];
          if constexpr (MASKED) {  // an excluded column counts as -inf: sign-extended bit, bit-field insert
            const int b0_ = (r & 3) + 8 * (r >> 2), b1_ = ((r + 1) & 3) + 8 * ((r + 1) >> 2);
            const uint32_t e0 = static_cast<uint32_t>(static_cast<int>(w32 << (31 - b0_)) >> 31);
            const uint32_t e1 = static_cast<uint32_t>(static_cast<int>(w32 << (31 - b1_)) >> 31);
            v0 = __uint_as_float((__float_as_uint(v0) & ~e0) | (0xFF800000u & e0));
            v1 = __uint_as_float((__float_as_uint(v1) & ~e1) | (0xFF800000u & e1));
          }
          best = __builtin_fmaxf(__builtin_fmaxf(best, v0), v1);
          const uint32_t a0 = __float_as_uint(v0) & 0x7FFFFFFFu, a1 = __float_as_uint(v1) & 0x7FFFFFFFu;
          amax = a0 > amax ? a0 : amax;
          amax = a1 > amax ? a1 : amax;
        }
        any = true;
      } else {
        uint32_t cbase = static_cast<uint32_t>(c_lo) + 4 * hh;
        asm volatile("" : "+v"(cbase));  // (not hoisted out of the rare branch)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint32_t col = cbase + (r & 3) + 8 * (r >> 2);
          const bool excluded = MASKED && ((mw[m] >> (col & 63u)) & 1ull);
          if (col < n_cols && col >= skip && !excluded) {
            best = __builtin_fmaxf(best, acc[m][r]);
            const uint32_t a0 = __float_as_uint(acc[m][r]) & 0x7FFFFFFFu;
            amax = a0 > amax ? a0 : amax;
            any = true;
          }
        }
      }
      uint32_t key = !any ? 0u : (amax > 0x7F800000u ? 0xFFFFFFFFu : order_key(best));
      const uint32_t other = __shfl_xor(key, 32, 64);
      key = other > key ? other : key;
      const int lrow = m * 32 + i32;
      if (hh == 0 && lrow < rows_here) ta.tilemax[(b0 + lrow) * ta.NT + j * 4 + wv] = key;
    }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) bq[ks] = bn[ks];
  }
}

// finalize for the bf16 prefilter, one WAVE per row (4 rows per workgroup, no barriers on the common path): gather the
// row's strip lists, cut by the k-th best bf16 key, exact f32 key of the survivors (the oracle's fmaf chain), rank sort.
// (One 256-thread workgroup per row spent 48 us on 4096 rows: barriers between six short phases, and a rank-counting
// pass over all ~200 candidates with 64-bit compares, 26 us of VALU work; the k-th key is found here by a 32-step
// bitwise search instead, ~15 instructions per step.)
constexpr int kStripSlots = 1024;  // ns * cap <= kStripSlots
constexpr int kFinRows = kBlock / 64;

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
  int inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  return inc;
}

__global__ __launch_bounds__(kBlock) void topk_finalize_exact_kernel(const float* __restrict__ U, const float* __restrict__ E,
                                                                     int64_t B, int64_t N, int64_t D, int k,
                                                                     int64_t n_skip_low, StripLists sl,
                                                                     const float* __restrict__ eps_row,
                                                                     float* __restrict__ vals, int64_t* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // kFinRows x (ns cap + kOvfLds) gathered entries
  __shared__ int cnt_all[kFinRows][128], off_all[kFinRows][128];
  __shared__ int ovf_row[kFinRows];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int row = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * kFinRows + wv);  // wave-uniform
  const int lds_cap = sl.ns * sl.cap + kOvfLds;  // entries of the row kept in LDS; a longer overflow list is read in place
  uint64_t* lc = reinterpret_cast<uint64_t*>(smem) + wv * lds_cap;
  int* lcnt = cnt_all[wv];
  int* loff = off_all[wv];
  bool overflow = false;
  if (row < B) {
    const int ns = sl.ns, cap = sl.cap;  // ns <= 128, cap a power of two, ns * cap <= kStripSlots
    const int cap_shift = __builtin_ctz(cap);
    // the lists' lengths and their offsets in the gathered array
    // (counts are clamped to their lists' capacities HERE as well as where they are written: every address this kernel
    //  forms from workspace contents then stays inside the workspace whatever those contents are -- a filter pass that
    //  did not write a counter, as a development build of it once did not ("the fault of record"),
    //  costs wrong candidates and the exact fallback, never an access outside the buffers)
    auto clamp_cnt = [&](int c) { return c < 0 ? 0 : (c > cap ? cap : c); };
    const int c0 = lane < ns ? clamp_cnt(sl.cnt[static_cast<int64_t>(row) * ns + lane]) : 0;
    const int c1 = lane + 64 < ns ? clamp_cnt(sl.cnt[static_cast<int64_t>(row) * ns + lane + 64]) : 0;
    int oc_raw = sl.ovf_cnt[row];
    oc_raw = oc_raw < 0 ? kOvfCap + 1 : oc_raw;  // (a negative count can only be garbage: exact fallback)
    const int i0 = wave_incl_scan(c0, lane), t0 = __shfl(i0, 63, 64);
    const int i1 = wave_incl_scan(c1, lane), t1 = __shfl(i1, 63, 64);
    lcnt[lane] = c0;
    lcnt[lane + 64] = c1;
    loff[lane] = i0 - c0;
    loff[lane + 64] = t0 + i1 - c1;
    overflow = oc_raw > kOvfCap;
    const int oc = oc_raw < kOvfCap ? oc_raw : kOvfCap;
    const int n = t0 + t1 + oc;
    __builtin_amdgcn_wave_barrier();
    if (!overflow) {
      const uint64_t* lists = sl.cand + static_cast<int64_t>(row) * ns * cap;
      for (int s0 = 0; s0 < ns * cap; s0 += 256) {  // 4 independent loads per lane and round
        uint64_t v[4];
        int dst[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int slot = s0 + q * 64 + lane;
          const int g = slot >> cap_shift, j = slot & (cap - 1);
          dst[q] = -1;
          if (slot < ns * cap && j < lcnt[g]) {
            dst[q] = loff[g] + j;
            v[q] = lists[slot];
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (dst[q] >= 0) lc[dst[q]] = v[q];
      }
      const uint64_t* ovf_row_p = sl.ovf + static_cast<int64_t>(row) * kOvfCap;
      const int n_lds = t0 + t1 + (oc < kOvfLds ? oc : kOvfLds);
      for (int j = lane; j < n_lds - t0 - t1; j += 64) lc[t0 + t1 + j] = ovf_row_p[j];
      auto entry = [&](int i) { return i < n_lds ? lc[i] : ovf_row_p[i - t0 - t1]; };  // i < n
      __builtin_amdgcn_wave_barrier();
      // Stage 1: v_k = the k-th best bf16 key: the largest T with at least kk keys >= T, bit by bit.  The k candidates
      // at or above it have exact scores >= v_k - eps, so a candidate whose bf16 score is below v_k - 2 eps (exact
      // score < v_k - eps) cannot be in the top k: only the others are re-scored exactly.  eps is the bound the filter
      // pass used (the keys are shifted by the row's threshold, which cancels in the comparison).
      const int kk = n < k ? n : k;
      uint32_t mykey[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) mykey[q] = (q * 64 + lane < n) ? static_cast<uint32_t>(entry(q * 64 + lane) >> 32) : 0u;
      uint32_t T = 0u;
      if (kk > 0 && kk <= 64) {
        // Any LOWER bound of the k-th best key serves (a lower cut only lets a few more candidates through to the exact
        // re-score, which decides).  The kk-th best of the 64 per-lane maxima is one -- the lanes' maxima are kk distinct
        // candidates at least as good -- and so is that value with its low 16 bits cleared (2^-7 relative of a shifted
        // score: far inside the 2 eps the cut gives away anyway): 16 steps of ONE ballot instead of 32 steps of four.
        // (round 4: this search was 5.8 of the kernel's 21.4 us at 4096 x 50 000, k = 20.)
        uint32_t lmax = mykey[0];
#pragma unroll
        for (int q = 1; q < 4; ++q) lmax = mykey[q] > lmax ? mykey[q] : lmax;
        for (int b = 256; b < n; b += 64) {  // lists longer than 256 entries (rare): from LDS / the overflow list
          const uint32_t kx = b + lane < n ? static_cast<uint32_t>(entry(b + lane) >> 32) : 0u;
          lmax = kx > lmax ? kx : lmax;
        }
        for (int bit = 31; bit >= 16; --bit) {
          const uint32_t c = T | (1u << bit);
          if (__popcll(__ballot(lmax >= c)) >= kk) T = c;  // (a lane without a candidate holds 0 and never counts: c > 0)
        }
      } else if (kk > 0) {  // k > 64: the exact k-th best key, bit by bit over every candidate
        for (int bit = 31; bit >= 0; --bit) {
          const uint32_t c = T | (1u << bit);
          int have = 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) have += __popcll(__ballot(mykey[q] >= c));  // (an absent slot's key 0 never counts: c > 0)
          for (int b = 256; b < n; b += 64)  // lists longer than 256 entries (rare): from LDS / the overflow list
            have += __popcll(__ballot(b + lane < n && static_cast<uint32_t>(entry(b + lane < n ? b + lane : 0) >> 32) >= c));
          if (have >= kk) T = c;
        }
      }
      const float eps = eps_row[row];
      const float cut = key_to_float(T) - 2.f * eps;  // NaN / -inf when anything is not finite: then every candidate is re-scored
      // Stage 2: exact f32 score of the survivors -- 16 independent float4 loads of the item row, then the oracle's chain
      // acc = fma(u[d], e[d], acc), d = 0..63 from +0 (the order the f32 MFMA kernel runs); the user row comes in
      // scalar registers
      const float* u = U + static_cast<int64_t>(row) * D;
      int m = 0;
      for (int base = 0; base < n; base += 64) {  // survivors to the front (a survivor's new slot is never above its old one)
        const int i = base + lane;
        const uint64_t me = i < n ? entry(i) : 0;
        const bool keep = i < n && !(key_to_float(static_cast<uint32_t>(me >> 32)) < cut);
        const uint64_t mask = __ballot(keep);
        if (m + __popcll(mask) > lds_cap) {  // more survivors than LDS holds (never seen: ~45 of ~110 survive): exact fallback
          overflow = true;
          break;
        }
        __builtin_amdgcn_wave_barrier();  // every lane holds its lc[i] before slots <= i are rewritten
        if (keep)
          lc[m + static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u)))] = me;
        m += __popcll(mask);
      }
      __builtin_amdgcn_wave_barrier();
      if (overflow) m = 0;  // (the row is redone below)
      for (int base = 0; base < m; base += 64) {  // usually one round: all the gathers of the row in flight together
        const int i = base + lane;
        if (i < m) {
          const uint32_t inv = static_cast<uint32_t>(lc[i]);
          float acc = 0.f;
          if (static_cast<int64_t>(0xFFFFFFFFu - inv) >= N) {  // no such column (a list entry the filter pass never wrote:
            lc[i] = 0;                                          // see clamp_cnt above): ranked below every real key
            continue;
          }
          if (D == 64) {
            const float4* e4 = reinterpret_cast<const float4*>(E + static_cast<int64_t>(0xFFFFFFFFu - inv) * 64);
            float4 ev[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) ev[c] = e4[c];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
              acc = __builtin_fmaf(u[4 * c + 0], ev[c].x, acc);
              acc = __builtin_fmaf(u[4 * c + 1], ev[c].y, acc);
              acc = __builtin_fmaf(u[4 * c + 2], ev[c].z, acc);
              acc = __builtin_fmaf(u[4 * c + 3], ev[c].w, acc);
            }
          } else if (D == 128) {  // two 64-float halves, the same chain continued (16 independent float4 loads each)
            const float4* e4 = reinterpret_cast<const float4*>(E + static_cast<int64_t>(0xFFFFFFFFu - inv) * 128);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              float4 ev[16];
#pragma unroll
              for (int c = 0; c < 16; ++c) ev[c] = e4[half * 16 + c];
#pragma unroll
              for (int c = 0; c < 16; ++c) {
                acc = __builtin_fmaf(u[half * 64 + 4 * c + 0], ev[c].x, acc);
                acc = __builtin_fmaf(u[half * 64 + 4 * c + 1], ev[c].y, acc);
                acc = __builtin_fmaf(u[half * 64 + 4 * c + 2], ev[c].z, acc);
                acc = __builtin_fmaf(u[half * 64 + 4 * c + 3], ev[c].w, acc);
              }
            }
          } else {  // other widths: the oracle's chain for that width (DotKeys), zero-padded to the f32 kernel's K chunk
            const float* e = E + static_cast<int64_t>(0xFFFFFFFFu - inv) * D;
            for (int64_t d = 0; d < D; ++d) acc = __builtin_fmaf(u[d], e[d], acc);
            if (D % KC) acc = __builtin_fmaf(0.f, 0.f, acc);
          }
          lc[i] = (static_cast<uint64_t>(order_key(acc)) << 32) | inv;
        }
      }
      __builtin_amdgcn_wave_barrier();
      // rank sort of the m survivors; the best kk go out
      const int kout = m < k ? m : k;
      float* vals_row = vals + static_cast<int64_t>(row) * k;
      int64_t* idx_row = idx + static_cast<int64_t>(row) * k;
      for (int i = lane; i < m; i += 64) {
        const uint64_t me = lc[i];
        int rank = 0;
        for (int j = 0; j < m; ++j) rank += (lc[j] > me) ? 1 : 0;
        if (rank < kout) {
          vals_row[rank] = key_to_float(static_cast<uint32_t>(me >> 32));
          idx_row[rank] = static_cast<int64_t>(0xFFFFFFFFu - static_cast<uint32_t>(me));
        }
      }
      for (int t = kout + lane; t < k; t += 64) {  // fewer than k candidate columns exist
        vals_row[t] = -__builtin_inff();
        idx_row[t] = -1;
      }
    }
  }
  // rows whose overflow list ran over: exact selection over the whole catalogue by the full workgroup (rare)
  if (lane == 0) ovf_row[wv] = overflow ? 1 : 0;
  __syncthreads();
  for (int w = 0; w < kFinRows; ++w) {
    if (!ovf_row[w]) continue;
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kFinRows + w;
    const uint64_t* mrow = sl.mask ? sl.mask + r * sl.mask_words : nullptr;
    select_topk_row(AnyDotKeys{U + r * D, E, D, mrow}, N, k, n_skip_low, vals + r * k, idx + r * k);
    if (mrow) {  // excluded columns rank below every real key and are blanked afterwards (never returned)
      __syncthreads();
      for (int t = threadIdx.x; t < k; t += kBlock) {
        const int64_t c = idx[r * k + t];
        if (c >= 0 && ((mrow[c >> 6] >> (c & 63)) & 1ull)) {
          idx[r * k + t] = -1;
          vals[r * k + t] = -__builtin_inff();
        }
      }
    }
    __syncthreads();
  }
}

static size_t full_sort_lds() { return static_cast<size_t>(BM + BN) * LDK * sizeof(float); }

template <bool VEC, int EPI>
static int launch_tiled(const float* U, int64_t B, const float* E, int64_t N, int64_t D, const float* bias, float* S,
                        int64_t ldS, hipStream_t st, TopkArgs ta = TopkArgs{}) {
  int64_t nblk = (N + BN - 1) / BN;
  if (EPI == EPI_TILEMAX && ta.col_stride > 1) nblk = (nblk + ta.col_stride - 1) / ta.col_stride;
  const dim3 grid(static_cast<unsigned>(nblk), static_cast<unsigned>((B + BM - 1) / BM));
  const size_t lds = full_sort_lds();
  auto k = full_sort_kernel<VEC, EPI>;
  if (int rc = set_lds(k, lds)) return rc;
  hipLaunchKernelGGL(k, grid, dim3(kBlock), lds, st, U, B, E, N, D, bias, S, ldS, ta);
  return check_launch();
}

static int launch_full_sort(const float* U, int64_t B, const float* E, int64_t N, int64_t D, float* S, int64_t ldS,
                            hipStream_t st) {
  const bool vec = (D % 4 == 0) && aligned16(U) && aligned16(E);
  return vec ? launch_tiled<true, EPI_NONE>(U, B, E, N, D, nullptr, S, ldS, st)
             : launch_tiled<false, EPI_NONE>(U, B, E, N, D, nullptr, S, ldS, st);
}

constexpr int64_t kTopkChunkBytes = 1LL << 30;  // scores workspace per user chunk

static int64_t topk_chunk_rows(int64_t B, int64_t N) {
  int64_t rows = kTopkChunkBytes / (N * static_cast<int64_t>(sizeof(float)));
  if (rows < 1) rows = 1;
  if (rows > B) rows = B;
  return rows;
}

}  // namespace mi_oov

using namespace mi_oov;

extern "C" int mi_oov_full_sort_scores(const float* U, int64_t B, const float* E, int64_t N, int64_t D, float* scores,
                                       void* stream) {
  if (B < 0 || N <= 0 || D <= 0) return MI_OOV_ERR_SHAPE;
  if (B == 0) return MI_OOV_OK;
  if (!U || !E || !scores) return MI_OOV_ERR_NULL;
  if ((B + BM - 1) / BM > 65535) return MI_OOV_ERR_SHAPE;
  return launch_full_sort(U, B, E, N, D, scores, N, static_cast<hipStream_t>(stream));
}

extern "C" int mi_oov_linear_act(const float* X, int64_t B, int64_t K, const float* W, const float* bias,
                                 int64_t N_out, int act, float* Y, void* stream) {
  if (B < 0 || K <= 0 || N_out <= 0) return MI_OOV_ERR_SHAPE;
  if (act < 0 || act > 2) return MI_OOV_ERR_KIND;
  if (B == 0) return MI_OOV_OK;
  if (!X || !W || !bias || !Y) return MI_OOV_ERR_NULL;
  if ((B + BM - 1) / BM > 65535) return MI_OOV_ERR_SHAPE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool vec = (K % 4 == 0) && aligned16(X) && aligned16(W);
#define MI_LIN(V)                                                                                   \
  switch (act) {                                                                                    \
    case 0: return launch_tiled<V, EPI_BIAS>(X, B, W, N_out, K, bias, Y, N_out, st);                \
    case 1: return launch_tiled<V, EPI_BIAS_GELU>(X, B, W, N_out, K, bias, Y, N_out, st);           \
    default: return launch_tiled<V, EPI_BIAS_SIGMOID>(X, B, W, N_out, K, bias, Y, N_out, st);       \
  }
  if (vec) { MI_LIN(true) }
  MI_LIN(false)
#undef MI_LIN
}

// Fused path layout inside the workspace (all regions 256-B aligned).
struct FusedLayout {
  int64_t NT, cap, stride, seg_width, off_tilemax, off_tau, off_tauf, off_u2, off_thr, off_eps, off_e2max, off_ub, off_eb, off_cnt, off_cand, bytes;
};
static int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }
static FusedLayout fused_layout(int64_t B, int64_t N, int64_t k, bool bf16, int kh = 1, bool with_eb = true) {
  FusedLayout L;
  // Pass 1 only needs a LOWER bound of the k-th best score, and the k-th best of any subset of the columns is
  // one: it visits every `stride`-th 128-column block (1/stride of the GEMM work).  The filter pass then lets
  // about k * stride candidates per row through instead of ~2k, so the lists get the full 1024 entries the
  // finalize kernel can rank, and stride is capped so that the expected count stays below half of that; the
  // sampled blocks must still hold at least 2k tiles.  Results are unchanged: every true top-k entry has a
  // score >= tau and the finalize kernel ranks exactly.
  const int64_t nblk = (N + BN - 1) / BN;
  // The bf16 path's passes are cheap next to its per-candidate work: stride 4-5 is its optimum (k = 20, 4096 x 50000:
  // 8 -> 124 us, 4-5 -> 106 us, 3 -> 153 us); the f32 path pays a full-rate GEMM for pass 1 and keeps 8.
  static const int64_t env_stride = env_knob("MI_OOV_TOPK_STRIDE", 0, 0, 64);
  int64_t stride = env_stride > 0 ? env_stride : (bf16 ? (k <= 32 ? 4 : 2) : 8);  // (k = 50: 2 -> 130 us, 3 -> 135 us, 4 -> 151 us)
  if (bf16 && env_stride <= 0 && k < 20) {
    // small k over a big catalogue (the knn search: k = 2, 10 M rows): ~80 candidates per row allow a wider stride, as
    // long as ~200 sampled tiles remain -- the k-th best of a few dozen tile maxima is a poor bound (k = 5, stride 16
    // at 50 000 items left 50 tiles and sent rows to the exact fallback)
    int64_t wide = 80 / k;
    if (wide > 16) wide = 16;
    if (wide > nblk / 96) wide = nblk / 96;
    if (wide > stride) stride = wide;
  }
  if (stride > 256 / k) stride = 256 / k;  // ~k * stride candidates per row: a quarter of the 1024 slots
  if (stride > nblk / k) stride = nblk / k;  // sampled 64-column tiles: 2 * nblk / stride >= 2k
  if (stride < 1) stride = 1;
  L.stride = stride;
  L.NT = (bf16 ? 4 : 2) * ((nblk + stride - 1) / stride);  // (bf16: room for the direct pass 1's 32-column tiles; the LDS-staged one fills half)
  L.cap = kSeg * kSegCap;
  L.seg_width = (nblk + kSeg - 1) / kSeg;
  L.off_tilemax = 0;
  L.off_tau = align256(L.off_tilemax + B * L.NT * 4);
  L.off_tauf = align256(L.off_tau + B * 4);
  L.off_u2 = align256(L.off_tauf + B * 4);
  L.off_thr = align256(L.off_u2 + B * 4);
  L.off_eps = align256(L.off_thr + B * 4);
  L.off_e2max = align256(L.off_eps + B * 4);
  L.off_ub = align256(L.off_e2max + kNormGrid * 4);        // bf16 copies of U and E (64-column inputs only)
  L.off_eb = align256(L.off_ub + B * 128 * kh);  // kh = 64-k halves per row: 1 (D <= 64) or 2 (D <= 128)
  L.off_cnt = align256(L.off_eb + (with_eb ? eb_rows(N) * 128 * kh : 0));  // (a prepared catalogue brings its own bf16 copy of E)
  L.off_cand = align256(L.off_cnt + B * (128 + 1) * 4);  // f32 path: kSeg + 1 counters per row; bf16 path: <= 128 strips + 1
  L.bytes = align256(L.off_cand + B * (L.cap + kOvfCap) * 8);
  return L;
}
// The fused two-pass path needs at least 2k column tiles per row (tau is the k-th tile maximum).
static bool use_fused_topk(int64_t N, int64_t k) { return k <= 256 && (N + 63) / 64 >= 2 * k; }

// the same for a known row width: one k-half of bf16 copies for D <= 64 (the query below keeps room for two)
extern "C" int64_t mi_oov_score_topk_workspace_d(int64_t B, int64_t N, int64_t D, int64_t k) {
  if (B <= 0 || N <= 0 || D <= 0 || k <= 0) return 0;
  if (use_fused_topk(N, k)) {
    const int64_t b = fused_layout(B, N, k, false).bytes;
    if (D > 128) return b;
    const int64_t a = fused_layout(B, N, k, true, D > 64 ? 2 : 1).bytes;
    return a > b ? a : b;
  }
  return topk_chunk_rows(B, N) * N * static_cast<int64_t>(sizeof(float));
}

extern "C" int64_t mi_oov_score_topk_workspace(int64_t B, int64_t N, int64_t k) {
  if (B <= 0 || N <= 0 || k <= 0) return 0;
  if (use_fused_topk(N, k)) {  // the path is chosen per call (D, alignment): room for either
    const int64_t a = fused_layout(B, N, k, true, 2).bytes, b = fused_layout(B, N, k, false).bytes;
    return a > b ? a : b;
  }
  return topk_chunk_rows(B, N) * N * static_cast<int64_t>(sizeof(float));
}

// The bf16 fused path can take per-row exclusions (a bitmap built from the CSR); `mask` is then B x ceil(N / 64) words.
static bool masked_topk_supported(int64_t B, int64_t N, int64_t D, int64_t k) {
  return B > 0 && N > 0 && N < (1LL << 32) && D > 0 && D <= 128 && k > 0 && use_fused_topk(N, k) && (B + BM - 1) / BM <= 65535;
}

// `catalogue`: bf16 copy of E + partial norm maxima made by mi_oov_topk_catalogue_prepare (null: made per call).
static int64_t catalogue_e2_offset(int64_t N, int64_t D) { return align256(eb_rows(N) * 128 * (D > 64 ? 2 : 1)); }
static int64_t catalogue_parts(int64_t N) {
  const int64_t ge = grid_for(N, kBlock / 16);
  return ge > kNormGrid ? kNormGrid : ge;
}

static int score_topk_impl(const float* U, int64_t B, const float* E, int64_t N, int64_t D, int64_t k, int64_t n_skip_low,
                           const int64_t* excl_ptr, const int64_t* excl_cols, unsigned long long* mask,
                           const void* catalogue, float* vals, int64_t* idx, void* workspace, void* stream) {
  if (B < 0 || N <= 0 || D <= 0 || k <= 0 || n_skip_low < 0 || N >= (1LL << 32)) return MI_OOV_ERR_SHAPE;
  if (B == 0) return MI_OOV_OK;
  if (!U || !E || !vals || !idx || !workspace) return MI_OOV_ERR_NULL;
  if ((reinterpret_cast<uintptr_t>(workspace) & 15u) != 0) return MI_OOV_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (use_fused_topk(N, k) && (B + BM - 1) / BM <= 65535) {
    const bool vec = (D % 4 == 0) && aligned16(U) && aligned16(E);
    static const bool bf16_path = env_knob("MI_OOV_TOPK_BF16", 1, 0, 1) != 0;
    // the bf16 path serves rows of up to 64 floats (narrower ones are zero-padded in the bf16 copies; 64-float rows must be 16-byte aligned)
    // the bf16 path serves rows of up to 128 floats (narrower ones are zero-padded in the bf16 copies to 64 or 128 k;
    // rows of exactly 64 / 128 floats are loaded as float4 and must be 16-byte aligned)
    const bool use_bf16 = (bf16_path || mask || catalogue) && D <= 128 && ((D != 64 && D != 128) || vec);
    if ((mask || catalogue) && !use_bf16) return MI_OOV_ERR_ALIGN;  // (the entry points checked the shape; only alignment is left)
    const int kh = (use_bf16 && D > 64) ? 2 : 1;
    const FusedLayout L = fused_layout(B, N, k, use_bf16, kh, catalogue == nullptr);
    char* ws = static_cast<char*>(workspace);
    TopkArgs ta{};
    ta.tilemax = reinterpret_cast<uint32_t*>(ws + L.off_tilemax);
    ta.NT = L.NT;
    ta.tau = reinterpret_cast<uint32_t*>(ws + L.off_tau);
    ta.tauf = reinterpret_cast<float*>(ws + L.off_tauf);
    ta.cnt = reinterpret_cast<int*>(ws + L.off_cnt);
    ta.cand = reinterpret_cast<uint64_t*>(ws + L.off_cand);
    ta.cap = static_cast<int>(L.cap);
    ta.seg_width = static_cast<int>(L.seg_width);
    ta.n_skip_low = n_skip_low;
    ta.col_stride = static_cast<int>(L.stride);
    int rc;
    if (use_bf16) {
      // both GEMM passes on the bf16 matrix cores, exact f32 re-score of the survivors (see bf16_tile_kernel)
      float* u2 = reinterpret_cast<float*>(ws + L.off_u2);
      float* thr = reinterpret_cast<float*>(ws + L.off_thr);
      float* eps = reinterpret_cast<float*>(ws + L.off_eps);
      uint32_t* e2max = reinterpret_cast<uint32_t*>(ws + L.off_e2max);
      __bf16* Ub = reinterpret_cast<__bf16*>(ws + L.off_ub);
      __bf16* Eb = reinterpret_cast<__bf16*>(ws + L.off_eb);
      if (catalogue) {  // prepared once for a catalogue that many user batches are scored against
        Eb = const_cast<__bf16*>(static_cast<const __bf16*>(catalogue));
        e2max = const_cast<uint32_t*>(reinterpret_cast<const uint32_t*>(static_cast<const char*>(catalogue) + catalogue_e2_offset(N, D)));
      }
      const int64_t nblk = (N + BN - 1) / BN;
      const int64_t rb = (B + BM - 1) / BM;
      static const int64_t target = env_knob("MI_OOV_STRIP_WGS", 1024, 1, 65536);
      static const int64_t target1 = env_knob("MI_OOV_STRIP_WGS1", 768, 1, 65536);  // pass 1: 3 workgroups per CU, one round
      auto strips = [&](int64_t nvisit, int64_t tgt, int64_t row_blocks) {  // ~tgt workgroups in all, at most 128 strips, a multiple of 8 when there are 8 blocks
        int64_t n = tgt / row_blocks;
        if (n > 128) n = 128;
        if (n > nvisit) n = nvisit;
        if (n < 1) n = 1;
        if (nvisit >= 8) n = (n + 7) / 8 * 8;
        return n > 128 ? 128 : n;
      };
      const int64_t nvisit1 = (nblk + L.stride - 1) / L.stride;
      static const bool direct_on = env_knob("MI_OOV_FILTER_DIRECT", 1, 0, 1) != 0;  // developer knob: 0 = pass 2 with LDS-staged operands
      const bool direct = direct_on && nblk < (1 << 20);  // (a queue record has 20 bits for the block)
      // direct pass 2: workgroups of 64 user rows (2 tiles of 32: 123 registers, 4 waves per SIMD) and one round of 1024;
      // MI_OOV_FILTER_TILES=4: 128 rows, 211 registers, 2 waves per SIMD, 512 workgroups (48 us instead of 41)
      // Rows per workgroup: 64 (four waves per SIMD) while the bf16 catalogue is served by the L2s -- every strip's blocks are
      // read by all B / 64 row blocks, 13 TB/s of L2 -> CU traffic at 10 M rows --; 128 (half that traffic, two waves per
      // SIMD) once an XCD's share of the copy no longer fits its 4 MB: 4096 x 10 M, k = 2: 6.24 -> 5.33 ms; 4096 x 50 000:
      // 85 -> 92 us, so not there.  MI_OOV_FILTER_TILES (developer knob): 2 or 4 = force, 0 = by size.
      static const int mt_env = static_cast<int>(env_knob("MI_OOV_FILTER_TILES", 0, 0, 4));
      const int mt_auto = eb_rows(N) * 128 / 8 > (4 << 20) ? 4 : 2;
      const int mt = kh == 2 ? 2 : ((B + 63) / 64 > 65535 ? 4 : (mt_env == 4 ? 4 : (mt_env == 2 ? 2 : mt_auto)));  // (grid.y; two k-halves: 64-row workgroups only)
      static const int64_t target_env = env_knob("MI_OOV_STRIP_WGS2", 0, 0, 65536);
      const int64_t target_d = target_env > 0 ? target_env : (mt == 2 ? 1024 : 512);
      const int64_t rbd = (B + mt * 32 - 1) / (mt * 32);
      const int64_t ns1 = strips(nvisit1, target1, rb), ns2 = direct ? strips(nblk, target_d, rbd) : strips(nblk, target, rb);
      // pass 1 in the direct layout (64-row workgroups, 32-column tiles: bf16_tilemax_direct_kernel) wherever the filter
      // pass runs its 64-row form; the large-catalogue searches keep the LDS-staged kernel (MI_OOV_TOPK_DIRECT1=0: everywhere)
      static const bool direct1_on = env_knob("MI_OOV_TOPK_DIRECT1", 1, 0, 1) != 0;
      static const int64_t target_d1 = env_knob("MI_OOV_STRIP_WGS1D", 512, 1, 65536);  // (512 / 1024 / 2048: 26.2 / 28.3 / 35.4 us at D = 128)
      const bool direct1 = direct1_on && direct && mt == 2 && kh == 2;  // (D <= 64: the LDS-staged kernel is as fast -- 16.8 against 18.4 us)
      const int64_t ns1d = strips(nvisit1, target_d1, rbd);
      ta.NT = direct1 ? L.NT : L.NT / 2;
      StripLists sl{};
      sl.ns = static_cast<int>(ns2);
      {  // list capacity: a power of two >= `MI_OOV_LIST_SLACK` (2) x the expected share of ~1.3 k stride candidates per
         // row, within the finalize kernel's slots: the rare longer list continues in the row's overflow list, and a
         // smaller gathered array lets the finalize kernel keep more rows in flight
        static const int64_t slack = env_knob("MI_OOV_LIST_SLACK", 2, 1, 64);
        const int64_t want = slack * 13 * k * L.stride / 10 / ns2 + 1;
        int64_t cap = 8;
        while (cap < want && cap * 2 * ns2 <= kStripSlots) cap *= 2;
        while (cap > 1 && cap * ns2 > kStripSlots) cap /= 2;
        sl.cap = static_cast<int>(cap);
      }
      sl.cnt = ta.cnt;                                  // [B, ns2]
      sl.ovf_cnt = ta.cnt + B * 128;                    // [B]
      sl.cand = ta.cand;                                // [B, ns2 * cap <= kStripSlots]
      sl.ovf = ta.cand + B * kStripSlots;               // [B, kOvfCap]
      if (mask) {
        sl.mask = reinterpret_cast<const uint64_t*>(mask);
        sl.mask_words = (N + 63) / 64;
        if (hipMemsetAsync(mask, 0, static_cast<size_t>(B) * sl.mask_words * 8, st) != hipSuccess) {
          check_launch();
          return MI_OOV_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(mask_build_kernel, dim3(static_cast<unsigned>(B)), dim3(kBlock), 0, st, excl_ptr, excl_cols, B, N, mask, sl.mask_words);
      }
      int64_t gu = grid_for(B, kBlock / 16);
      const int64_t ge = catalogue_parts(N);
      if (gu > kNormGrid) gu = kNormGrid;
      // prepared catalogue + rows of exactly 64 / 128 aligned floats: pass 1 converts the user rows itself (FoldU)
      static const bool fold_on = env_knob("MI_OOV_TOPK_FOLD_U", 1, 0, 1) != 0;  // developer A/B knob
      const bool fold_u = fold_on && catalogue && D == 64 * kh && vec;
      if (!fold_u) {
        if (kh == 2)
          hipLaunchKernelGGL(to_bf16_norm_kernel<2>, dim3(static_cast<unsigned>(gu + (catalogue ? 0 : ge))), dim3(kBlock), 0, st, U, B, Ub, u2,
                             sl.ovf_cnt, static_cast<int>(gu), E, N, Eb, e2max, static_cast<int>(D));
        else
          hipLaunchKernelGGL(to_bf16_norm_kernel<1>, dim3(static_cast<unsigned>(gu + (catalogue ? 0 : ge))), dim3(kBlock), 0, st, U, B, Ub, u2,
                             sl.ovf_cnt, static_cast<int>(gu), E, N, Eb, e2max, static_cast<int>(D));
        if ((rc = check_launch())) return rc;
      }
      const size_t lds_ops = static_cast<size_t>(BM + BN) * BLD * sizeof(__bf16) * kh;
      const size_t lds_filter = lds_ops + BM * (sizeof(float) + sizeof(int)) + 4 * kWaveQueue * 6;
      const FoldU fold{fold_u ? U : nullptr, Ub, u2, sl.ovf_cnt};
      auto launch_tile = [&](auto kern, size_t lds, int64_t ns, const float* thr_arg, int64_t nvisit) -> int {
        if (int rc2 = set_lds(kern, lds)) return rc2;  // (two k-halves: 73.7 KiB of operands)
        hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(ns), static_cast<unsigned>(rb)), dim3(kBlock), lds, st, Ub, B, Eb, N, thr_arg, ta, sl,
                           static_cast<int>(nvisit), fold);
        return MI_OOV_OK;
      };
      if (direct1) {
        const size_t lds1 = static_cast<size_t>(2) * 4 * kh * 64 * 16;
        auto launch1 = [&](auto kern) {
          hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(ns1d), static_cast<unsigned>(rbd)), dim3(kBlock), lds1, st, Ub, B, Eb, N, ta, sl,
                             static_cast<int>(nvisit1), fold);
        };
#define MI_P1(M, K2, F) launch1(bf16_tilemax_direct_kernel<M, K2, F>)
        if (kh == 2) { if (mask) { if (fold_u) MI_P1(true, 2, true); else MI_P1(true, 2, false); } else { if (fold_u) MI_P1(false, 2, true); else MI_P1(false, 2, false); } }
        else { if (mask) { if (fold_u) MI_P1(true, 1, true); else MI_P1(true, 1, false); } else { if (fold_u) MI_P1(false, 1, true); else MI_P1(false, 1, false); } }
#undef MI_P1
        rc = check_launch();
      } else if (fold_u) {
        if (kh == 2) rc = mask ? launch_tile(bf16_tile_kernel<EPI_TILEMAX, true, 2, true>, lds_ops, ns1, nullptr, nvisit1)
                               : launch_tile(bf16_tile_kernel<EPI_TILEMAX, false, 2, true>, lds_ops, ns1, nullptr, nvisit1);
        else rc = mask ? launch_tile(bf16_tile_kernel<EPI_TILEMAX, true, 1, true>, lds_ops, ns1, nullptr, nvisit1)
                       : launch_tile(bf16_tile_kernel<EPI_TILEMAX, false, 1, true>, lds_ops, ns1, nullptr, nvisit1);
      } else if (kh == 2) rc = mask ? launch_tile(bf16_tile_kernel<EPI_TILEMAX, true, 2>, lds_ops, ns1, nullptr, nvisit1)
                                    : launch_tile(bf16_tile_kernel<EPI_TILEMAX, false, 2>, lds_ops, ns1, nullptr, nvisit1);
      else rc = mask ? launch_tile(bf16_tile_kernel<EPI_TILEMAX, true, 1>, lds_ops, ns1, nullptr, nvisit1)
                     : launch_tile(bf16_tile_kernel<EPI_TILEMAX, false, 1>, lds_ops, ns1, nullptr, nvisit1);
      if (rc) return rc;
      Bf16Bound bb{u2, e2max, static_cast<int>(ge), thr, eps};
      if (direct1 && ta.NT <= 512)  // (the direct pass 1 leaves 32-column tiles: pairs of them are the 64-column tiles' keys)
        hipLaunchKernelGGL((tile_kth_wave_kernel<4, true>), dim3(static_cast<unsigned>((B + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0, st, ta.tilemax, B,
                           static_cast<int>(ta.NT / 2), static_cast<int>(k), reinterpret_cast<uint32_t*>(ws + L.off_tau),
                           reinterpret_cast<float*>(ws + L.off_tauf), bb);
      else if (ta.NT <= 256)
        hipLaunchKernelGGL(tile_kth_wave_kernel<4>, dim3(static_cast<unsigned>((B + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0, st, ta.tilemax, B,
                           static_cast<int>(ta.NT), static_cast<int>(k), reinterpret_cast<uint32_t*>(ws + L.off_tau),
                           reinterpret_cast<float*>(ws + L.off_tauf), bb);
      else if (ta.NT <= 64 * kKthRegs)
        hipLaunchKernelGGL(tile_kth_wave_kernel<kKthRegs>, dim3(static_cast<unsigned>((B + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0, st, ta.tilemax, B,
                           static_cast<int>(ta.NT), static_cast<int>(k), reinterpret_cast<uint32_t*>(ws + L.off_tau),
                           reinterpret_cast<float*>(ws + L.off_tauf), bb);
      else
        hipLaunchKernelGGL(tile_kth_kernel, dim3(static_cast<unsigned>(B)), dim3(kBlock), 0, st, ta.tilemax, B, ta.NT,
                           static_cast<int>(k), reinterpret_cast<uint32_t*>(ws + L.off_tau), reinterpret_cast<float*>(ws + L.off_tauf), bb);
      if ((rc = check_launch())) return rc;
      // row counters, the four waves' queues, and (MI_FD_ALDS) the user fragments: [32-row tiles][k-steps][64 lanes] x 16 B
      const size_t lds_direct = BM * sizeof(int) + 4 * (kDirectQueue + 1) * 16 + ((MI_FD_ALDS && (kh == 2 || mt == 2)) ? static_cast<size_t>(2) * 4 * kh * 64 * 16 : 0);
      auto launch_direct = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(ns2), static_cast<unsigned>(rbd)), dim3(kBlock), lds_direct, st, Ub, B, Eb, N, thr, ta, sl,
                           static_cast<int>(nblk));
      };
      // expected passing scores per wave and block: ~1.3 k stride candidates (+ the 2 eps slack) per row of N scores, 64 rows
      // x 32 columns per wave and block; below a quarter the whole-lane test comes first (RARE)
      static const int64_t rare_env = env_knob("MI_OOV_FILTER_RARE", -1, -1, 1);  // developer knob: force off / on
      const double lam = 2048.0 * 2.0 * 1.3 * static_cast<double>(k) * static_cast<double>(L.stride) / static_cast<double>(N);
      const bool rare = rare_env >= 0 ? rare_env == 1 : lam < 0.25;
      if (direct && kh == 2) {  // 128-k rows: the user fragments alone are 64 registers per 64 rows -- workgroups of 64 rows
        if (mask) launch_direct(bf16_filter_direct_kernel<true, 2, 2>);
        else if (rare) launch_direct(bf16_filter_direct_kernel<false, 2, 2, true>);
        else launch_direct(bf16_filter_direct_kernel<false, 2, 2>);
      } else if (direct && mask) {
        if (mt == 2) launch_direct(bf16_filter_direct_kernel<true, 2>);
        else launch_direct(bf16_filter_direct_kernel<true, 4>);
      } else if (direct) {
        if (mt == 2 && rare) launch_direct(bf16_filter_direct_kernel<false, 2, 1, true>);
        else if (mt == 2) launch_direct(bf16_filter_direct_kernel<false, 2>);
        else if (rare) launch_direct(bf16_filter_direct_kernel<false, 4, 1, true>);
        else launch_direct(bf16_filter_direct_kernel<false, 4>);
      } else if (kh == 2) {
        rc = mask ? launch_tile(bf16_tile_kernel<EPI_FILTER, true, 2>, lds_filter, ns2, thr, nblk)
                  : launch_tile(bf16_tile_kernel<EPI_FILTER, false, 2>, lds_filter, ns2, thr, nblk);
        if (rc) return rc;
      } else {
        rc = mask ? launch_tile(bf16_tile_kernel<EPI_FILTER, true, 1>, lds_filter, ns2, thr, nblk)
                  : launch_tile(bf16_tile_kernel<EPI_FILTER, false, 1>, lds_filter, ns2, thr, nblk);
        if (rc) return rc;
      }
      if ((rc = check_launch())) return rc;
      hipLaunchKernelGGL(topk_finalize_exact_kernel, dim3(static_cast<unsigned>((B + kFinRows - 1) / kFinRows)), dim3(kBlock),
                         static_cast<size_t>(kFinRows) * (sl.ns * sl.cap + kOvfLds) * 8, st, U, E, B, N, D,
                         static_cast<int>(k), n_skip_low, sl, eps, vals, idx);
      return check_launch();
    }
    if (hipMemsetAsync(ta.cnt, 0, static_cast<size_t>(B) * (kSeg + 1) * 4, st) != hipSuccess) {
      check_launch();
      return MI_OOV_ERR_LAUNCH;
    }
    rc = vec ? launch_tiled<true, EPI_TILEMAX>(U, B, E, N, D, nullptr, nullptr, 0, st, ta)
             : launch_tiled<false, EPI_TILEMAX>(U, B, E, N, D, nullptr, nullptr, 0, st, ta);
    if (rc) return rc;
    hipLaunchKernelGGL(tile_kth_kernel, dim3(static_cast<unsigned>(B)), dim3(kBlock), 0, st, ta.tilemax, B, L.NT,
                       static_cast<int>(k), reinterpret_cast<uint32_t*>(ws + L.off_tau), reinterpret_cast<float*>(ws + L.off_tauf), Bf16Bound{});
    if ((rc = check_launch())) return rc;
    rc = vec ? launch_tiled<true, EPI_FILTER>(U, B, E, N, D, nullptr, nullptr, 0, st, ta)
             : launch_tiled<false, EPI_FILTER>(U, B, E, N, D, nullptr, nullptr, 0, st, ta);
    if (rc) return rc;
    hipLaunchKernelGGL(topk_finalize_kernel, dim3(static_cast<unsigned>(B)), dim3(kBlock), 0, st, U, E, B, N, D,
                       static_cast<int>(k), n_skip_low, ta.cnt, ta.cand, vals, idx);
    return check_launch();
  }
  float* S = static_cast<float*>(workspace);
  const int64_t chunk = topk_chunk_rows(B, N);
  for (int64_t b0 = 0; b0 < B; b0 += chunk) {
    const int64_t rows = (b0 + chunk <= B) ? chunk : (B - b0);
    if (int rc = launch_full_sort(U + b0 * D, rows, E, N, D, S, N, st)) return rc;
    if (k <= 256)
      hipLaunchKernelGGL(topk_select_kernel, dim3(static_cast<unsigned>(rows)), dim3(kBlock), 0, st, S, rows, N, N,
                         static_cast<int>(k), n_skip_low, vals + b0 * k, idx + b0 * k);
    else
      hipLaunchKernelGGL(topk_rows_kernel, dim3(static_cast<unsigned>(rows)), dim3(kBlock), 0, st, S, rows, N, N, k,
                         n_skip_low, vals + b0 * k, idx + b0 * k);
    if (int rc = check_launch()) return rc;
  }
  return MI_OOV_OK;
}

extern "C" int mi_oov_score_topk(const float* U, int64_t B, const float* E, int64_t N, int64_t D, int64_t k,
                                 int64_t n_skip_low, float* vals, int64_t* idx, void* workspace, void* stream) {
  return score_topk_impl(U, B, E, N, D, k, n_skip_low, nullptr, nullptr, nullptr, nullptr, vals, idx, workspace, stream);
}

// Exclusions of any length on the fused path: the histories become a bitmap (B x ceil(N / 64) words) that pass 1 applies
// to the tile maxima (so tau is the k-th best ALLOWED tile maximum and the candidate count does not grow with the
// histories), pass 2 applies when it emits a candidate, and the exact fallback applies to its keys.
extern "C" int64_t mi_oov_score_topk_masked_workspace(int64_t B, int64_t N, int64_t D, int64_t k) {
  if (!masked_topk_supported(B, N, D, k)) return 0;
  return align256(mi_oov_score_topk_workspace(B, N, k)) + align256(B * ((N + 63) / 64) * 8);
}

extern "C" int mi_oov_score_topk_masked(const float* U, int64_t B, const float* E, int64_t N, int64_t D, int64_t k,
                                        int64_t n_skip_low, const int64_t* excl_ptr, const int64_t* excl_cols, float* vals,
                                        int64_t* idx, void* workspace, void* stream) {
  if (B == 0 && N > 0 && D > 0 && k > 0) return MI_OOV_OK;
  if (!masked_topk_supported(B, N, D, k) || n_skip_low < 0) return MI_OOV_ERR_SHAPE;
  if (!U || !E || !excl_ptr || !excl_cols || !vals || !idx || !workspace) return MI_OOV_ERR_NULL;
  if ((reinterpret_cast<uintptr_t>(workspace) & 15u) != 0) return MI_OOV_ERR_ALIGN;
  unsigned long long* mask = reinterpret_cast<unsigned long long*>(static_cast<char*>(workspace) + align256(mi_oov_score_topk_workspace(B, N, k)));
  return score_topk_impl(U, B, E, N, D, k, n_skip_low, excl_ptr, excl_cols, mask, nullptr, vals, idx, workspace, stream);
}

// A catalogue that many user batches are scored against (the knn search's feature table, the item table of an evaluation
// run) is converted once: the counterpart of the reference building its ScaNN searcher at construction
// (knn_embedder.py:84-93).  The buffer holds the bf16 copy of E and the per-workgroup maxima of its squared row norms.
extern "C" int64_t mi_oov_topk_catalogue_bytes(int64_t N, int64_t D) {
  if (N <= 0 || N >= (1LL << 32) || D <= 0 || D > 128) return 0;
  return catalogue_e2_offset(N, D) + align256(kNormGrid * 4);
}

extern "C" int mi_oov_topk_catalogue_prepare(const float* E, int64_t N, int64_t D, void* catalogue, void* stream) {
  if (N <= 0 || N >= (1LL << 32) || D <= 0 || D > 128) return MI_OOV_ERR_SHAPE;
  if (!E || !catalogue) return MI_OOV_ERR_NULL;
  if (((D == 64 || D == 128) && !aligned16(E)) || (reinterpret_cast<uintptr_t>(catalogue) & 15u) != 0) return MI_OOV_ERR_ALIGN;
  __bf16* Eb = static_cast<__bf16*>(catalogue);
  uint32_t* e2part = reinterpret_cast<uint32_t*>(static_cast<char*>(catalogue) + catalogue_e2_offset(N, D));
  if (D > 64)
    hipLaunchKernelGGL(to_bf16_norm_kernel<2>, dim3(static_cast<unsigned>(catalogue_parts(N))), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(nullptr), 0, static_cast<__bf16*>(nullptr), static_cast<float*>(nullptr),
                       static_cast<int*>(nullptr), 0, E, N, Eb, e2part, static_cast<int>(D));
  else
    hipLaunchKernelGGL(to_bf16_norm_kernel<1>, dim3(static_cast<unsigned>(catalogue_parts(N))), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(nullptr), 0, static_cast<__bf16*>(nullptr), static_cast<float*>(nullptr),
                       static_cast<int*>(nullptr), 0, E, N, Eb, e2part, static_cast<int>(D));
  return check_launch();
}

// Workspace of a call against a prepared catalogue: the lists and the user rows' copies for THIS row width, no room for a
// bf16 copy of E (the catalogue holds it) -- the general query keeps room for either path and two k-halves, which at
// 10 M rows is 2.56 GB of copy and made the host cut a 4096-user batch into 32 chunks (round 3's D <= 128 change:
// 16.4 ms instead of 6.4 for the knn search of BASELINE's table).  The mask, when there is one, sits behind it.
static int64_t prepared_lists_bytes(int64_t B, int64_t N, int64_t D, int64_t k) {
  return align256(fused_layout(B, N, k, true, D > 64 ? 2 : 1, false).bytes);
}

// mi_oov_score_topk (excl_ptr == NULL) or mi_oov_score_topk_masked (excl_ptr != NULL) against a prepared catalogue of
// the same E; workspace: mi_oov_score_topk_prepared_workspace (the respective call's, larger, is fine too).  Shapes the bf16 path does not take are an error here (a caller
// that prepared a catalogue has a 64-column, aligned E).
extern "C" int mi_oov_score_topk_prepared(const float* U, int64_t B, const float* E, int64_t N, int64_t D, int64_t k,
                                          int64_t n_skip_low, const int64_t* excl_ptr, const int64_t* excl_cols,
                                          const void* catalogue, float* vals, int64_t* idx, void* workspace, void* stream) {
  if (B == 0 && N > 0 && D > 0 && k > 0) return MI_OOV_OK;
  if (!masked_topk_supported(B, N, D, k) || n_skip_low < 0) return MI_OOV_ERR_SHAPE;
  if (!U || !E || !catalogue || !vals || !idx || !workspace || (excl_ptr && !excl_cols)) return MI_OOV_ERR_NULL;
  if ((reinterpret_cast<uintptr_t>(workspace) & 15u) != 0 || (reinterpret_cast<uintptr_t>(catalogue) & 15u) != 0 || !aligned16(U) || !aligned16(E))
    return MI_OOV_ERR_ALIGN;
  unsigned long long* mask = nullptr;
  if (excl_ptr) mask = reinterpret_cast<unsigned long long*>(static_cast<char*>(workspace) + prepared_lists_bytes(B, N, D, k));
  return score_topk_impl(U, B, E, N, D, k, n_skip_low, excl_ptr, excl_cols, mask, catalogue, vals, idx, workspace, stream);
}

extern "C" int64_t mi_oov_score_topk_prepared_workspace(int64_t B, int64_t N, int64_t D, int64_t k, int masked) {
  if (!masked_topk_supported(B, N, D, k)) return 0;
  return prepared_lists_bytes(B, N, D, k) + (masked ? align256(B * ((N + 63) / 64) * 8) : 0);
}

// ---- full-sort evaluation with per-user exclusions (history masks) ---------------------------------------------
// InductiveEvaluator.eval_batch (R/inductive/evaluator.py:70-96) sets scores[:, 0] and scores[history_index] to -inf
// before the collector's topk.  Here: top-(k + h_max) through mi_oov_score_topk (fused, nothing materialised), then
// one pass per row that walks the ranked candidates and keeps the first k whose column is not in the row's sorted
// exclusion list -- at most h_max of the k + h_max best can be excluded, so the first k survivors are the answer.
__global__ __launch_bounds__(kBlock) void topk_exclude_kernel(const float* __restrict__ vals2, const int64_t* __restrict__ idx2,
                                                              int64_t B, int k2, int k, const int64_t* __restrict__ excl_ptr,
                                                              const int64_t* __restrict__ excl_cols,
                                                              float* __restrict__ vals, int64_t* __restrict__ idx) {
  __shared__ int wave_tot[kBlock / 64];
  const int64_t row = blockIdx.x;
  if (row >= B) return;
  const int64_t e0 = excl_ptr[row], e1 = excl_ptr[row + 1];
  const int t = threadIdx.x;  // k2 <= 256 = kBlock: one candidate per thread
  bool keep = false;
  int64_t c = -1;
  float v = -__builtin_inff();
  if (t < k2) {
    c = idx2[row * k2 + t];
    v = vals2[row * k2 + t];
    keep = c >= 0;
    int64_t lo = e0, hi = e1;  // binary search in the sorted exclusion list
    while (keep && lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      const int64_t m = excl_cols[mid];
      if (m == c) keep = false;
      else if (m < c) lo = mid + 1;
      else hi = mid;
    }
  }
  int total;
  const int pos = block_excl_scan(keep ? 1 : 0, wave_tot, total);
  if (keep && pos < k) {
    vals[row * k + pos] = v;
    idx[row * k + pos] = c;
  }
  for (int j = total + t; j < k; j += kBlock) {  // fewer than k admissible columns
    vals[row * k + j] = -__builtin_inff();
    idx[row * k + j] = -1;
  }
}

extern "C" int64_t mi_oov_score_topk_excl_workspace(int64_t B, int64_t N, int64_t k, int64_t h_max) {
  if (B <= 0 || N <= 0 || k <= 0 || h_max < 0 || k + h_max > 256) return 0;
  const int64_t k2 = k + h_max;
  return align256(mi_oov_score_topk_workspace(B, N, k2)) + align256(B * k2 * 4) + align256(B * k2 * 8);
}

extern "C" int mi_oov_score_topk_excl(const float* U, int64_t B, const float* E, int64_t N, int64_t D, int64_t k,
                                      int64_t n_skip_low, const int64_t* excl_ptr, const int64_t* excl_cols, int64_t h_max,
                                      float* vals, int64_t* idx, void* workspace, void* stream) {
  if (k <= 0 || h_max < 0 || k + h_max > 256) return MI_OOV_ERR_SHAPE;
  if (B == 0) return MI_OOV_OK;
  if (!excl_ptr || !excl_cols || !workspace || !vals || !idx) return MI_OOV_ERR_NULL;
  const int64_t k2 = k + h_max;
  char* ws = static_cast<char*>(workspace);
  const int64_t off_v = align256(mi_oov_score_topk_workspace(B, N, k2));
  float* vals2 = reinterpret_cast<float*>(ws + off_v);
  int64_t* idx2 = reinterpret_cast<int64_t*>(ws + off_v + align256(B * k2 * 4));
  if (int rc = mi_oov_score_topk(U, B, E, N, D, k2, n_skip_low, vals2, idx2, workspace, stream)) return rc;
  hipLaunchKernelGGL(topk_exclude_kernel, dim3(static_cast<unsigned>(B)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), vals2,
                     idx2, B, static_cast<int>(k2), static_cast<int>(k), excl_ptr, excl_cols, vals, idx);
  return check_launch();
}

// ---- the route of last resort: any D, any k, any history length, any catalogue size -----------------------------------------
// Scores of a chunk of users materialised (<= 1 GiB), the chunk's exclusion bitmap beside them, exact selection that skips
// excluded columns.  What ops.score_topk_excl used to do with torch (materialise, a Python loop over rows, torch.topk).
extern "C" int64_t mi_oov_score_topk_excl_dense_workspace(int64_t B, int64_t N) {
  if (B <= 0 || N <= 0) return 0;
  const int64_t rows = topk_chunk_rows(B, N);
  return align256(rows * N * static_cast<int64_t>(sizeof(float))) + align256(rows * ((N + 63) / 64) * 8);
}

extern "C" int mi_oov_score_topk_excl_dense(const float* U, int64_t B, const float* E, int64_t N, int64_t D, int64_t k,
                                            int64_t n_skip_low, const int64_t* excl_ptr, const int64_t* excl_cols,
                                            float* vals, int64_t* idx, void* workspace, void* stream) {
  if (B < 0 || N <= 0 || D <= 0 || k <= 0 || n_skip_low < 0 || N >= (1LL << 32)) return MI_OOV_ERR_SHAPE;
  if (B == 0) return MI_OOV_OK;
  if (!U || !E || !excl_ptr || !excl_cols || !vals || !idx || !workspace) return MI_OOV_ERR_NULL;
  if ((reinterpret_cast<uintptr_t>(workspace) & 15u) != 0) return MI_OOV_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t chunk = topk_chunk_rows(B, N), words = (N + 63) / 64;
  float* S = static_cast<float*>(workspace);
  unsigned long long* mask = reinterpret_cast<unsigned long long*>(static_cast<char*>(workspace) + align256(chunk * N * static_cast<int64_t>(sizeof(float))));
  for (int64_t b0 = 0; b0 < B; b0 += chunk) {
    const int64_t rows = (b0 + chunk <= B) ? chunk : (B - b0);
    if (int rc = launch_full_sort(U + b0 * D, rows, E, N, D, S, N, st)) return rc;
    if (hipMemsetAsync(mask, 0, static_cast<size_t>(rows) * words * 8, st) != hipSuccess) {
      check_launch();
      return MI_OOV_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(mask_build_kernel, dim3(static_cast<unsigned>(rows)), dim3(kBlock), 0, st, excl_ptr + b0, excl_cols, rows, N, mask, words);
    if (k <= 256)
      hipLaunchKernelGGL(topk_select_masked_kernel, dim3(static_cast<unsigned>(rows)), dim3(kBlock), 0, st, S, rows, N, N, static_cast<int>(k),
                         n_skip_low, reinterpret_cast<const uint64_t*>(mask), words, vals + b0 * k, idx + b0 * k);
    else
      hipLaunchKernelGGL(topk_rows_kernel, dim3(static_cast<unsigned>(rows)), dim3(kBlock), 0, st, S, rows, N, N, k, n_skip_low, vals + b0 * k,
                         idx + b0 * k, reinterpret_cast<const uint64_t*>(mask), words);
    if (int rc = check_launch()) return rc;
  }
  return MI_OOV_OK;
}
