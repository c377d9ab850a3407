// lsh / slsh: fused gather -> sign-random-projection -> bucket aggregate (+ optional score,
// + optional in-vocabulary splice).  HBM-bound: per lookup one random F-float row read and one
// D-float row write; planes and bucket rows live in LDS for the whole workgroup.
//
// Work decomposition (wave64, gfx950):
//   * one lookup is owned by a 16-lane DPP row ("group"); a wave owns 4 groups and keeps R
//     lookups per group in flight, i.e. 4*R independent 16-B loads per lane are issued before
//     the first use, which is what hides the ~900-cycle HBM miss latency;
//   * lane l of a group holds elements e with (e/4)%16 == l of the feature row (one
//     global_load_dwordx4 per 64 floats), the matching slice of every plane comes from LDS
//     (ds_read_b128, identical address in the 4 groups -> broadcast), the 16 partials are
//     reduced with 4 DPP adds (no LDS, no ds_bpermute);
//   * every lane of the group then knows all H bits and accumulates its 4-float slice of the
//     output row from the LDS-resident bucket rows; the row is stored as 16 x 16 B = one
//     contiguous 256-B segment per group at D = 64.
//
// Reference functions restated: R/inductive/torch_hash.py:55-60, lsh_embedder.py:116-179,
// single_lsh_embedder.py:82-109, bpr.py:48-125,145-149 (see include/mi_oov.h).
#include <stdlib.h>

#include "common.hpp"

namespace mi_oov {

struct LshParams {
  const int64_t* ids;
  int64_t B;
  const float* feat;
  int64_t N, F;
  const float* planes;
  int64_t H;
  const float* buckets;  // lsh: [H,D]   slsh: [n_buckets,D]
  int64_t n_buckets;
  int64_t D;
  const float* table;  // lookup mode: [n_vocab,D] or nullptr
  int64_t n_vocab;
  const float* other;  // score mode: [B,D] or nullptr
  float* score;
  float* out;
  uint8_t* bits;
  int64_t* idx;  // slsh only
  int h_chunk;   // lsh_fused_kernel<..., CHUNK>: planes (and bucket rows) resident in LDS at a time
  int64_t d0, Dw;  // lsh: the window of output columns [d0, d0 + Dw) this launch writes (rows wider than 256 floats take
                   // one launch per 256 columns; D stays the row stride of buckets / table / out)
};

// Stage a [rows, L] matrix (row stride ld in memory) into LDS with row stride LP (zero padded).
__device__ __forceinline__ void stage_padded(float* dst, const float* src, int64_t rows, int64_t L, int LP, int64_t ld) {
  const int64_t total = rows * LP;
  for (int64_t i = threadIdx.x; i < total; i += kBlock) {
    const int64_t r = i / LP;
    const int e = static_cast<int>(i - r * LP);
    dst[i] = (e < L) ? src[r * ld + e] : 0.f;
  }
}
__device__ __forceinline__ void stage_padded(float* dst, const float* src, int64_t rows, int64_t L, int LP) {
  stage_padded(dst, src, rows, L, LP, L);
}

// ------------------------------------------------------------------------------------------
// lsh: F <= 256 (FC chunks of 64 floats in registers), D <= 256 (DC chunks)
// ------------------------------------------------------------------------------------------
// CHUNK (round 4): a model with hundreds or thousands of OOV buckets has as many hyperplanes (lsh_embedder.py:108-114), and
// H x (F + D) floats no longer fit the LDS (H = 1000 at F = D = 64: 512 KB).  The planes and bucket rows are then staged
// h_chunk at a time: the gathered rows, the bit count and the bucket-row chain acc = fma(bit_h, W[h], acc) stay in
// registers across the chunks, h still runs 0 .. H-1 in order, so the results are those of the unchunked kernel bit for
// bit.  The workgroup's four waves walk their tiles in lock step (two barriers per chunk): a wave past the last tile
// works on clamped rows and stores nothing.
template <int FC, int DC, bool VEC, int R, bool CHUNK = false>
__global__ __launch_bounds__(kBlock) void lsh_fused_kernel(LshParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int FP = FC * 64, DP = DC * 64;
  const int lane = threadIdx.x & 63;
  const int l16 = lane & 15;
  const int grp = lane >> 4;
  const int wv = threadIdx.x >> 6;
  const int64_t ntiles = (p.B + 4 * R - 1) / (4 * R);
  const int64_t tile0 = static_cast<int64_t>(blockIdx.x) * 4 + wv;
  // ids of the first tile before the weights are staged (as in lsh64_kernel): the first hop overlaps the staging
  int64_t idn[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t row = tile0 * (4 * R) + r * 4 + grp;
    idn[r] = p.ids[row < p.B ? row : p.B - 1];
  }
  const int H = static_cast<int>(p.H);
  const int HC = CHUNK ? p.h_chunk : H;  // planes resident at a time
  float* sP = smem;
  float* sW = smem + HC * FP;
  if constexpr (!CHUNK) {
    stage_padded(sP, p.planes, p.H, p.F, FP);
    if (p.buckets) stage_padded(sW, p.buckets + p.d0, p.H, p.Dw, DP, p.D);
    __syncthreads();
  }
  const bool want_emb = (p.out != nullptr) || (p.score != nullptr);

  // CHUNK: the loop bound is the workgroup's (wave 0's) tile, so that all four waves meet at the barriers
  for (int64_t tile = tile0; (CHUNK ? tile - wv : tile) < ntiles; tile += static_cast<int64_t>(gridDim.x) * 4) {
    int64_t row[R];
    int64_t id[R];
    bool live[R], valid[R], oov[R];
    float4 x[R][FC];
    // ids first, then every row gather back-to-back on clamped (always valid) addresses: no wait
    // and no branch sits between two gathers, so a tile costs ONE HBM round trip, not R.
#pragma unroll
    for (int r = 0; r < R; ++r) {
      row[r] = tile * (4 * R) + r * 4 + grp;
      live[r] = row[r] < p.B;
      id[r] = (tile == tile0) ? idn[r] : p.ids[live[r] ? row[r] : p.B - 1];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      oov[r] = (p.table == nullptr) || (id[r] >= p.n_vocab);
      valid[r] = live[r] && (oov[r] ? (static_cast<uint64_t>(id[r]) < static_cast<uint64_t>(p.N))
                                    : (id[r] >= 0));
      const bool ld = valid[r] && oov[r];
      const float* frow = p.feat + (ld ? id[r] : 0) * p.F;
#pragma unroll
      for (int c = 0; c < FC; ++c) {
        const int e = (c * 16 + l16) * 4;
        x[r][c] = load4<VEC>(frow, e, p.F);
      }
    }

    float4 acc[R][DC];
    float cnt[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      cnt[r] = 0.f;
#pragma unroll
      for (int c = 0; c < DC; ++c) acc[r][c] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    for (int h0 = 0; h0 < H; h0 += HC) {
    const int hn = (H - h0 < HC) ? H - h0 : HC;
    if constexpr (CHUNK) {
      __syncthreads();  // every wave is done with the previous chunk (or tile)
      stage_padded(sP, p.planes + static_cast<int64_t>(h0) * p.F, hn, p.F, FP);
      if (p.buckets) stage_padded(sW, p.buckets + static_cast<int64_t>(h0) * p.D + p.d0, hn, p.Dw, DP, p.D);
      __syncthreads();
    }
    for (int hl = 0; hl < hn; ++hl) {
      const int h = h0 + hl;
      float4 pw[FC];
#pragma unroll
      for (int c = 0; c < FC; ++c)
        pw[c] = *reinterpret_cast<const float4*>(sP + hl * FP + (c * 16 + l16) * 4);
      float4 bw[DC];
      if (want_emb) {
#pragma unroll
        for (int c = 0; c < DC; ++c)
          bw[c] = *reinterpret_cast<const float4*>(sW + hl * DP + (c * 16 + l16) * 4);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float part = 0.f;
#pragma unroll
        for (int c = 0; c < FC; ++c) part = dot4_fma(x[r][c], pw[c], part);
        const float s = row16_sum(part);
        const float bit = (s < 0.f) ? 0.f : 1.f;  // >=0, +-0 and NaN -> 1 (torch_hash.py:57-59)
        cnt[r] = cnt[r] + bit;
        if (want_emb) {
#pragma unroll
          for (int c = 0; c < DC; ++c) {
            acc[r][c].x = __builtin_fmaf(bit, bw[c].x, acc[r][c].x);
            acc[r][c].y = __builtin_fmaf(bit, bw[c].y, acc[r][c].y);
            acc[r][c].z = __builtin_fmaf(bit, bw[c].z, acc[r][c].z);
            acc[r][c].w = __builtin_fmaf(bit, bw[c].w, acc[r][c].w);
          }
        }
        if (p.bits && l16 == 0 && live[r])
          p.bits[row[r] * p.H + h] = valid[r] ? static_cast<uint8_t>(bit) : static_cast<uint8_t>(0xFF);
      }
    }
    }  // chunks

    if (want_emb) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float4 emb[DC];
#pragma unroll
        for (int c = 0; c < DC; ++c) {
          const int e = (c * 16 + l16) * 4;
          if (oov[r]) {
            emb[c].x = acc[r][c].x / cnt[r];  // 0/0 -> NaN row, as lsh_embedder.py:178
            emb[c].y = acc[r][c].y / cnt[r];
            emb[c].z = acc[r][c].z / cnt[r];
            emb[c].w = acc[r][c].w / cnt[r];
          } else {
            emb[c] = valid[r] ? load4<VEC>(p.table + id[r] * p.D + p.d0, e, p.Dw) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
          if (!valid[r]) emb[c] = make_float4(qnan(), qnan(), qnan(), qnan());
          if (p.out && live[r]) store4<VEC>(p.out + row[r] * p.D + p.d0, e, p.Dw, emb[c]);
        }
        if (p.score) {
          float sp = 0.f;
          const float* orow = p.other + (live[r] ? row[r] : 0) * p.D;
#pragma unroll
          for (int c = 0; c < DC; ++c) {
            const int e = (c * 16 + l16) * 4;
            float4 o = live[r] ? load4<VEC>(orow, e, p.D) : make_float4(0.f, 0.f, 0.f, 0.f);
            // padded tail lanes hold emb = x/cnt of zero-padded acc: force exact zeros there
            float4 m = emb[c];
            if (e + 0 >= p.D) m.x = 0.f;
            if (e + 1 >= p.D) m.y = 0.f;
            if (e + 2 >= p.D) m.z = 0.f;
            if (e + 3 >= p.D) m.w = 0.f;
            sp = dot4_muladd(o, m, sp);
          }
          const float s = row16_sum(sp);
          if (l16 == 0 && live[r]) p.score[row[r]] = s;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// lsh, wide feature rows (F > 256): chunks streamed per block of 8 planes, the row is re-read
// from L1/L2 for every plane block.  Same canonical order (lane chain over increasing chunk).
// ------------------------------------------------------------------------------------------
// CHUNK: as lsh_fused_kernel<..., CHUNK> -- h_chunk (a multiple of 8) planes and bucket rows in LDS at a time, the four
// waves in lock step (item features from a text or image encoder are hundreds of floats wide, and the model may have a
// thousand OOV buckets: 768 x 1000 planes are 3 MB).
template <int DC, bool VEC, bool CHUNK = false>
__global__ __launch_bounds__(kBlock) void lsh_wide_kernel(LshParams p, int FP) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int DP = DC * 64;
  constexpr int HB = 8;
  const int H = static_cast<int>(p.H);
  const int HC = CHUNK ? p.h_chunk : H;
  float* sP = smem;
  float* sW = smem + HC * FP;
  if constexpr (!CHUNK) {
    stage_padded(sP, p.planes, p.H, p.F, FP);
    if (p.buckets) stage_padded(sW, p.buckets + p.d0, p.H, p.Dw, DP, p.D);
    __syncthreads();
  }

  const int lane = threadIdx.x & 63;
  const int l16 = lane & 15;
  const int grp = lane >> 4;
  const int wv = threadIdx.x >> 6;
  const int64_t ntiles = (p.B + 3) / 4;
  const int nchunk = FP / 64;
  const bool want_emb = (p.out != nullptr) || (p.score != nullptr);

  for (int64_t tile = static_cast<int64_t>(blockIdx.x) * 4 + wv; (CHUNK ? tile - wv : tile) < ntiles;
       tile += static_cast<int64_t>(gridDim.x) * 4) {
    const int64_t row = tile * 4 + grp;
    const bool live = row < p.B;
    const int64_t id = live ? p.ids[row] : -1;
    const bool oov = (p.table == nullptr) || (id >= p.n_vocab);
    const bool valid = live && (oov ? (static_cast<uint64_t>(id) < static_cast<uint64_t>(p.N)) : (id >= 0));
    const bool ld = valid && oov;
    const float* frow = p.feat + (ld ? id : 0) * p.F;

    float4 acc[DC];
    float cnt = 0.f;
#pragma unroll
    for (int c = 0; c < DC; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int h0 = 0; h0 < H; h0 += HB) {
      if constexpr (CHUNK) {
        if (h0 % HC == 0) {  // (uniform) the next h_chunk planes and bucket rows
          const int hn = (H - h0 < HC) ? H - h0 : HC;
          __syncthreads();
          stage_padded(sP, p.planes + static_cast<int64_t>(h0) * p.F, hn, p.F, FP);
          if (p.buckets) stage_padded(sW, p.buckets + static_cast<int64_t>(h0) * p.D + p.d0, hn, p.Dw, DP, p.D);
          __syncthreads();
        }
      }
      const int hb = CHUNK ? h0 % HC : h0;  // LDS row of plane h0
      float part[HB];
#pragma unroll
      for (int j = 0; j < HB; ++j) part[j] = 0.f;
      for (int c = 0; c < nchunk; ++c) {
        const int e = (c * 16 + l16) * 4;
        const float4 xv = ld ? load4<VEC>(frow, e, p.F) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < HB; ++j) {
          if (h0 + j < H) {
            const float4 pw = *reinterpret_cast<const float4*>(sP + (hb + j) * FP + e);
            part[j] = dot4_fma(xv, pw, part[j]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < HB; ++j) {
        const int h = h0 + j;
        const float s = row16_sum(part[j]);
        if (h < H) {
          const float bit = (s < 0.f) ? 0.f : 1.f;
          cnt = cnt + bit;
          if (want_emb) {
#pragma unroll
            for (int c = 0; c < DC; ++c) {
              const float4 bw = *reinterpret_cast<const float4*>(sW + (hb + j) * DP + (c * 16 + l16) * 4);
              acc[c].x = __builtin_fmaf(bit, bw.x, acc[c].x);
              acc[c].y = __builtin_fmaf(bit, bw.y, acc[c].y);
              acc[c].z = __builtin_fmaf(bit, bw.z, acc[c].z);
              acc[c].w = __builtin_fmaf(bit, bw.w, acc[c].w);
            }
          }
          if (p.bits && l16 == 0 && live)
            p.bits[row * p.H + h] = valid ? static_cast<uint8_t>(bit) : static_cast<uint8_t>(0xFF);
        }
      }
    }

    if (want_emb) {
      float4 emb[DC];
      float sp = 0.f;
      const float* orow = p.other ? p.other + (live ? row : 0) * p.D : nullptr;
#pragma unroll
      for (int c = 0; c < DC; ++c) {
        const int e = (c * 16 + l16) * 4;
        if (oov) {
          emb[c].x = acc[c].x / cnt;
          emb[c].y = acc[c].y / cnt;
          emb[c].z = acc[c].z / cnt;
          emb[c].w = acc[c].w / cnt;
        } else {
          emb[c] = valid ? load4<VEC>(p.table + id * p.D + p.d0, e, p.Dw) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (!valid) emb[c] = make_float4(qnan(), qnan(), qnan(), qnan());
        if (p.out && live) store4<VEC>(p.out + row * p.D + p.d0, e, p.Dw, emb[c]);
        if (p.score) {
          float4 o = live ? load4<VEC>(orow, e, p.D) : make_float4(0.f, 0.f, 0.f, 0.f);
          float4 m = emb[c];
          if (e + 0 >= p.D) m.x = 0.f;
          if (e + 1 >= p.D) m.y = 0.f;
          if (e + 2 >= p.D) m.z = 0.f;
          if (e + 3 >= p.D) m.w = 0.f;
          sp = dot4_muladd(o, m, sp);
        }
      }
      if (p.score) {
        const float s = row16_sum(sp);
        if (l16 == 0 && live) p.score[row] = s;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// slsh: popcount of H = bits_req projections -> bucket index -> single bucket-row gather.
// Planes in LDS; bucket rows come from HBM (n_buckets may be as large as the catalogue).
// Chunks of the feature row are streamed (any F).
// ------------------------------------------------------------------------------------------
template <bool VEC, bool CHUNK = false>
__global__ __launch_bounds__(kBlock) void slsh_kernel(LshParams p, int FP) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int HB = 8;
  constexpr int R = 4;
  const int lane = threadIdx.x & 63;
  const int l16 = lane & 15;
  const int grp = lane >> 4;
  const int wv = threadIdx.x >> 6;
  const int64_t ntiles = (p.B + 4 * R - 1) / (4 * R);
  const int64_t tile0 = static_cast<int64_t>(blockIdx.x) * 4 + wv;
  // ids of the first tile before the planes are staged: the first hop of ids -> feature row -> bucket row
  // overlaps the staging + barrier (same ordering as lsh64_kernel)
  int64_t idn[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t row = tile0 * (4 * R) + r * 4 + grp;
    idn[r] = p.ids[row < p.B ? row : p.B - 1];
  }
  float* sP = smem;
  const int H = static_cast<int>(p.H);
  const int HC = CHUNK ? p.h_chunk : H;  // CHUNK: h_chunk (a multiple of 8) planes in LDS at a time, waves in lock step
  if constexpr (!CHUNK) {
    stage_padded(sP, p.planes, p.H, p.F, FP);
    __syncthreads();
  }
  const int nchunk = FP / 64;
  const int dchunks = static_cast<int>((p.D + 63) / 64);

  for (int64_t tile = tile0; (CHUNK ? tile - wv : tile) < ntiles; tile += static_cast<int64_t>(gridDim.x) * 4) {
    int64_t row[R], id[R];
    bool live[R], valid[R];
    int pop[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      row[r] = tile * (4 * R) + r * 4 + grp;
      live[r] = row[r] < p.B;
      id[r] = (tile == tile0) ? idn[r] : p.ids[live[r] ? row[r] : p.B - 1];  // clamped, never branched on
      valid[r] = live[r] && (static_cast<uint64_t>(id[r]) < static_cast<uint64_t>(p.N));
      if (!valid[r]) id[r] = 0;
      pop[r] = 0;
    }
    for (int h0 = 0; h0 < H; h0 += HB) {
      if constexpr (CHUNK) {
        if (h0 % HC == 0) {  // (uniform)
          __syncthreads();
          stage_padded(sP, p.planes + static_cast<int64_t>(h0) * p.F, (H - h0 < HC) ? H - h0 : HC, p.F, FP);
          __syncthreads();
        }
      }
      const int hb = CHUNK ? h0 % HC : h0;  // LDS row of plane h0
      float part[R][HB];
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < HB; ++j) part[r][j] = 0.f;
      for (int c = 0; c < nchunk; ++c) {
        const int e = (c * 16 + l16) * 4;
        float4 xv[R];
#pragma unroll
        for (int r = 0; r < R; ++r)
          xv[r] = load4<VEC>(p.feat + id[r] * p.F, e, p.F);
#pragma unroll
        for (int j = 0; j < HB; ++j) {
          if (h0 + j < H) {
            const float4 pw = *reinterpret_cast<const float4*>(sP + (hb + j) * FP + e);
#pragma unroll
            for (int r = 0; r < R; ++r) part[r][j] = dot4_fma(xv[r], pw, part[r][j]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < HB; ++j) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float s = row16_sum(part[r][j]);
          if (h0 + j < H) pop[r] += (s < 0.f) ? 0 : 1;
        }
      }
    }
    // (2 ** bits).sum(1) = sum of 1 or 2 per plane = H + popcount  (single_lsh_embedder.py:86).
    // All R bucket ids first, then all R bucket-row gathers back-to-back, then the stores: the
    // second dependent HBM round trip is paid once per tile, not once per lookup.
    int64_t bkt[R];
#pragma unroll
    for (int r = 0; r < R; ++r) bkt[r] = valid[r] ? (static_cast<int64_t>(H) + pop[r]) % p.n_buckets : -1;
    if (p.idx && !p.out) {
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (l16 == 0 && live[r]) p.idx[row[r]] = bkt[r];
    }
    if (p.out) {
      for (int c = 0; c < dchunks; ++c) {
        const int e = (c * 16 + l16) * 4;
        float4 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = load4<VEC>(p.buckets + (valid[r] ? bkt[r] : 0) * p.D, e, p.D);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (!valid[r]) v[r] = make_float4(qnan(), qnan(), qnan(), qnan());
          if (live[r]) store4<VEC>(p.out + row[r] * p.D, e, p.D, v[r]);
        }
      }
      if (p.idx) {  // after the gathers: a store between them would sit in front of their vmcnt waits
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (l16 == 0 && live[r]) p.idx[row[r]] = bkt[r];
      }
    }
  }
}

// planes per LDS chunk of the chunked form: 64 (32 KB at F = D = 64: four workgroups per CU); developer knob
static int lsh_h_chunk() {
  static const int v = static_cast<int>(env_knob("MI_OOV_LSH_HCHUNK", 64, 8, 1024));
  return v;
}

template <int FC, int DC, bool VEC>
static int launch_fused(const LshParams& p, hipStream_t st, bool chunked) {
  constexpr int R = (FC >= 4 || DC >= 4) ? 2 : 4;
  const int grid = grid_for(p.B, 16 * R);
  if (chunked) {
    LshParams q = p;
    int hc = lsh_h_chunk();
    while (hc > 8 && static_cast<int64_t>(hc) * (FC * 64 + DC * 64) * 4 > 64 * 1024) hc /= 2;  // <= 64 KB per workgroup
    q.h_chunk = hc;
    const size_t lds = static_cast<size_t>(hc) * (FC * 64 + DC * 64) * sizeof(float);
    auto k = lsh_fused_kernel<FC, DC, VEC, R, true>;
    if (int rc = set_lds(k, lds)) return rc;
    hipLaunchKernelGGL(k, dim3(grid), dim3(kBlock), lds, st, q);
    return check_launch();
  }
  const size_t lds = static_cast<size_t>(p.H) * (FC * 64 + DC * 64) * sizeof(float);
  auto k = lsh_fused_kernel<FC, DC, VEC, R>;
  if (int rc = set_lds(k, lds)) return rc;
  hipLaunchKernelGGL(k, dim3(grid), dim3(kBlock), lds, st, p);
  return check_launch();
}

// planes per chunk of the wide / slsh chunked forms: the largest multiple of 8 that keeps the chunk within `budget` bytes
static int planes_per_chunk(int64_t floats_per_plane, int64_t budget) {
  int64_t hc = budget / (floats_per_plane * 4) / 8 * 8;
  return static_cast<int>(hc < 8 ? 8 : hc);
}

template <int DC, bool VEC>
static int launch_wide(const LshParams& p, int FP, hipStream_t st, bool chunked) {
  const int grid = grid_for(p.B, 16);
  if (chunked) {
    LshParams q = p;
    q.h_chunk = planes_per_chunk(FP + DC * 64, 48 * 1024);
    const size_t lds = static_cast<size_t>(q.h_chunk) * (FP + DC * 64) * sizeof(float);
    if (static_cast<int64_t>(lds) > kLdsLimit) return MI_OOV_ERR_SHAPE;  // eight planes do not fit: F beyond ~5000 floats
    auto k = lsh_wide_kernel<DC, VEC, true>;
    if (int rc = set_lds(k, lds)) return rc;
    hipLaunchKernelGGL(k, dim3(grid), dim3(kBlock), lds, st, q, FP);
    return check_launch();
  }
  const size_t lds = static_cast<size_t>(p.H) * (FP + DC * 64) * sizeof(float);
  auto k = lsh_wide_kernel<DC, VEC>;
  if (int rc = set_lds(k, lds)) return rc;
  hipLaunchKernelGGL(k, dim3(grid), dim3(kBlock), lds, st, p, FP);
  return check_launch();
}

template <bool VEC>
static int dispatch_lsh_window(const LshParams& p, hipStream_t st) {
  const int fc = static_cast<int>((p.F + 63) / 64);
  const int dc = static_cast<int>((p.Dw + 63) / 64);
  const int fcq = fc <= 1 ? 1 : (fc <= 2 ? 2 : (fc <= 4 ? 4 : 0));
  const int dcq = dc <= 1 ? 1 : (dc <= 2 ? 2 : 4);
  const int FPw = fc * 64;
  const int64_t lds = p.H * ((fcq ? fcq * 64 : FPw) + dcq * 64) * static_cast<int64_t>(sizeof(float));
  // All planes and bucket rows resident up to 32 KB (four workgroups per CU); more planes than that are
  // staged a chunk at a time (F <= 256).  Wider rows with more planes than fit: no kernel.
  static const int64_t chunk_from = env_knob("MI_OOV_LSH_CHUNK_FROM_BYTES", 32 * 1024, 0, kLdsLimit);
  const bool chunked = fcq != 0 ? lds > chunk_from : lds > kLdsLimit;  // (wide rows: chunked only when they must be)
#define MI_FUSED(FCV, DCV) \
  if (fcq == FCV && dcq == DCV) return launch_fused<FCV, DCV, VEC>(p, st, chunked);
  MI_FUSED(1, 1) MI_FUSED(1, 2) MI_FUSED(1, 4)
  MI_FUSED(2, 1) MI_FUSED(2, 2) MI_FUSED(2, 4)
  MI_FUSED(4, 1) MI_FUSED(4, 2) MI_FUSED(4, 4)
#undef MI_FUSED
  if (dcq == 1) return launch_wide<1, VEC>(p, FPw, st, chunked);
  if (dcq == 2) return launch_wide<2, VEC>(p, FPw, st, chunked);
  return launch_wide<4, VEC>(p, FPw, st, chunked);
}

// Embedding rows wider than the 256 floats a lane group keeps in registers (embedding_size 300, 512, ...): one launch per
// window of 256 output columns -- the projections are recomputed per window, the bucket-row chain of a column does not
// depend on the other columns, so every window holds the bits a single launch would.  The fused score needs the whole
// row in one launch: refused here (MI_OOV_ERR_SHAPE), the host composes rows + mi_oov_rowdot.
template <bool VEC>
static int dispatch_lsh(LshParams p, hipStream_t st) {
  if (p.D <= 256) {
    p.d0 = 0;
    p.Dw = p.D;
    return dispatch_lsh_window<VEC>(p, st);
  }
  if (p.score) return MI_OOV_ERR_SHAPE;
  for (int64_t d0 = 0; d0 < p.D; d0 += 256) {
    LshParams q = p;
    q.d0 = d0;
    q.Dw = (p.D - d0 < 256) ? p.D - d0 : 256;
    if (d0) q.bits = nullptr;  // the codes leave with the first window
    if (int rc = dispatch_lsh_window<VEC>(q, st)) return rc;
  }
  return MI_OOV_OK;
}

// lsh64.hip: lane-owns-lookup kernel for the hot shape F = D = 64
int launch_lsh64(const int64_t* ids, int64_t B, const float* feat, int64_t N, const float* vtable, int64_t n_vocab,
                 const float* planes, int H, const float* buckets, const float* other, float* score, float* out,
                 hipStream_t st, uint8_t* bits);

int launch_slsh64(const void* ids, int64_t B, int64_t K, bool tab, const float* feat, int64_t N, const float* planes, int H,
                  const float* buckets, int64_t n_buckets, int64_t D, void* out, void* idx, hipStream_t st);

static bool lsh64_enabled() {
  static const bool on = env_knob("MI_OOV_LSH64", 1, 0, 1) != 0;  // developer A/B knob; default on
  return on;
}

static int run_lsh(LshParams p, void* stream) {
  if (p.B < 0 || p.N <= 0 || p.F <= 0 || p.H <= 0) return MI_OOV_ERR_SHAPE;
  if (p.B == 0) return MI_OOV_OK;
  if (!p.ids || !p.feat || !p.planes) return MI_OOV_ERR_NULL;
  const bool want_emb = p.out || p.score;
  if (want_emb && (!p.buckets || p.D <= 0)) return MI_OOV_ERR_NULL;
  if (!want_emb && !p.bits) return MI_OOV_ERR_NULL;
  if (p.score && !p.other) return MI_OOV_ERR_NULL;
  if (p.table && p.n_vocab < 0) return MI_OOV_ERR_SHAPE;
  if (p.D <= 0) p.D = 1;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool vec = (p.F % 4 == 0) && (!want_emb || p.D % 4 == 0) && aligned16(p.feat) && (!p.out || aligned16(p.out)) &&
                   (!p.other || aligned16(p.other)) && (!p.table || aligned16(p.table));
  // hot shape: F = D = 64 (D is irrelevant for a codes-only call), H <= 8; codes from that kernel only for H = 8
  // without score / in-vocabulary table (the training forward and TorchLSHash.hash_points)
  const bool d_ok = want_emb ? (p.D == 64 && aligned16(p.buckets)) : true;
  const bool bits_ok = !p.bits || (p.H == 8 && !p.score && !p.table && (reinterpret_cast<uintptr_t>(p.bits) & 7u) == 0);
  const bool h_ok = p.H <= 8 || (p.H <= 64 && want_emb && !p.bits);  // 9..64 planes: lsh64g_kernel, embeddings only
  if (vec && p.F == 64 && d_ok && h_ok && bits_ok && aligned16(p.planes) && lsh64_enabled())
    return launch_lsh64(p.ids, p.B, p.feat, p.N, p.table, p.n_vocab, p.planes, static_cast<int>(p.H), p.buckets,
                        p.other, p.score, p.out, st, p.bits);
  return vec ? dispatch_lsh<true>(p, st) : dispatch_lsh<false>(p, st);
}

}  // namespace mi_oov

using mi_oov::LshParams;

extern "C" int mi_oov_lsh_embed(const int64_t* ids, int64_t B, const float* feat, int64_t N, int64_t F,
                                const float* planes, int64_t H, const float* buckets, int64_t D, float* out,
                                uint8_t* bits, void* stream) {
  LshParams p{};
  p.ids = ids; p.B = B; p.feat = feat; p.N = N; p.F = F; p.planes = planes; p.H = H;
  p.buckets = buckets; p.n_buckets = H; p.D = D; p.out = out; p.bits = bits;
  return mi_oov::run_lsh(p, stream);
}

extern "C" int mi_oov_lsh_embed_score(const int64_t* ids, int64_t B, const float* feat, int64_t N, int64_t F,
                                      const float* planes, int64_t H, const float* buckets, int64_t D,
                                      const float* other, float* score, float* out, void* stream) {
  if (!score) return MI_OOV_ERR_NULL;
  LshParams p{};
  p.ids = ids; p.B = B; p.feat = feat; p.N = N; p.F = F; p.planes = planes; p.H = H;
  p.buckets = buckets; p.n_buckets = H; p.D = D; p.other = other; p.score = score; p.out = out;
  return mi_oov::run_lsh(p, stream);
}

extern "C" int mi_oov_lsh_lookup(const int64_t* ids, int64_t B, const float* table, int64_t n_vocab,
                                 const float* feat, int64_t N, int64_t F, const float* planes, int64_t H,
                                 const float* buckets, int64_t D, float* out, void* stream) {
  if (!table || !out) return MI_OOV_ERR_NULL;
  LshParams p{};
  p.ids = ids; p.B = B; p.feat = feat; p.N = N; p.F = F; p.planes = planes; p.H = H;
  p.buckets = buckets; p.n_buckets = H; p.D = D; p.table = table; p.n_vocab = n_vocab; p.out = out;
  return mi_oov::run_lsh(p, stream);
}

extern "C" int mi_oov_lsh_lookup_score(const int64_t* ids, int64_t B, const float* table, int64_t n_vocab,
                                       const float* feat, int64_t N, int64_t F, const float* planes, int64_t H,
                                       const float* buckets, int64_t D, const float* other, float* score, float* out,
                                       void* stream) {
  if (!table || !score) return MI_OOV_ERR_NULL;
  LshParams p{};
  p.ids = ids; p.B = B; p.feat = feat; p.N = N; p.F = F; p.planes = planes; p.H = H;
  p.buckets = buckets; p.n_buckets = H; p.D = D; p.table = table; p.n_vocab = n_vocab;
  p.other = other; p.score = score; p.out = out;
  return mi_oov::run_lsh(p, stream);
}

extern "C" int mi_oov_slsh_embed(const int64_t* ids, int64_t B, const float* feat, int64_t N, int64_t F,
                                 const float* planes, int64_t H, const float* buckets, int64_t n_buckets,
                                 int64_t D, float* out, int64_t* idx, void* stream) {
  using namespace mi_oov;
  // H == 0: n_buckets == 1 gives bits_req = ceil(log2(1)) = 0 planes (single_lsh_embedder.py:77-80); every lookup then
  // lands in bucket (0 + 0) % n_buckets = 0.  `planes` may be NULL then (an empty tensor has no storage).
  if (B < 0 || N <= 0 || F <= 0 || H < 0 || n_buckets <= 0) return MI_OOV_ERR_SHAPE;
  if (B == 0) return MI_OOV_OK;
  if (!ids || !feat || (!planes && H > 0)) return MI_OOV_ERR_NULL;
  if (!out && !idx) return MI_OOV_ERR_NULL;
  if (out && (!buckets || D <= 0)) return MI_OOV_ERR_NULL;
  LshParams p{};
  p.ids = ids; p.B = B; p.feat = feat; p.N = N; p.F = F; p.planes = planes; p.H = H;
  p.buckets = buckets; p.n_buckets = n_buckets; p.D = D > 0 ? D : 1; p.out = out; p.idx = idx;
  const int FP = static_cast<int>((F + 63) / 64) * 64;
  size_t lds = static_cast<size_t>(H) * FP * sizeof(float);
  const bool chunked = static_cast<int64_t>(lds) > kLdsLimit;  // wide rows x many planes: the planes a chunk at a time
  if (chunked) {
    p.h_chunk = planes_per_chunk(FP, 48 * 1024);
    lds = static_cast<size_t>(p.h_chunk) * FP * sizeof(float);
    if (static_cast<int64_t>(lds) > kLdsLimit) return MI_OOV_ERR_SHAPE;  // eight planes do not fit: F beyond ~5000 floats
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool vec = (F % 4 == 0) && aligned16(feat) && (!out || (p.D % 4 == 0 && aligned16(out) && aligned16(buckets)));
  if (vec && F == 64 && H >= 1 && H <= 32 && (!out || D == 64 || D == 128) && aligned16(planes) && lsh64_enabled())
    return launch_slsh64(ids, B, 1, false, feat, N, planes, static_cast<int>(H), buckets, n_buckets, D, out, idx, st);
  const int grid = grid_for(B, 64);
#define MI_SLSH_GO(V, C) \
  do { \
    auto k = slsh_kernel<V, C>; \
    if (int rc = set_lds(k, lds)) return rc; \
    hipLaunchKernelGGL(k, dim3(grid), dim3(kBlock), lds, st, p, FP); \
  } while (0)
  if (vec && chunked) MI_SLSH_GO(true, true);
  else if (vec) MI_SLSH_GO(true, false);
  else if (chunked) MI_SLSH_GO(false, true);
  else MI_SLSH_GO(false, false);
#undef MI_SLSH_GO
  return check_launch();
}

// K queued batches of mi_oov_slsh_embed in ONE launch (hot tile only: F == 64, H <= 32, D 64 or 128 -- other shapes
// return MI_OOV_ERR_SHAPE and the caller issues K single launches).  ids_tab / out_tab / idx_tab: DEVICE arrays of K
// device pointers (out_tab or idx_tab may be NULL).
extern "C" int mi_oov_slsh_embed_multi(const int64_t* const* ids_tab, float* const* out_tab, int64_t* const* idx_tab, int64_t K,
                                       int64_t B, const float* feat, int64_t N, int64_t F, const float* planes, int64_t H,
                                       const float* buckets, int64_t n_buckets, int64_t D, void* stream) {
  using namespace mi_oov;
  if (K < 0 || B < 0 || N <= 0 || F <= 0 || H <= 0 || n_buckets <= 0) return MI_OOV_ERR_SHAPE;
  if (K == 0 || B == 0) return MI_OOV_OK;
  if (!ids_tab || !feat || !planes || (!out_tab && !idx_tab) || (out_tab && !buckets)) return MI_OOV_ERR_NULL;
  if (F != 64 || H > 32 || (out_tab && D != 64 && D != 128)) return MI_OOV_ERR_SHAPE;
  if (!aligned16(feat) || !aligned16(planes) || (out_tab && !aligned16(buckets)) || (reinterpret_cast<uintptr_t>(ids_tab) & 7u) ||
      (reinterpret_cast<uintptr_t>(out_tab) & 7u) || (reinterpret_cast<uintptr_t>(idx_tab) & 7u))
    return MI_OOV_ERR_ALIGN;
  return launch_slsh64(ids_tab, B, K, true, feat, N, planes, static_cast<int>(H), buckets, n_buckets, D > 0 ? D : 64,
                       const_cast<float**>(out_tab), const_cast<int64_t**>(idx_tab), static_cast<hipStream_t>(stream));
}
