// lsh, hot shape F = D = 64 and H <= 8: the fused kernel with a per-workgroup CODE TABLE.
//
// With H <= 8 planes a lookup's H sign bits form a code in [0, 2^H).  The embedding depends on the
// code only:  emb(code) = (sum_{h in code} W[h]) / popcount(code), so each workgroup builds the
// 2^H x 64 table of all possible output rows in LDS (64 KiB for H = 8) WHILE its row gathers are in
// flight -- the VALU is idle during the ~4 us ids -> rows dependent HBM round trips anyway -- and a
// lookup then costs only its H projections plus ONE ds_read_b128 of its slice of table[code].
// The per-lookup aggregate (4 FMA per plane per lane) and the 64 IEEE divisions per lookup of the
// generic kernel are gone; what remains per lookup is 8 x (4 FMA + 4 DPP adds) shared by 16 lanes.
//
// Table entries are computed with the very chain the generic kernel uses per lookup (fmaf over
// h = 0..H-1 from +0, then one IEEE division by the exact float count; code 0 -> 0/0 = NaN row), so
// results are bit-identical to lsh_fused_kernel and to oracle/oov_oracle.c.
//
// Measured on MI355X (65536 lookups, N = 10 M, H = 8; kernel time from rocprofv3):
//   generic lsh_fused_kernel 16.8 us (branchy gathers) -> 14.9 us (clamped gathers)
//   plane/bucket rows in VGPRs, unrolled                12.7 us   (VALU-bound: ~42 instr/lookup)
//   "lane owns lookup" via LDS transpose, 1 wave/SIMD   23 us     (LDS latency exposed)
//   this kernel                                         see profiles/
// Floor for this access pattern (tools/microbench.hip): gather+store 7.8 us, gather only 5.7 us.
#include <stdlib.h>

#include "common.hpp"

namespace mi_oov {

constexpr int kTblBlock = 1024;  // 16 waves: one workgroup per CU, 256 lookups per pass

template <int H, bool SCORE, bool STORE, bool TABLE, int BLOCK>
__global__ __launch_bounds__(BLOCK) void lsh_table_kernel(const int64_t* __restrict__ ids, int64_t B,
                                                              const float* __restrict__ feat, int64_t N,
                                                              const float* __restrict__ planes,
                                                              const float* __restrict__ buckets,
                                                              const float* __restrict__ other,
                                                              float* __restrict__ score, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float4 table[];  // [1 << H][16] float4
  constexpr int R = 4;
  const int tid = threadIdx.x;
  const int lane = tid & 63, l16 = lane & 15, grp = lane >> 4, wv = tid >> 6;
  const int64_t ntiles = (B + 4 * R - 1) / (4 * R);
  const int64_t tstep = static_cast<int64_t>(gridDim.x) * (BLOCK / 64);
  int64_t tile = static_cast<int64_t>(blockIdx.x) * (BLOCK / 64) + wv;

  int64_t row[R];
  bool valid[R];
  float4 x[R];
  float4 u[SCORE ? R : 1];

  auto issue = [&](int64_t t) {
    int64_t idc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      row[r] = t * (4 * R) + r * 4 + grp;
      idc[r] = ids[row[r] < B ? row[r] : B - 1];  // clamped: tail groups recompute the last row
    }
    if (SCORE) {
#pragma unroll
      for (int r = 0; r < R; ++r)
        u[r] = *reinterpret_cast<const float4*>(other + (row[r] < B ? row[r] : B - 1) * 64 + l16 * 4);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      valid[r] = static_cast<uint64_t>(idc[r]) < static_cast<uint64_t>(N);
      x[r] = *reinterpret_cast<const float4*>(feat + (valid[r] ? idc[r] : 0) * 64 + l16 * 4);
    }
  };

  // Small, L2-resident operands FIRST: vmcnt retires in order, so anything issued after the
  // gathers could not be consumed before the gathers have landed.
  float4 pw[H], bw[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    pw[h] = *reinterpret_cast<const float4*>(planes + h * 64 + l16 * 4);
    bw[h] = *reinterpret_cast<const float4*>(buckets + h * 64 + l16 * 4);
  }
  bool have = tile < ntiles;
  // ids -> row gathers; the table build below overlaps their latency.  Issued unconditionally (idle
  // waves re-read tile 0): a branch here makes the compiler's vmcnt bookkeeping assume the worst
  // path and park the build behind the gathers.
  issue(have ? tile : 0);

  // ---- code table: thread -> (column chunk l16, codes tid/16 + 64 j) ------------------------------
  for (int code = tid >> 4; TABLE && code < (1 << H); code += BLOCK / 16) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float cnt = 0.f;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const float bit = ((code >> h) & 1) ? 1.f : 0.f;
      cnt = cnt + bit;
      acc.x = __builtin_fmaf(bit, bw[h].x, acc.x);
      acc.y = __builtin_fmaf(bit, bw[h].y, acc.y);
      acc.z = __builtin_fmaf(bit, bw[h].z, acc.z);
      acc.w = __builtin_fmaf(bit, bw[h].w, acc.w);
    }
    float4 v;
    v.x = acc.x / cnt;  // code 0: 0/0 -> NaN row (lsh_embedder.py:178)
    v.y = acc.y / cnt;
    v.z = acc.z / cnt;
    v.w = acc.w / cnt;
    table[code * 16 + l16] = v;
  }
  if (TABLE) __syncthreads();

  while (have) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float4 emb;
      if (TABLE) {
        int code = 0;
#pragma unroll
        for (int h = 0; h < H; ++h) {
          const float s = row16_sum(dot4_fma(x[r], pw[h], 0.f));
          code |= (s < 0.f) ? 0 : (1 << h);  // >= 0, +-0 and NaN -> bit 1 (torch_hash.py:57-59)
        }
        emb = table[code * 16 + l16];
      } else {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float cnt = 0.f;
#pragma unroll
        for (int h = 0; h < H; ++h) {
          const float s = row16_sum(dot4_fma(x[r], pw[h], 0.f));
          const float bit = (s < 0.f) ? 0.f : 1.f;
          cnt = cnt + bit;
          acc.x = __builtin_fmaf(bit, bw[h].x, acc.x);
          acc.y = __builtin_fmaf(bit, bw[h].y, acc.y);
          acc.z = __builtin_fmaf(bit, bw[h].z, acc.z);
          acc.w = __builtin_fmaf(bit, bw[h].w, acc.w);
        }
        emb.x = acc.x / cnt;
        emb.y = acc.y / cnt;
        emb.z = acc.z / cnt;
        emb.w = acc.w / cnt;
      }
      if (!valid[r]) emb = make_float4(qnan(), qnan(), qnan(), qnan());
      const bool live = row[r] < B;
      if (STORE && live) *reinterpret_cast<float4*>(out + row[r] * 64 + l16 * 4) = emb;
      if (SCORE) {
        const float s = row16_sum(dot4_muladd(u[r], emb, 0.f));
        if (l16 == 0 && live) score[row[r]] = s;
      }
    }
    tile += tstep;
    have = tile < ntiles;
    if (have) issue(tile);
  }
}

static int lsh64_variant() {
  static const int v = [] {
    const char* e = getenv("MI_OOV_LSH64_VARIANT");  // developer A/B knob: 0 table/1024, 1 table/256, 2 regs/256
    return e ? atoi(e) : 2;
  }();
  return v;
}

template <int H, bool SCORE, bool STORE, bool TABLE, int BLOCK>
static int launch_variant(const int64_t* ids, int64_t B, const float* feat, int64_t N, const float* planes,
                          const float* buckets, const float* other, float* score, float* out, hipStream_t st) {
  const size_t lds = TABLE ? (static_cast<size_t>(1) << H) * 64 * sizeof(float) : 0;
  auto k = lsh_table_kernel<H, SCORE, STORE, TABLE, BLOCK>;
  if (int rc = set_lds(k, lds)) return rc;
  const int64_t per_block = 16 * (BLOCK / 64);  // lookups per workgroup pass
  int64_t blocks = (B + per_block - 1) / per_block;
  const int64_t max_blocks = TABLE ? (BLOCK == 1024 ? 256 : 512) : 2048;
  if (blocks > max_blocks) blocks = max_blocks;
  hipLaunchKernelGGL(k, dim3(static_cast<unsigned>(blocks)), dim3(BLOCK), lds, st, ids, B, feat, N, planes, buckets,
                     other, score, out);
  return check_launch();
}

template <int H, bool SCORE, bool STORE>
static int launch_table(const int64_t* ids, int64_t B, const float* feat, int64_t N, const float* planes,
                        const float* buckets, const float* other, float* score, float* out, hipStream_t st) {
  switch (lsh64_variant()) {
    case 0: return launch_variant<H, SCORE, STORE, true, 1024>(ids, B, feat, N, planes, buckets, other, score, out, st);
    case 1: return launch_variant<H, SCORE, STORE, true, 256>(ids, B, feat, N, planes, buckets, other, score, out, st);
    default: return launch_variant<H, SCORE, STORE, false, 256>(ids, B, feat, N, planes, buckets, other, score, out, st);
  }
}

template <int H>
static int launch_table_h(const int64_t* ids, int64_t B, const float* feat, int64_t N, const float* planes,
                          const float* buckets, const float* other, float* score, float* out, hipStream_t st) {
  if (score && out) return launch_table<H, true, true>(ids, B, feat, N, planes, buckets, other, score, out, st);
  if (score) return launch_table<H, true, false>(ids, B, feat, N, planes, buckets, other, score, out, st);
  return launch_table<H, false, true>(ids, B, feat, N, planes, buckets, other, score, out, st);
}

// Host entry used by run_lsh (lsh.hip) when the shape qualifies: F = D = 64, 1 <= H <= 8.
int launch_lsh64(const int64_t* ids, int64_t B, const float* feat, int64_t N, const float* planes, int H,
                 const float* buckets, const float* other, float* score, float* out, hipStream_t st) {
  switch (H) {
#define MI_CASE(HV) \
  case HV: return launch_table_h<HV>(ids, B, feat, N, planes, buckets, other, score, out, st);
    MI_CASE(1) MI_CASE(2) MI_CASE(3) MI_CASE(4) MI_CASE(5) MI_CASE(6) MI_CASE(7) MI_CASE(8)
#undef MI_CASE
    default: return MI_OOV_ERR_SHAPE;
  }
}

}  // namespace mi_oov
