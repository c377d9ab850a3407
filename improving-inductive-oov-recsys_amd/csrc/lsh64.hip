// lsh, hot shape F = D = 64 and H <= 8: register-resident specialisation of the fused kernel.
//
// Same decomposition as lsh_fused_kernel (lsh.hip): a 16-lane DPP row owns a lookup, lane l holds
// floats [4l, 4l+4) of the feature row and of the output row.  What changes for the hot shape:
//   * the lane's slice of every plane and of every bucket row (2 x H float4) lives in VGPRs for the
//     whole kernel: the inner loop touches neither LDS nor memory;
//   * the H-loop is fully unrolled; optional outputs are template parameters, not run-time branches;
//   * ids are clamped instead of branched on, so all R row gathers (and the R rows of the other
//     side, for the fused score) are issued back-to-back before the first wait: one HBM round trip
//     per tile instead of R;
//   * LOOKUP (BPR.get_*_embedding, bpr.py:48-125): because F == D the in-vocabulary row and the
//     feature row have the same width, so ONE gather per lookup serves both cases -- its base
//     pointer is selected by `id < n_vocab`; in-vocabulary lookups return the gathered row itself.
// Arithmetic and summation order are those of lsh_fused_kernel, so the two kernels and the oracle
// agree bit for bit.
//
// Measured on MI355X (65536 lookups, N = 10 M, H = 8; rocprofv3 kernel time, all bit-identical):
//   generic kernel, branchy gathers 16.8 us -> clamped gathers 14.9 us
//   this kernel (operands in VGPRs)  12.2 us storing rows / 11.4 us fused score
//   + SLP packing off (v_add_f32_dpp stays fused; -fno-slp-vectorize)  10.6 us fused score
// Tried and rejected: "lane owns lookup" through an LDS transpose, one wave per SIMD (23 us, LDS
// latency exposed); per-workgroup 2^H-row code table in LDS replacing aggregate + division (13.2 us
// with 1024-thread groups, 17.3 us with 256: build + barrier cost more than they save).
// Floor for this access pattern (tools/microbench.hip, back-to-back launches): empty launch 2.3 us,
// gather only 5.7 us, gather x and u rows 7.5 us, gather + 256-B row store 7.8 us.  Stage costs on top
// of the 7.5 us floor: 8 projections +1.1 us, aggregate +1.0, division +0.7, score +0.3.
// Per-wave s_memrealtime stamps (MB_STAMPS=1 tools/microbench): all 4096 waves start within 0.4 us,
// the kernel spans 8.4 us in-kernel (the other ~2.3 us of a launch are dispatch + boundary); ids
// land at p50 2.6 / max 5.2 us and a wave's four rows land over a further ~2.4 us: the 34.9 MB of a
// launch need >= 5.8 us at the ~6 TB/s the memory system sustains, i.e. the middle of the kernel is
// bandwidth-bound and only its head (launch, first dependent hop) and tail (last rounds' VALU)
// are not.  Re-ordering the issue (ids before weights, user rows after the gathers) changed nothing.
#include "common.hpp"

namespace mi_oov {

// One tile = 16 lookups of one wave (4 rounds x 4 groups).  FULL tiles (all 16 rows < B) skip every
// tail clamp and liveness test; only the last tile of a launch can be partial.
template <int H, bool SCORE, bool STORE, bool LOOKUP, bool FULL>
__device__ __forceinline__ void lsh64_tile(int64_t tile, int l16, int grp, const float4 (&pw)[H], const float4 (&bw)[H],
                                           const int64_t* __restrict__ ids, int64_t B,
                                           const float* __restrict__ feat, int64_t N,
                                           const float* __restrict__ vtable, int64_t n_vocab,
                                           const float* __restrict__ other, float* __restrict__ score,
                                           float* __restrict__ out) {
  constexpr int R = 4;
  int64_t row[R], idc[R];
  bool valid[R], oov[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    row[r] = tile * (4 * R) + r * 4 + grp;
    idc[r] = ids[(FULL || row[r] < B) ? row[r] : B - 1];  // clamped: tail groups recompute the last row
  }
  float4 u[SCORE ? R : 1];
  if (SCORE) {
#pragma unroll
    for (int r = 0; r < R; ++r)
      u[r] = *reinterpret_cast<const float4*>(other + ((FULL || row[r] < B) ? row[r] : B - 1) * 64 + l16 * 4);
  }
  float4 x[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    oov[r] = !LOOKUP || idc[r] >= n_vocab;
    valid[r] = oov[r] ? static_cast<uint64_t>(idc[r]) < static_cast<uint64_t>(N) : idc[r] >= 0;
    const float* base = oov[r] ? feat : vtable;
    x[r] = *reinterpret_cast<const float4*>(base + (valid[r] ? idc[r] : 0) * 64 + l16 * 4);
  }

#pragma unroll
  for (int r = 0; r < R; ++r) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float cnt = 0.f;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const float s = row16_sum(dot4_fma(x[r], pw[h], 0.f));
      const float bit = (s < 0.f) ? 0.f : 1.f;  // >= 0, +-0 and NaN -> 1 (torch_hash.py:57-59)
      cnt = cnt + bit;
      acc.x = __builtin_fmaf(bit, bw[h].x, acc.x);
      acc.y = __builtin_fmaf(bit, bw[h].y, acc.y);
      acc.z = __builtin_fmaf(bit, bw[h].z, acc.z);
      acc.w = __builtin_fmaf(bit, bw[h].w, acc.w);
    }
    // emb = acc / cnt, correctly rounded (0/0 -> NaN row, lsh_embedder.py:178).  cnt is an integer in
    // [0, 8], so instead of four IEEE division sequences (~9 VALU each) one reciprocal r = RN(1/cnt)
    // is shared and each quotient is refined once:  q = a r;  e = fma(-q, cnt, a);  q' = fma(e, r, q).
    // Verified EXHAUSTIVELY on the CPU (same IEEE fma) for cnt = 1..8 and all 2^32 values of a: q'
    // equals RN(a / cnt) except for a = -0 (acc is never -0: it starts at +0 and every step is
    // RN(bit*w + acc)) and for |a| < 2^-124 with cnt = 6 (result subnormal).  Lanes whose |acc| is
    // below 2^-100 (including exact zeros, i.e. the cnt = 0 rows) or infinite take the IEEE division.
    const float rc = 1.0f / cnt;
    float4 emb;
    {
      float q;
      q = acc.x * rc; emb.x = __builtin_fmaf(__builtin_fmaf(-q, cnt, acc.x), rc, q);
      q = acc.y * rc; emb.y = __builtin_fmaf(__builtin_fmaf(-q, cnt, acc.y), rc, q);
      q = acc.z * rc; emb.z = __builtin_fmaf(__builtin_fmaf(-q, cnt, acc.z), rc, q);
      q = acc.w * rc; emb.w = __builtin_fmaf(__builtin_fmaf(-q, cnt, acc.w), rc, q);
    }
    const float amin = fminf(fminf(fabsf(acc.x), fabsf(acc.y)), fminf(fabsf(acc.z), fabsf(acc.w)));
    const float amax = fmaxf(fmaxf(fabsf(acc.x), fabsf(acc.y)), fmaxf(fabsf(acc.z), fabsf(acc.w)));
    if (!(amin >= 0x1p-100f) || !(amax < __builtin_inff())) {
      emb.x = acc.x / cnt;
      emb.y = acc.y / cnt;
      emb.z = acc.z / cnt;
      emb.w = acc.w / cnt;
    }
    if (LOOKUP && !oov[r]) emb = x[r];
    if (!valid[r]) emb = make_float4(qnan(), qnan(), qnan(), qnan());
    const bool live = FULL || row[r] < B;
    if (STORE && live) *reinterpret_cast<float4*>(out + row[r] * 64 + l16 * 4) = emb;
    if (SCORE) {
      const float s = row16_sum(dot4_muladd(u[r], emb, 0.f));
      if (l16 == 0 && live) score[row[r]] = s;
    }
  }
}

template <int H, bool SCORE, bool STORE, bool LOOKUP>
__global__ __launch_bounds__(kBlock) void lsh64_kernel(const int64_t* __restrict__ ids, int64_t B,
                                                       const float* __restrict__ feat, int64_t N,
                                                       const float* __restrict__ vtable, int64_t n_vocab,
                                                       const float* __restrict__ planes,
                                                       const float* __restrict__ buckets,
                                                       const float* __restrict__ other,
                                                       float* __restrict__ score, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, l16 = lane & 15, grp = lane >> 4, wv = threadIdx.x >> 6;
  const int64_t ntiles = (B + 15) / 16;
  const int64_t nfull = B / 16;
  const int64_t tstep = static_cast<int64_t>(gridDim.x) * 4;

  // Plane / bucket slices -> VGPRs through LDS: the workgroup fetches the 2 x H x 256 B once (one 16-B
  // load per thread) and every lane reads its 2 x H float4 back with ds_read_b128, instead of 2 x H
  // global loads per lane (16 KiB of L1 traffic per wave) queued in front of the ids -> rows gathers
  // (-0.4 us in tools/microbench.hip).
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [2][H][64]
  for (int i = threadIdx.x; i < 2 * H * 16; i += kBlock) {
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

  for (int64_t tile = static_cast<int64_t>(blockIdx.x) * 4 + wv; tile < ntiles; tile += tstep) {
    if (tile < nfull)
      lsh64_tile<H, SCORE, STORE, LOOKUP, true>(tile, l16, grp, pw, bw, ids, B, feat, N, vtable, n_vocab, other, score, out);
    else
      lsh64_tile<H, SCORE, STORE, LOOKUP, false>(tile, l16, grp, pw, bw, ids, B, feat, N, vtable, n_vocab, other, score, out);
  }
}

template <int H, bool SCORE, bool STORE, bool LOOKUP>
static int launch64(const int64_t* ids, int64_t B, const float* feat, int64_t N, const float* vtable, int64_t n_vocab,
                    const float* planes, const float* buckets, const float* other, float* score, float* out,
                    hipStream_t st) {
  const int grid = grid_for(B, 64);  // 4 waves x 16 lookups per workgroup pass
  hipLaunchKernelGGL((lsh64_kernel<H, SCORE, STORE, LOOKUP>), dim3(grid), dim3(kBlock), 2 * H * 64 * sizeof(float), st, ids, B, feat, N, vtable,
                     n_vocab, planes, buckets, other, score, out);
  return check_launch();
}

template <int H>
static int launch64_h(const int64_t* ids, int64_t B, const float* feat, int64_t N, const float* vtable,
                      int64_t n_vocab, const float* planes, const float* buckets, const float* other, float* score,
                      float* out, hipStream_t st) {
#define MI_GO(S, T, L) return launch64<H, S, T, L>(ids, B, feat, N, vtable, n_vocab, planes, buckets, other, score, out, st)
  if (vtable) {
    if (score && out) MI_GO(true, true, true);
    if (score) MI_GO(true, false, true);
    MI_GO(false, true, true);
  }
  if (score && out) MI_GO(true, true, false);
  if (score) MI_GO(true, false, false);
  MI_GO(false, true, false);
#undef MI_GO
}

// Host entry used by run_lsh (lsh.hip) when the shape qualifies: F = D = 64, 1 <= H <= 8.
int launch_lsh64(const int64_t* ids, int64_t B, const float* feat, int64_t N, const float* vtable, int64_t n_vocab,
                 const float* planes, int H, const float* buckets, const float* other, float* score, float* out,
                 hipStream_t st) {
  switch (H) {
#define MI_CASE(HV) \
  case HV: return launch64_h<HV>(ids, B, feat, N, vtable, n_vocab, planes, buckets, other, score, out, st);
    MI_CASE(1) MI_CASE(2) MI_CASE(3) MI_CASE(4) MI_CASE(5) MI_CASE(6) MI_CASE(7) MI_CASE(8)
#undef MI_CASE
    default: return MI_OOV_ERR_SHAPE;
  }
}

}  // namespace mi_oov
