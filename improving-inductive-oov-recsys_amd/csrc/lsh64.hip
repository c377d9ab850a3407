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
// Measured on MI355X (65536 lookups, N = 10 M, H = 8; all versions bit-identical):
//   generic kernel, branchy gathers 16.8 us -> clamped gathers 14.9 us
//   this kernel (operands in VGPRs)  12.2 us storing rows / 11.4 us fused score
//   + SLP packing off (v_add_f32_dpp stays fused; -fno-slp-vectorize)  10.6 us fused score
//   + Markstein division, full-tile fast path                         10.1 us (graph replay, see below)
//   + ids requested before the weights are staged, 32-bit batch offsets (119 VGPRs), bank-masked
//     8-plane DPP reduce, Newton reciprocal                            9.40 us
//   + gathers issued before the user rows (forced)                     9.09 us
//   + kernel arguments preloaded into SGPRs (-amdgpu-kernarg-preload-count=16)  8.90 us
//   + the 16 scores of a tile stored by ONE instruction at the end of the tile     8.50 us = 4.1 TB/s
// How it is measured now: bench.py replays ONE HIP graph holding the K launches, which is reproducible to
// +-0.01 us; launched from Python a step costs 8.6 us of host time against ~9 us on the GPU and the number
// wanders between 10.0 and 11.5 us from process to process (tools/stability.py, tools/ab_bench.sh).
// What bounds it (tools/microbench.hip, tools/valu_rate.hip, tools/pmc_hot.sh):
//   * NOT the VALU: with cache-resident inputs H = 8 / 4 / 2 planes take 6.47 / 6.07 / 5.64 us, i.e. all of the
//     arithmetic of 6 planes is worth 0.8 us.  CDNA4 issues a wave64 v_fma_f32 in ~2.6-3 cycles, v_pk_fma_f32
//     in ~4.7 (no gain from packing: a packed version of this kernel was slower), v_add_f32_dpp in ~4.2.
//     SQ counters: a wave lives ~4200 quad-cycles, waits on memory for 46 % of them and issues VALU for 15 %.
//   * the dependent chain launch -> ids -> rows -> score store: empty launch 2.3-2.5 us back to back, ids land
//     ~1.3 us after the wave starts, the first gathered row ~1.5 us later, and the 33.4 MB of rows then stream
//     at > 6 TB/s.  Launches that are allowed to overlap (2-4 streams) reach 5.9 us per launch = 5.9 TB/s:
//     the next batch's head hides under this batch's tail.  A single serialized launch cannot do that.
// Tried and rejected: "lane owns lookup" through an LDS transpose, one wave per SIMD (23 us, LDS
// latency exposed); per-workgroup 2^H-row code table in LDS replacing aggregate + division (13.2 us
// with 1024-thread groups, 17.3 us with 256: build + barrier cost more than they save); v_pk_fma_f32
// math; non-temporal loads; 64 / 128 / 512 / 1024-thread workgroups; user rows requested late.
#include <stdlib.h>

#include "common.hpp"
#include "lsh64_tile.hpp"

namespace mi_oov {

// One tile = 16 lookups of one wave (4 rounds x 4 groups).  FULL tiles (all 16 rows < B) skip every
// tail clamp and liveness test; only the last tile of a launch can be partial.
template <int H, bool SCORE, bool STORE, bool LOOKUP, bool FULL, bool BITS = false>
__device__ __forceinline__ void lsh64_tile(unsigned tile, int l16, int grp, const float4 (&pw)[H], const float4 (&bw)[H],
                                           const int64_t (&idc)[4], unsigned B,
                                           const float* __restrict__ feat, int64_t N,
                                           const float* __restrict__ vtable, int64_t n_vocab,
                                           const float* __restrict__ other, float* __restrict__ score,
                                           float* __restrict__ out, uint8_t* __restrict__ bits = nullptr) {
  static_assert(!BITS || H == 8, "codes are written as one 8-byte word per lookup");
  constexpr int R = 4;
  // rows of the batch are addressed with 32-bit byte offsets from uniform bases (B <= kMaxRows64, checked by
  // the host): global_load/store with an SGPR base + VGPR offset instead of 64-bit VGPR pointers.
  unsigned row[R];
  bool valid[R], oov[R];
#pragma unroll
  for (int r = 0; r < R; ++r) row[r] = tile * (4 * R) + r * 4 + grp;
  // the gathers first (they wait on nothing but the ids), the sequential rows of the other side behind them
  // Issue order matters (A/B under graph replay, +-0.01 us): the R gathers first -- they are the second hop of the
  // ids -> rows chain and the first thing the VALU waits for -- and the sequential rows of the other side behind
  // them (needed only by the score at the end of each round): 9.57 us with the user rows first, 9.10 us this
  // way.  The compiler is free to reorder independent loads, hence the scheduling barrier between the groups.
  // Non-temporal loads (streaming rows, no reuse) changed nothing for the gathers and cost 0.3 us on the user rows;
  // requesting 1 / 2 / 4 of the user rows before the ids have landed costs 0.03 / 0.07 / 4 us (their traffic delays
  // everybody's ids).
  float4 x[R];
  float4 u[SCORE ? R : 1];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    oov[r] = !LOOKUP || idc[r] >= n_vocab;
    valid[r] = oov[r] ? static_cast<uint64_t>(idc[r]) < static_cast<uint64_t>(N) : idc[r] >= 0;
    const float* base = oov[r] ? feat : vtable;
    x[r] = *reinterpret_cast<const float4*>(base + (valid[r] ? idc[r] : 0) * 64 + l16 * 4);
  }
  if (SCORE) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int r = 0; r < R; ++r)
      u[r] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(other) +
                                              (((FULL || row[r] < B) ? row[r] : B - 1) * 256u + l16 * 16u));
  }

  float sc_all = 0.f;
  uint32_t bits_lo = 0, bits_hi = 0;
  // Row stores are held back to the end of the tile when the registers allow it (no user rows in flight): stores
  // count against vmcnt like loads do, so a store issued between rounds makes the wait for the next gathered
  // row also wait for everything issued before that store.
#ifndef MI_EXP_DEFER
#define MI_EXP_DEFER 1
#endif
  constexpr bool kDeferRows = MI_EXP_DEFER && STORE && !SCORE;
  float4 emb_all[kDeferRows ? R : 1];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float cnt = 0.f;
    if constexpr (H == 8) {
      // Eight 16-lane sums for the price of two: the stride-halving tree of row16_sum, with the live copies of
      // each level packed into fewer registers by bank-masked DPP writes (see rows8_sum).  Lane l then holds
      // the sums of the two planes of its bank; the 0/1 bits are handed to the whole row with row_newbcast.
      float p[8];
#pragma unroll
      for (int h = 0; h < 8; ++h) p[h] = dot4_fma(x[r], pw[h], 0.f);
      float t0, t1;
      rows8_sum(p, t0, t1);
      const float b0 = (t0 < 0.f) ? 0.f : 1.f;  // >= 0, +-0 and NaN -> 1 (torch_hash.py:57-59)
      const float b1 = (t1 < 0.f) ? 0.f : 1.f;
      cnt = b0 + b1;                      // this bank's two planes ...
      cnt = cnt + dpp_f32<0x124>(cnt);    // ... + the other banks (small integers: exact in any order)
      cnt = cnt + dpp_f32<0x128>(cnt);
      uint32_t code_lo = 0, code_hi = 0;  // BITS: the u8[8] code row of this lookup as one little-endian word
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        const float bit = bcast8(h, b0, b1);
        if (BITS) {
          const uint32_t one = (__float_as_uint(bit) >> 23) & 1u;  // 1.0f = 0x3F800000, 0.0f = 0
          if (h < 4) code_lo |= one << (8 * h);
          else code_hi |= one << (8 * (h - 4));
        }
        if (SCORE || STORE) {
          acc.x = __builtin_fmaf(bit, bw[h].x, acc.x);
          acc.y = __builtin_fmaf(bit, bw[h].y, acc.y);
          acc.z = __builtin_fmaf(bit, bw[h].z, acc.z);
          acc.w = __builtin_fmaf(bit, bw[h].w, acc.w);
        }
      }
      if (BITS) {
        if (!valid[r]) code_lo = code_hi = 0xFFFFFFFFu;  // invalid id: 0xFF bytes
        if (FULL) {
          if (l16 == r) { bits_lo = code_lo; bits_hi = code_hi; }
        } else if (l16 == 0 && row[r] < B) {
          *reinterpret_cast<uint2*>(bits + static_cast<size_t>(row[r]) * 8u) = make_uint2(code_lo, code_hi);
        }
      }
    } else {
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
    }
    float4 emb = masked_mean(acc, cnt);
    if (LOOKUP && !oov[r]) emb = x[r];
    if (!valid[r]) emb = make_float4(qnan(), qnan(), qnan(), qnan());
    const bool live = FULL || row[r] < B;
    if (STORE && !kDeferRows && live) *reinterpret_cast<float4*>(reinterpret_cast<char*>(out) + (row[r] * 256u + l16 * 16u)) = emb;
    if (STORE && kDeferRows) emb_all[r] = emb;
    if (SCORE) {
      const float s = row16_sum(dot4_muladd(u[r], emb, 0.f));
      if (FULL) {
        if (l16 == r) sc_all = s;  // lane r of the group keeps round r's score: stored once, below
      } else if (l16 == 0 && live) {
        *reinterpret_cast<float*>(reinterpret_cast<char*>(score) + row[r] * 4u) = s;
      }
    }
  }
  if (kDeferRows) {
    // (non-temporal: the rows are written once and not read by this launch -- out of L2 / Infinity Cache they leave
    //  those to the gathers; MI_EXP_NTROWS=0: plain stores)
#ifndef MI_EXP_NTROWS
#define MI_EXP_NTROWS 1
#endif
    typedef float v4f_ __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (FULL || row[r] < B) {
        v4f_* dst = reinterpret_cast<v4f_*>(reinterpret_cast<char*>(out) + (row[r] * 256u + l16 * 16u));
        const v4f_ v = {emb_all[r].x, emb_all[r].y, emb_all[r].z, emb_all[r].w};
        if (MI_EXP_NTROWS) __builtin_nontemporal_store(v, dst);
        else *dst = v;
      }
  }
  if (BITS && FULL && l16 < 4)  // the tile's 16 code rows (128 contiguous bytes) in one store, like the scores
    *reinterpret_cast<uint2*>(bits + static_cast<size_t>(tile * 16u + l16 * 4u + grp) * 8u) = make_uint2(bits_lo, bits_hi);
  // ONE 16-lane store of the tile's 16 contiguous scores instead of four 4-lane stores (-0.4 us per launch:
  // the memory instructions a wave issues are worth more than its arithmetic here)
  if (SCORE && FULL && l16 < 4)
    *reinterpret_cast<float*>(reinterpret_cast<char*>(score) + (tile * 16u + l16 * 4u + grp) * 4u) = sc_all;
}

// The ids of a tile: one 8-byte load per round (4 distinct addresses per instruction).  Fetching all 16 ids
// with ONE instruction and handing them round with row_newbcast was measured too: 8.61 us against 8.52.
__device__ __forceinline__ void load_tile_ids(const int64_t* __restrict__ ids, unsigned tile, unsigned B, int l16, int grp,
                                              int64_t (&idc)[4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned row = tile * 16 + r * 4 + grp;
    // clamped: tail groups recompute the last row (B >= 1 here)
    idc[r] = *reinterpret_cast<const int64_t*>(reinterpret_cast<const char*>(ids) + (row < B ? row : B - 1) * 8u);
  }
}

// 4 waves (256 threads) per workgroup: 1, 2, 8 and 16 waves measured 9.96 / 9.50 / 9.14 / 9.20 us against 9.09.
constexpr int kWpb = 4, kBlk = 64 * kWpb;
template <int H, bool SCORE, bool STORE, bool LOOKUP, bool BITS = false>
__global__ __launch_bounds__(kBlk, 4) void lsh64_kernel(const int64_t* __restrict__ ids, unsigned B,
                                                       const float* __restrict__ feat, int64_t N,
                                                       const float* __restrict__ vtable, int64_t n_vocab,
                                                       const float* __restrict__ planes,
                                                       const float* __restrict__ buckets,
                                                       const float* __restrict__ other,
                                                       float* __restrict__ score, float* __restrict__ out,
                                                       uint8_t* __restrict__ bits) {
  const int lane = threadIdx.x & 63, l16 = lane & 15, grp = lane >> 4, wv = threadIdx.x >> 6;
  const unsigned ntiles = (B + 15) / 16;
  const unsigned nfull = B / 16;
  const unsigned tstep = gridDim.x * kWpb;

  // The ids of the wave's first tile are requested before anything else: their round trip (the first hop of the
  // ids -> rows chain) then overlaps the staging of the weights instead of following it.
  unsigned tile = blockIdx.x * kWpb + wv;
  int64_t idc[4];
  load_tile_ids(ids, tile, B, l16, grp, idc);

  // Plane / bucket slices -> VGPRs through LDS: the workgroup fetches the 2 x H x 256 B once (one 16-B
  // load per thread) and every lane reads its 2 x H float4 back with ds_read_b128, instead of 2 x H
  // global loads per lane (16 KiB of L1 traffic per wave) queued in front of the ids -> rows gathers
  // (-0.4 us in tools/microbench.hip).
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [2][H][64]
  constexpr bool kNeedBuckets = SCORE || STORE;  // codes-only launches have no bucket table
  for (int i = threadIdx.x; i < (kNeedBuckets ? 2 : 1) * H * 16; i += kBlk) {
    const float* src = (i < H * 16) ? planes + i * 4 : buckets + (i - H * 16) * 4;
    *reinterpret_cast<float4*>(sw + i * 4) = *reinterpret_cast<const float4*>(src);
  }
  __syncthreads();
  float4 pw[H], bw[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    pw[h] = *reinterpret_cast<const float4*>(sw + (h * 16 + l16) * 4);
    bw[h] = kNeedBuckets ? *reinterpret_cast<const float4*>(sw + (H * 16 + h * 16 + l16) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }

  while (tile < ntiles) {
    if (tile < nfull)
      lsh64_tile<H, SCORE, STORE, LOOKUP, true, BITS>(tile, l16, grp, pw, bw, idc, B, feat, N, vtable, n_vocab, other, score, out, bits);
    else
      lsh64_tile<H, SCORE, STORE, LOOKUP, false, BITS>(tile, l16, grp, pw, bw, idc, B, feat, N, vtable, n_vocab, other, score, out, bits);
    tile += tstep;
    if (tile < ntiles) {  // only when the grid was capped (B > 16 * 4 * kMaxGrid)
      load_tile_ids(ids, tile, B, l16, grp, idc);
    }
  }
}

// ---- 8 < H <= 64: the same tile, planes taken eight at a time ------------------------------------------------
// The plane / bucket slices no longer fit in VGPRs for the whole kernel (2 x H float4 per lane), so they stay in
// LDS (zero-padded to a multiple of 8 planes) and each group of 8 is read into the same 16 float4 registers per
// tile; the four gathered rows stay in registers across the groups, the bucket-row chain acc = fma(bit_h, W[h],
// acc) runs over h = 0..H-1 in order and the bit counts add up, so the result is bit-identical to the generic
// kernel (lsh_fused_kernel) at roughly half its time (H = 16: 18.2 -> ~10.5 us).  Padding planes project to +0,
// whose bit would be 1: their lanes are masked to 0 in the last group.
template <bool SCORE, bool STORE, bool LOOKUP>
__global__ __launch_bounds__(kBlk, 4) void lsh64g_kernel(const int64_t* __restrict__ ids, unsigned B,
                                                        const float* __restrict__ feat, int64_t N,
                                                        const float* __restrict__ vtable, int64_t n_vocab,
                                                        const float* __restrict__ planes,
                                                        const float* __restrict__ buckets, int H,
                                                        const float* __restrict__ other,
                                                        float* __restrict__ score, float* __restrict__ out) {
  constexpr int R = 4;
  const int lane = threadIdx.x & 63, l16 = lane & 15, grp = lane >> 4, wv = threadIdx.x >> 6;
  const unsigned ntiles = (B + 15) / 16;
  const unsigned tstep = gridDim.x * kWpb;
  const int G = (H + 7) / 8, HP = G * 8;
  unsigned tile = blockIdx.x * kWpb + wv;
  int64_t idc[4];
  load_tile_ids(ids, tile, B, l16, grp, idc);

  extern __shared__ __attribute__((aligned(16))) float sw[];  // [2][HP][64], rows >= H zero
  for (int i = threadIdx.x; i < 2 * HP * 16; i += kBlk) {
    const int h = (i < HP * 16) ? i / 16 : (i - HP * 16) / 16;
    const float* src = (i < HP * 16) ? planes + i * 4 : buckets + (i - HP * 16) * 4;
    *reinterpret_cast<float4*>(sw + i * 4) = (h < H) ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  // which planes of the last group exist: the lane's bank holds planes {0,2,1,3}[bank] (t0) and 4 + that (t1)
  const int bank = l16 >> 2;
  const int pl = ((bank & 1) << 1) | (bank >> 1);
  const int hl = H - (G - 1) * 8;  // 1..8 planes in the last group
  const float last0 = (pl < hl) ? 1.f : 0.f, last1 = (4 + pl < hl) ? 1.f : 0.f;

  while (tile < ntiles) {
    unsigned row[R];
    bool valid[R], oov[R];
    float4 x[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      row[r] = tile * (4 * R) + r * 4 + grp;
      oov[r] = !LOOKUP || idc[r] >= n_vocab;
      valid[r] = oov[r] ? static_cast<uint64_t>(idc[r]) < static_cast<uint64_t>(N) : idc[r] >= 0;
      const float* base = oov[r] ? feat : vtable;
      x[r] = *reinterpret_cast<const float4*>(base + (valid[r] ? idc[r] : 0) * 64 + l16 * 4);
    }
    float4 acc[R];
    float cnt[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      cnt[r] = 0.f;
    }
    for (int g = 0; g < G; ++g) {
      float4 pw[8], bw[8];
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        pw[h] = *reinterpret_cast<const float4*>(sw + ((g * 8 + h) * 16 + l16) * 4);
        bw[h] = *reinterpret_cast<const float4*>(sw + ((HP + g * 8 + h) * 16 + l16) * 4);
      }
      const float m0 = (g == G - 1) ? last0 : 1.f, m1 = (g == G - 1) ? last1 : 1.f;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float p[8];
#pragma unroll
        for (int h = 0; h < 8; ++h) p[h] = dot4_fma(x[r], pw[h], 0.f);
        float t0, t1;
        rows8_sum(p, t0, t1);
        const float b0 = ((t0 < 0.f) ? 0.f : 1.f) * m0;  // >= 0, +-0 and NaN -> 1 (torch_hash.py:57-59)
        const float b1 = ((t1 < 0.f) ? 0.f : 1.f) * m1;
        cnt[r] = cnt[r] + (b0 + b1);  // per-bank share; the banks are added up once, after the last group
#pragma unroll
        for (int h = 0; h < 8; ++h) {
          const float bit = bcast8(h, b0, b1);
          acc[r].x = __builtin_fmaf(bit, bw[h].x, acc[r].x);
          acc[r].y = __builtin_fmaf(bit, bw[h].y, acc[r].y);
          acc[r].z = __builtin_fmaf(bit, bw[h].z, acc[r].z);
          acc[r].w = __builtin_fmaf(bit, bw[h].w, acc[r].w);
        }
      }
    }
    float sc_all = 0.f;
    const bool full = tile * 16 + 16 <= B;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float c = cnt[r];
      c = c + dpp_f32<0x124>(c);
      c = c + dpp_f32<0x128>(c);
      float4 emb = masked_mean(acc[r], c);
      if (LOOKUP && !oov[r]) emb = x[r];
      if (!valid[r]) emb = make_float4(qnan(), qnan(), qnan(), qnan());
      const bool live = row[r] < B;
      if (STORE && live) *reinterpret_cast<float4*>(reinterpret_cast<char*>(out) + (row[r] * 256u + l16 * 16u)) = emb;
      if (SCORE) {
        const float4 u = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(other) +
                                                          ((live ? row[r] : B - 1) * 256u + l16 * 16u));
        const float sdot = row16_sum(dot4_muladd(u, emb, 0.f));
        if (full) {
          if (l16 == r) sc_all = sdot;
        } else if (l16 == 0 && live) {
          *reinterpret_cast<float*>(reinterpret_cast<char*>(score) + row[r] * 4u) = sdot;
        }
      }
    }
    if (SCORE && full && l16 < 4)
      *reinterpret_cast<float*>(reinterpret_cast<char*>(score) + (tile * 16u + l16 * 4u + grp) * 4u) = sc_all;
    tile += tstep;
    if (tile < ntiles) load_tile_ids(ids, tile, B, l16, grp, idc);
  }
}

template <bool SCORE, bool STORE, bool LOOKUP>
static int launch64g(const int64_t* ids, int64_t B, const float* feat, int64_t N, const float* vtable, int64_t n_vocab,
                     const float* planes, const float* buckets, int H, const float* other, float* score, float* out,
                     hipStream_t st) {
  constexpr int64_t kMaxRows64 = int64_t(1) << 23;
  const int HP = (H + 7) / 8 * 8;
  for (int64_t b0 = 0; b0 < B; b0 += kMaxRows64) {
    const int64_t nb = (B - b0 < kMaxRows64) ? B - b0 : kMaxRows64;
    const int grid = grid_for(nb, 16 * kWpb);
    hipLaunchKernelGGL((lsh64g_kernel<SCORE, STORE, LOOKUP>), dim3(grid), dim3(kBlk), 2 * HP * 64 * sizeof(float), st,
                       ids + b0, static_cast<unsigned>(nb), feat, N, vtable, n_vocab, planes, buckets, H,
                       other ? other + b0 * 64 : nullptr, score ? score + b0 : nullptr, out ? out + b0 * 64 : nullptr);
    if (int rc = check_launch()) return rc;
  }
  return MI_OOV_OK;
}

// ---- slsh on the hot tile: F = 64, up to 32 planes, D = 64 or 128 -------------------------------------------------
// single_lsh_embedder.py:82-109: idx = (bits_req + popcount) % n_buckets, out = buckets[idx].  Same tile, same
// bank-masked reduce as above with the planes taken eight at a time from LDS; only the per-bank bit counts are
// kept.  The reference's arithmetic reaches only the H + 1 bucket rows (H + c) % n_buckets, c = 0..H, however large the
// bucket table is (n_buckets = 8 only ever yields ids 3..6): the workgroup copies those rows into
// LDS once (<= 33 rows, 17 KiB at D = 128) and a lookup's bucket row is a conflict-free ds_read_b128 at its count -- a
// row is a multiple of 256 B = all 64 banks, so the bank of a read depends on the lane only -- instead of a third
// dependent hop ids -> feature row -> bucket row through HBM (n_buckets = N: 12.95 -> see DESIGN.md section 5; the
// rows are copies, so the output is the same bits).  The tile's 16 bucket ids leave in one store, the output rows with
// non-temporal stores.  TAB: ids_src / out_src / idx_src are DEVICE arrays of K pointers (mi_oov_slsh_embed_multi).
// Pipeline (round 4): a wave that has more than one tile keeps the gathered rows of its NEXT tile and the ids of the one after
// that in flight while it reduces the current one (ids(i+2) requested, rows(i+1) requested, tile i reduced and stored, in
// that issue order -- the scheme of lsh64p.hip): with one tile's rows in flight per wave, 16 waves per CU hold 4 MB of
// requests chip-wide, less than the ~10 MB that 5 TB/s x 2 us of latency want.  20 x 65536 lookups, same box, before -> after:
// 24 planes, D = 64 (n_buckets = N): 146.7 -> 129.6 us; 10 planes, D = 128: 185 -> 176 us (0.72 of the HBM peak on moved
// bytes); a single 65536-lookup launch, where a wave has one tile, is unchanged (11.4 -> 11.8, 11.8 -> 11.9).  At 24 planes
// the projections are ~660 vector instructions per tile, ~88 us of issue per launch next to ~106 us of row traffic.  The steady-state loop holds no conditional
// memory instruction (a load or store issued on one path only turns the waits behind it into vmcnt(0)): rows of a partial
// tile beyond the batch are stored ONTO its last row -- their lanes were given the last row's id by the clamped id loads, so
// they hold the same bits -- and what is stored at all is a template parameter (ROWS, IDX).  The ids of a tile are dead once
// its rows are requested (validity is kept as a 4-bit mask), which is what lets the loop rotate without moving registers
// that are still in flight.
template <int DCH, bool TAB, bool ROWS, bool IDX>
__global__ __launch_bounds__(kBlk, 4) void slsh64_kernel(const void* __restrict__ ids_src, unsigned B, unsigned K,
                                                        const float* __restrict__ feat, int64_t N,
                                                        const float* __restrict__ planes, int H,
                                                        const float* __restrict__ buckets, int64_t n_buckets,
                                                        void* __restrict__ out_src, void* __restrict__ idx_src) {
  constexpr int R = 4;
  typedef float v4f_ __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, l16 = lane & 15, grp = lane >> 4;
  const unsigned wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned tpb = (B + 15) / 16;
  const unsigned ntiles = tpb * (TAB ? K : 1u);
  const unsigned tstep = gridDim.x * kWpb;
  const int G = (H + 7) / 8, HP = G * 8;
  struct Tile {
    unsigned batch, local;
  };
  auto tile_of = [&](unsigned t) {
    Tile p;
    p.batch = TAB ? t / tpb : 0u;
    p.local = TAB ? t - p.batch * tpb : t;
    return p;
  };
  // (pointers out of the tables are cast to the GLOBAL address space: through a generic pointer the compiler emits flat_load /
  //  flat_store, which count against lgkmcnt as well as vmcnt and may return out of order -- every wait becomes vmcnt(0))
  typedef const int64_t __attribute__((address_space(1))) * gptr_ci64;
  typedef int64_t __attribute__((address_space(1))) * gptr_i64w;
  typedef v4f_ __attribute__((address_space(1))) * gptr_v4w;
  typedef const v4f_ __attribute__((address_space(1))) * gptr_cv4;
  auto load_ids = [&](unsigned t, int64_t (&id)[4]) {
    const Tile p = tile_of(t);
    gptr_ci64 src = TAB ? (gptr_ci64) reinterpret_cast<const int64_t* const*>(ids_src)[p.batch] : (gptr_ci64)ids_src;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned row = p.local * 16u + r * 4u + grp;
      id[r] = src[row < B ? row : B - 1u];  // clamped: tail groups recompute the last row (B >= 1 here)
    }
  };
  // the 4 gathered rows of a tile; returns which of its 4 rounds address a row (the ids are not needed after this)
  auto gather = [&](const int64_t (&id)[4], float4 (&x)[R]) -> unsigned {
    unsigned ok = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool v = static_cast<uint64_t>(id[r]) < static_cast<uint64_t>(N);
      ok |= v ? (1u << r) : 0u;
      const v4f_ g = *(gptr_cv4)(feat + (v ? id[r] : 0) * 64 + l16 * 4);
      x[r] = make_float4(g.x, g.y, g.z, g.w);
    }
    return ok;
  };
  unsigned ta = blockIdx.x * kWpb + wv, tb = ta + tstep;
  int64_t ida[4] = {0, 0, 0, 0}, idb[4] = {0, 0, 0, 0};
  if (ta < ntiles) load_ids(ta, ida);
  if (tb < ntiles) load_ids(tb, idb);

  extern __shared__ __attribute__((aligned(16))) float sw[];  // [HP][64] planes (rows >= H zero), [H + 1][64 DCH] reachable bucket rows
  float* srows = sw + HP * 64;
  {
    // every staging load first, then the LDS stores: one memory round trip in front of the barrier, not two
    constexpr int kPl = (32 * 16 + kBlk - 1) / kBlk, kRw = (33 * 16 * DCH + kBlk - 1) / kBlk;
    float4 pv[kPl], rv[kRw];
#pragma unroll
    for (int q = 0; q < kPl; ++q) {
      const int i = q * kBlk + static_cast<int>(threadIdx.x);
      pv[q] = (i < H * 16) ? *reinterpret_cast<const float4*>(planes + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < kRw; ++q) {
      const int i = q * kBlk + static_cast<int>(threadIdx.x);
      rv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ROWS && i < (H + 1) * 16 * DCH) {
        const int c = i / (16 * DCH), e = i - c * (16 * DCH);
        const int64_t v = H + c;
        const int64_t b = v < n_buckets ? v : static_cast<int64_t>(static_cast<uint32_t>(v) % static_cast<uint32_t>(n_buckets));
        rv[q] = *reinterpret_cast<const float4*>(buckets + b * (64 * DCH) + e * 4);
      }
    }
#pragma unroll
    for (int q = 0; q < kPl; ++q) {
      const int i = q * kBlk + static_cast<int>(threadIdx.x);
      if (i < HP * 16) *reinterpret_cast<float4*>(sw + i * 4) = pv[q];
    }
#pragma unroll
    for (int q = 0; q < kRw; ++q) {
      const int i = q * kBlk + static_cast<int>(threadIdx.x);
      if (ROWS && i < (H + 1) * 16 * DCH) *reinterpret_cast<float4*>(srows + i * 4) = rv[q];
    }
  }
  __syncthreads();
  if (ta >= ntiles) return;
  const int bank = l16 >> 2;
  const int pl = ((bank & 1) << 1) | (bank >> 1);
  const int hl = H - (G - 1) * 8;
  const float last0 = (pl < hl) ? 1.f : 0.f, last1 = (4 + pl < hl) ? 1.f : 0.f;

  // one tile whose rows were requested an iteration ago: popcounts, bucket ids, stores
  auto process = [&](unsigned t, unsigned ok, const float4 (&x)[R]) {
    const Tile p = tile_of(t);
    float cnt[R];
#pragma unroll
    for (int r = 0; r < R; ++r) cnt[r] = 0.f;
    for (int g = 0; g < G; ++g) {
      float4 pw[8];
#pragma unroll
      for (int h = 0; h < 8; ++h) pw[h] = *reinterpret_cast<const float4*>(sw + ((g * 8 + h) * 16 + l16) * 4);
      const float m0 = (g == G - 1) ? last0 : 1.f, m1 = (g == G - 1) ? last1 : 1.f;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float q[8];
#pragma unroll
        for (int h = 0; h < 8; ++h) q[h] = dot4_fma(x[r], pw[h], 0.f);
        float t0, t1;
        rows8_sum(q, t0, t1);
        cnt[r] = cnt[r] + (((t0 < 0.f) ? 0.f : 1.f) * m0 + ((t1 < 0.f) ? 0.f : 1.f) * m1);
      }
    }
    // (2 ** bits).sum(1) = one or two per plane = H + popcount (single_lsh_embedder.py:86)
    int64_t bkt[R];
    int pc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float c = cnt[r];
      c = c + dpp_f32<0x124>(c);
      c = c + dpp_f32<0x128>(c);
      pc[r] = static_cast<int>(c);  // 0..H
      const int64_t v = H + pc[r];  // <= 64
      bkt[r] = ((ok >> r) & 1u) ? (v < n_buckets ? v : static_cast<int64_t>(static_cast<uint32_t>(v) % static_cast<uint32_t>(n_buckets)))
                                : -1;
    }
    if constexpr (ROWS) {
      float* out = TAB ? reinterpret_cast<float* const*>(out_src)[p.batch] : static_cast<float*>(out_src);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        unsigned row = p.local * (4 * R) + r * 4 + grp;
        row = row < B ? row : B - 1u;  // (a tail row lands on the batch's last row with that row's own value)
#pragma unroll
        for (int c = 0; c < DCH; ++c) {
          const float4 v = *reinterpret_cast<const float4*>(srows + ((pc[r] * DCH + c) * 16 + l16) * 4);
          const bool good = (ok >> r) & 1u;
          __builtin_nontemporal_store(v4f_{good ? v.x : qnan(), good ? v.y : qnan(), good ? v.z : qnan(), good ? v.w : qnan()},
                                      (gptr_v4w)(out + static_cast<size_t>(row) * (64 * DCH) + c * 64 + l16 * 4));
        }
      }
    }
    if constexpr (IDX) {
      int64_t* idx = TAB ? reinterpret_cast<int64_t* const*>(idx_src)[p.batch] : static_cast<int64_t*>(idx_src);
      // the tile's 16 bucket ids in one store: lane r of a group keeps round r's
      int64_t mine = bkt[0];
      if (l16 == 1) mine = bkt[1];
      if (l16 == 2) mine = bkt[2];
      if (l16 == 3) mine = bkt[3];
      unsigned row = p.local * 16u + (l16 & 3) * 4u + grp;
      row = row < B ? row : B - 1u;
      if (l16 < 4) ((gptr_i64w)idx)[row] = mine;
    }
  };

  float4 xa[R], xb[R];
  unsigned oka = gather(ida, xa), okb = 0;
  // steady state: the two tiles after b exist
  while (tb + 2 * tstep < ntiles) {
    const unsigned tn = tb + tstep, tm = tn + tstep;
    int64_t idn[4];
    load_ids(tn, idn);
    asm volatile("" ::: "memory");
    okb = gather(idb, xb);
    asm volatile("" ::: "memory");
    process(ta, oka, xa);
    load_ids(tm, idb);
    asm volatile("" ::: "memory");
    oka = gather(idn, xa);
    asm volatile("" ::: "memory");
    process(tb, okb, xb);
    ta = tn;
    tb = tm;
  }
  // drain: a (rows in flight), then b and the tile after it where they exist
  const bool has_b = tb < ntiles, has_n = has_b && tb + tstep < ntiles;
  int64_t idn[4] = {0, 0, 0, 0};
  if (has_n) load_ids(tb + tstep, idn);
  if (has_b) okb = gather(idb, xb);
  process(ta, oka, xa);
  if (has_b) {
    if (has_n) oka = gather(idn, xa);
    process(tb, okb, xb);
    if (has_n) process(tb + tstep, oka, xa);
  }
}

template <bool TAB>
static void slsh64_launch(int dch, int grid, size_t lds, hipStream_t st, const void* ids, unsigned B, unsigned K, const float* feat,
                          int64_t N, const float* planes, int H, const float* buckets, int64_t n_buckets, void* out, void* idx) {
#define MI_SLSH_GO(DC, RW, IX) \
  hipLaunchKernelGGL((slsh64_kernel<DC, TAB, RW, IX>), dim3(grid), dim3(kBlk), lds, st, ids, B, K, feat, N, planes, H, buckets, n_buckets, out, idx)
  if (out && idx) {
    if (dch == 2) MI_SLSH_GO(2, true, true); else MI_SLSH_GO(1, true, true);
  } else if (out) {
    if (dch == 2) MI_SLSH_GO(2, true, false); else MI_SLSH_GO(1, true, false);
  } else {
    MI_SLSH_GO(1, false, true);
  }
#undef MI_SLSH_GO
}

// Host entry used by mi_oov_slsh_embed / mi_oov_slsh_embed_multi (lsh.hip) when F = 64, H <= 32 and D is 64 or 128 (or no
// rows are wanted).  tab: ids / out / idx are device arrays of K pointers.
int launch_slsh64(const void* ids, int64_t B, int64_t K, bool tab, const float* feat, int64_t N, const float* planes, int H,
                  const float* buckets, int64_t n_buckets, int64_t D, void* out, void* idx, hipStream_t st) {
  constexpr int64_t kMaxRows = int64_t(1) << 22;  // 32-bit row arithmetic inside the kernel
  const int HP = (H + 7) / 8 * 8;
  const int dch = (out && D == 128) ? 2 : 1;
  if (!out && !idx) return MI_OOV_OK;  // nothing asked for
  const size_t lds = (static_cast<size_t>(HP) * 64 + (out ? static_cast<size_t>(H + 1) * 64 * dch : 0)) * sizeof(float);
  if (tab) {
    if (B > kMaxRows || K * ((B + 15) / 16) >= (int64_t(1) << 31)) return MI_OOV_ERR_SHAPE;
    // (2048 workgroups of 4 waves, ~10 tiles per wave at 20 x 65536 lookups: 512 ... 2048 measured within 5 %)
    const int grid = grid_for(K * ((B + 15) / 16) * 16, 16 * kWpb);
    slsh64_launch<true>(dch, grid, lds, st, ids, static_cast<unsigned>(B), static_cast<unsigned>(K), feat, N, planes, H, buckets, n_buckets, out, idx);
    return check_launch();
  }
  for (int64_t b0 = 0; b0 < B; b0 += kMaxRows) {
    const int64_t nb = (B - b0 < kMaxRows) ? B - b0 : kMaxRows;
    const int grid = grid_for(nb, 16 * kWpb);
    const void* ids_b = static_cast<const int64_t*>(ids) + b0;
    void* out_b = out ? static_cast<void*>(static_cast<float*>(out) + b0 * D) : nullptr;
    void* idx_b = idx ? static_cast<void*>(static_cast<int64_t*>(idx) + b0) : nullptr;
    slsh64_launch<false>(dch, grid, lds, st, ids_b, static_cast<unsigned>(nb), 1u, feat, N, planes, H, buckets, n_buckets, out_b, idx_b);
    if (int rc = check_launch()) return rc;
  }
  return MI_OOV_OK;
}

template <int H, bool SCORE, bool STORE, bool LOOKUP, bool BITS = false>
static int launch64(const int64_t* ids, int64_t B, const float* feat, int64_t N, const float* vtable, int64_t n_vocab,
                    const float* planes, const float* buckets, const float* other, float* score, float* out,
                    hipStream_t st, uint8_t* bits = nullptr) {
  // the kernel addresses batch rows with 32-bit byte offsets: launches of at most kMaxRows64 lookups
  constexpr int64_t kMaxRows64 = int64_t(1) << 23;  // x 256 B = 2 GiB
  for (int64_t b0 = 0; b0 < B; b0 += kMaxRows64) {
    const int64_t nb = (B - b0 < kMaxRows64) ? B - b0 : kMaxRows64;
    const int grid = grid_for(nb, 16 * kWpb);  // kWpb waves x 16 lookups per workgroup pass
    hipLaunchKernelGGL((lsh64_kernel<H, SCORE, STORE, LOOKUP, BITS>), dim3(grid), dim3(kBlk), 2 * H * 64 * sizeof(float), st,
                       ids + b0, static_cast<unsigned>(nb), feat, N, vtable, n_vocab, planes, buckets,
                       other ? other + b0 * 64 : nullptr, score ? score + b0 : nullptr, out ? out + b0 * 64 : nullptr,
                       bits ? bits + b0 * H : nullptr);
    if (int rc = check_launch()) return rc;
  }
  return MI_OOV_OK;
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

int launch_lsh64_codes_persistent(const int64_t* ids, int64_t B, const float* feat, int64_t N, const float* planes,
                                  uint8_t* bits, hipStream_t st);  // lsh64p.hip
int launch_lsh64_persistent_single(const int64_t* ids, int64_t B, const float* feat, int64_t N, const float* vtable,
                                   int64_t n_vocab, const float* planes, int H, const float* buckets, const float* other,
                                   float* score, float* out, hipStream_t st);  // lsh64p.hip
constexpr int64_t kCodesMinB = 262144;
// One batch of scores from this many lookups on: the persistent, software-pipelined kernel (a wave then walks several
// tile pairs; below, the one-tile-per-wave launch of this file is faster).  MI_OOV_PERSIST_MIN_B: developer knob.
// Round 4 sweep (tools/single_sizes.py under MI_OOV_PERSIST_MIN_B = 65536 ... 524288, us per call, this kernel / the persistent
// one): 65536 lookups 8.9 / 10.9, 131072: 14.7 / 16.5, 262144: 28.5 / 27.4, 524288: 50.7 / 49.9 -- the persistent kernel's
// head (pointer hop, table load, first burst) is only paid back from a quarter of a million lookups on.
static int64_t persist_min_b() {
  static const int64_t v = env_knob("MI_OOV_PERSIST_MIN_B", 262144, 1, int64_t(1) << 40);
  return v;
}
static bool codes_persistent_enabled() {
  static const bool on = env_knob("MI_OOV_CODES_PERSISTENT", 1, 0, 1) != 0;  // developer A/B knob; default on
  return on;
}

// Host entry used by run_lsh (lsh.hip) when the shape qualifies: F = D = 64, 1 <= H <= 8; `bits` (the u8[B,H] codes,
// what the training forward keeps for its backward) only with H == 8, no score and no in-vocabulary table.
int launch_lsh64(const int64_t* ids, int64_t B, const float* feat, int64_t N, const float* vtable, int64_t n_vocab,
                 const float* planes, int H, const float* buckets, const float* other, float* score, float* out,
                 hipStream_t st, uint8_t* bits) {
  if (H > 8) {  // 9..64 planes: groups of eight through LDS (no codes output on this path)
    if (H > 64 || bits) return MI_OOV_ERR_SHAPE;
#define MI_GOG(S, T, L) return launch64g<S, T, L>(ids, B, feat, N, vtable, n_vocab, planes, buckets, H, other, score, out, st)
    if (vtable) {
      if (score && out) MI_GOG(true, true, true);
      if (score) MI_GOG(true, false, true);
      MI_GOG(false, true, true);
    }
    if (score && out) MI_GOG(true, true, false);
    if (score) MI_GOG(true, false, false);
    MI_GOG(false, true, false);
#undef MI_GOG
  }
  if (bits) {
    if (H != 8 || score || vtable) return MI_OOV_ERR_SHAPE;
    // codes only, many lookups (the owner side of a sharded table answers a million per exchange): the persistent,
    // software-pipelined kernel of lsh64p.hip.  Below kCodesMinB a wave has one or two tiles and nothing to pipeline.
    if (!out && B >= kCodesMinB && codes_persistent_enabled()) return launch_lsh64_codes_persistent(ids, B, feat, N, planes, bits, st);
    if (out) return launch64<8, false, true, false, true>(ids, B, feat, N, vtable, n_vocab, planes, buckets, other, score, out, st, bits);
    return launch64<8, false, false, false, true>(ids, B, feat, N, vtable, n_vocab, planes, buckets, other, score, out, st, bits);
  }
  // one large batch of SCORES: the persistent kernel from kPersistMinB lookups on (524288: 49.3 vs 52.0 us, 4 M: 352 vs
  // 401 us).  Rows: the grid-stride launch below keeps pace with it at every size (4 M rows: 391 vs 392 us) and stays.
  if (score && !out && B >= persist_min_b())
    return launch_lsh64_persistent_single(ids, B, feat, N, vtable, n_vocab, planes, H, buckets, other, score, out, st);
  switch (H) {
#define MI_CASE(HV) \
  case HV: return launch64_h<HV>(ids, B, feat, N, vtable, n_vocab, planes, buckets, other, score, out, st);
    MI_CASE(1) MI_CASE(2) MI_CASE(3) MI_CASE(4) MI_CASE(5) MI_CASE(6) MI_CASE(7) MI_CASE(8)
#undef MI_CASE
    default: return MI_OOV_ERR_SHAPE;
  }
}

}  // namespace mi_oov
