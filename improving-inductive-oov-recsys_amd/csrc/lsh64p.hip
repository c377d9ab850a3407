// lsh, hot shape F = D = 64, H <= 8: the PERSISTENT, software-pipelined form of the fused hash-gather-aggregate.
//
//   lsh64_persistent_kernel<H, MODE_SCORE, TAB = true>   K queued batches, fused with the pairwise score
//       (mi_oov_lsh_embed_score_multi).  Replaces K back-to-back launches of lsh64_kernel<H, SCORE> (lsh64.hip), i.e.
//       K times the reference's per-batch op sequence LSHInductiveEmbedder.embed_item_ids
//       (lsh_embedder.py:161-179) + BPR.predict (bpr.py:145-149).
//   lsh64_persistent_kernel<8, MODE_CODES, TAB = false>  one large batch, codes only (TorchLSHash.hash_points,
//       torch_hash.py:55-60, over gathered rows): what mi_oov_lsh_embed(bits only) runs for B >= kCodesMinB, e.g. the
//       owner side of a row-sharded table (sharded.py), which answers a million ids per exchange.
//   lsh64_persistent_kernel<8, MODE_FROM_CODES, TAB = false>  the REQUESTER side of that exchange for D = 64: the
//       "feature row" of a lookup is the 8-byte code an owner sent back (at row slot[b] of the answer array), the rest
//       -- table of aggregates, sequential rows of the other side, score -- is the score mode
//       (mi_oov_lsh_codes_embed with score only; csrc/exchange.hip holds the general-shape kernel).
//
// A serialised launch of 65536 lookups spends ~2 of its 8.5 us in a head nothing overlaps (launch ramp, ids hop,
// first gathered row) and a tail; the row traffic itself moves at ~5.2 TB/s.  Here the waves
// stay resident and walk the tiles of ALL batches (tile t of the launch = 16 consecutive lookups of batch
// t / tiles_per_batch), software-pipelined three deep per wave:
//     ids of tile i+2   requested   (one 8-byte load per lane, handed round with row_newbcast)
//     rows of tile i+1  requested   (4 gathered feature rows + 4 sequential rows of the other side per lane)
//     tile i            reduced, scored, stored
// in that ISSUE ORDER: vmcnt retires in order, so a wait for the ids of tile i+2 must not have the rows of tile
// i+1 queued in front of it.  The steady-state loop holds no conditional load (an s_waitcnt count is an immediate: a
// load issued on one path only forces vmcnt(0) at the join); waves leave it through drain blocks.
//
// The aggregate (bits @ W) / popcount takes only 2^H values, so the workgroup computes all of them once per launch
// into LDS (code c, lane slice l -> float4 at (c * 16 + l) * 16 B; 64 KiB at H = 8) with exactly the per-lookup
// arithmetic -- fmaf chain over the bucket rows in plane order from +0, one correctly rounded division (code 0 ->
// 0/0 -> the reference's NaN row) -- and a lookup becomes one ds_read_b128 at its code: same bits, no bucket rows in
// VGPRs, ~45 % fewer VALU instructions per lookup.  A row of the table is 256 B = all 64 banks, so the bank of a read
// depends on the lane only: conflict-free whatever the four codes of a wave are.  Projections: dot4_fma chain per
// lane + the bank-masked 16-lane tree of lsh64_tile.hpp -- the same additions in the same order as the per-batch
// kernel, so results are bit-identical (tests/test_gpu_parity.py).
//
// Measured on MI355X (tools/multi_bench.cpp: N = 10 M x 64, B = 65536, H = 8, 512 distinct id batches, 1 GiB ring of
// user rows, so nothing is re-read out of the 256 MiB Infinity Cache; gpurun_out/r02_multi_ab*.log), per batch:
//   K single launches (lsh64_kernel)                                   9.05 us = 3.85 TB/s of 532 B/lookup
//   this kernel, weights in VGPRs, 3 waves/SIMD x 4-wave workgroups     6.11 us
//   + aggregate from the 2^H-row table in LDS (12 waves / CU)           5.86 us
//   + 2 waves/SIMD, one 8-wave workgroup per CU                         5.73 us  (1-wave-per-SIMD: 6.13)
//   + non-temporal loads for the sequential rows of the other side      5.45 us = 6.40 TB/s = 0.80 of the 8 TB/s peak
//     (read once, 16.7 MB per batch: kept out of the caches they leave L2 / Infinity Cache to the gathers; the
//     same hint on the gathered rows costs 0.15 us)
//   K = 20 batches per launch 5.6 us, K = 4: 6.5 us (the launch's head and tail are paid once per launch).
//
// Scheduling: the first quarter of a launch's tiles is dealt statically (wave g of G takes tiles g, g+G, ...: no
// counter, and the sequential rows of the other side are swept by all waves as one moving front), the rest comes
// from a pool of tile pairs the waves draw tickets for (scalar atomics; see "schedule" in the kernel) -- with every
// tile dealt statically the waves of a 20-batch launch ended between 86 and 110 us after its start, with the pool
// between 98 and 106.  Per batch, same harness: K = 64: 5.39 -> 5.16 us (0.845 of peak), K = 20: 5.5 -> 5.4,
// launches of fewer than 12 pairs per wave keep the fixed order (the pool's own costs outweigh it there).
// Every wave's loop is bounded by the tile count (static rounds) or ends at the first ticket past the pool's last
// pair, so the grid always drains.
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "common.hpp"
#include "lsh64_tile.hpp"

namespace mi_oov {

// Developer knobs (tools/multi_bench.sh builds variants): MI_PW = waves per SIMD the register allocation is bounded
// for; MI_PWPB = waves per workgroup; MI_PNT_X / MI_PNT_U = non-temporal loads for the gathered / sequential rows.
#ifndef MI_PSTATIC
#define MI_PSTATIC 25  // per cent of a launch's tiles dealt statically (the rest: the ticket pool)
#endif
#ifndef MI_PW
#define MI_PW 2
#endif
#ifndef MI_PWPB
#define MI_PWPB 8
#endif
#ifndef MI_PNT_X
#define MI_PNT_X 0
#endif
#ifndef MI_PNT_U
#define MI_PNT_U 1
#endif
#ifndef MI_PROWS_WAIT
#define MI_PROWS_WAIT 1  // rows / mean modes: s_waitcnt vmcnt(n) in front of each reduction, 0 = none (see step_pair)
#endif
#ifndef MI_PSCORE_WAIT
#define MI_PSCORE_WAIT 0
#endif
#ifndef MI_PROWS_WG
#define MI_PROWS_WG 1
#endif
#ifndef MI_PNT_O
#define MI_PNT_O 1  // non-temporal stores for the rows of the rows mode
#endif
constexpr int kPWpb = MI_PWPB, kPBlk = 64 * kPWpb;
constexpr int kModeScore = 0, kModeCodes = 1, kModeFromCodes = 2, kModeRows = 3, kModeMean = 4;

typedef const int64_t __attribute__((address_space(1))) * gptr_i64;
typedef float __attribute__((address_space(1))) * gptr_f32;
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef v2u __attribute__((address_space(1))) * gptr_u2;
typedef const int32_t __attribute__((address_space(1))) * gptr_i32;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const v4f __attribute__((address_space(1))) * gptr_cv4;
typedef v4f __attribute__((address_space(1))) * gptr_v4;
// 16-byte load through a pointer KNOWN to be global: a pointer read from a table in memory is generic to the
// compiler, and a flat_load counts against lgkmcnt as well as vmcnt, which would serialise the pipeline below
template <bool NT = false>
__device__ __forceinline__ float4 gload4(const void* p) {
  const v4f v = NT ? __builtin_nontemporal_load((gptr_cv4)p) : *(gptr_cv4)p;
  return make_float4(v.x, v.y, v.z, v.w);
}

struct TilePos {
  unsigned batch, local;  // wave-uniform
};

template <int R>
__device__ __forceinline__ int64_t bcast_id(int64_t v) {  // lane R of every 16-lane row -> the whole row
  const int lo = __builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x150 + R, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, static_cast<int>(v >> 32), 0x150 + R, 0xF, 0xF, true);
  return (static_cast<int64_t>(hi) << 32) | static_cast<uint32_t>(lo);
}

__device__ __forceinline__ void round_ids(int64_t idv, int64_t (&id)[4]) {
  id[0] = bcast_id<0>(idv);
  id[1] = bcast_id<1>(idv);
  id[2] = bcast_id<2>(idv);
  id[3] = bcast_id<3>(idv);
}

// The 2^H-row table of aggregates as a function of the bucket rows alone (sbuckets: [H][64] floats in LDS).  Stage 1 of
// every table-using launch, or -- mi_oov_lsh_table_prepare -- made once per bucket table and then LOADED by the launches
// (64 KiB out of L2 instead of ~4 us of dependent fmaf chains at the head of every launch).  The chain of a code runs
// over the planes in increasing order, so all codes with the same LOW bits share its first steps: a thread takes one lane
// slice and one pattern of the low kLow planes, runs that prefix once, and finishes the 2^(H - kLow) codes above it --
// every fmaf(bit, w, acc) of the per-lookup chain is still executed (zero bits included: a non-finite bucket weight must
// poison the row exactly as it does there), in the same order, so the entries are the bits the per-lookup code produces
// (stamps: 5.5 us with one independent 8-step chain per entry, 3.7 with the shared prefix).
template <int H, typename Dst>
__device__ __forceinline__ void build_code_table(const float* sbuckets, Dst dst, int tid, int nthreads) {
  constexpr int kLow = H < 5 ? H : 5, kHigh = H - kLow;
  for (int i = tid; i < (16 << kLow); i += nthreads) {
    const int m = i >> 4, l = i & 15;
    float4 w[H];
#pragma unroll
    for (int h = 0; h < H; ++h) w[h] = *reinterpret_cast<const float4*>(sbuckets + (h * 16 + l) * 4);
    float4 pre = make_float4(0.f, 0.f, 0.f, 0.f);
    float cpre = 0.f;
#pragma unroll
    for (int h = 0; h < kLow; ++h) {
      const float bit = ((m >> h) & 1) ? 1.f : 0.f;
      cpre = cpre + bit;
      pre.x = __builtin_fmaf(bit, w[h].x, pre.x);
      pre.y = __builtin_fmaf(bit, w[h].y, pre.y);
      pre.z = __builtin_fmaf(bit, w[h].z, pre.z);
      pre.w = __builtin_fmaf(bit, w[h].w, pre.w);
    }
#pragma unroll
    for (int hi = 0; hi < (1 << kHigh); ++hi) {
      float4 acc = pre;
      float cnt = cpre;
#pragma unroll
      for (int h = kLow; h < H; ++h) {
        const float bit = ((hi >> (h - kLow)) & 1) ? 1.f : 0.f;
        cnt = cnt + bit;
        acc.x = __builtin_fmaf(bit, w[h].x, acc.x);
        acc.y = __builtin_fmaf(bit, w[h].y, acc.y);
        acc.z = __builtin_fmaf(bit, w[h].z, acc.z);
        acc.w = __builtin_fmaf(bit, w[h].w, acc.w);
      }
      const float4 e = masked_mean(acc, cnt);
      dst[((hi << kLow) | m) * 16 + l] = v4f{e.x, e.y, e.z, e.w};
    }
  }
}

// where the empty result that primes a wave's pipeline is "stored" (16 rows of 64 floats at most; contents never read).
// One sink per wave slot of the resident grid: 2048 waves writing the SAME 4 KiB at the start of every launch are a hot
// spot in one memory channel (rows mode, K = 64: 5.90 us per batch with one shared sink).
constexpr int kJunkSlots = 2048;
__device__ float g_junk[kJunkSlots][16 * 64];

// mi_oov_lsh_table_prepare: table f32[2^H][64] in global memory, one workgroup
template <int H>
__global__ __launch_bounds__(kPBlk) void lsh64_table_kernel(const float* __restrict__ buckets, float* __restrict__ table) {
  __shared__ __attribute__((aligned(16))) float sb[H * 64];
  if (threadIdx.x < H * 16)
    *reinterpret_cast<float4*>(sb + threadIdx.x * 4) = *reinterpret_cast<const float4*>(buckets + threadIdx.x * 4);
  __syncthreads();
  build_code_table<H>(sb, (gptr_v4)table, threadIdx.x, kPBlk);
}

// TAB: ids_src / other_src / out_src are device arrays of K pointers (one per batch); otherwise they ARE the
// pointers of the single batch (K = 1).  MODE_SCORE writes f32[B] scores; MODE_CODES writes u8[B,8] codes (H == 8);
// MODE_ROWS writes the f32[B,64] embedding rows themselves -- what LSHInductiveEmbedder.embed_*_ids returns
// (lsh_embedder.py:161-179).  LOOKUP (score / rows): BPR.get_*_embedding (bpr.py:48-125) -- ids below n_vocab address
// `vtable` f32[n_vocab,64] and return that row, the others take the lsh path on feat[id]; F == D, so ONE gather per
// lookup serves both cases with the base pointer selected per row (as lsh64_kernel<.., LOOKUP> does per batch).
// tab_prep: the table of aggregates made by lsh64_table_kernel for this bucket table, or null (built here).
template <int H, int MODE, bool TAB, bool LOOKUP = false>
__global__ __launch_bounds__(kPBlk, MI_PW) void lsh64_persistent_kernel(const void* __restrict__ ids_src,
                                                                        const void* __restrict__ other_src,
                                                                        void* __restrict__ out_src, unsigned K, unsigned B,
                                                                        const float* __restrict__ feat, int64_t N,
                                                                        const float* __restrict__ planes,
                                                                        const float* __restrict__ buckets,
                                                                        unsigned* __restrict__ sched,
                                                                        unsigned* __restrict__ sched_busy,
                                                                        const float* __restrict__ vtable, int64_t n_vocab,
                                                                        const float* __restrict__ tab_prep) {
  static_assert(MODE == kModeScore || MODE == kModeRows || MODE == kModeMean || H == 8, "codes are written as one 8-byte word per lookup");
  static_assert(!LOOKUP || MODE == kModeScore || MODE == kModeRows, "the in-vocabulary splice exists for scores and rows");
  constexpr bool kFromCodes = MODE == kModeFromCodes;  // ids = int32 slots, feat = u8[N,8] codes
  constexpr bool kScore = MODE == kModeScore || kFromCodes;
  // The knn aggregate on the same pipeline (mi_oov_gather_mean_multi, D = 64, the reference's group size 2):
  //   MODE_MEAN    out[o] = (W[idx[2o]] + W[idx[2o+1]]) / 2     `.split(2)` + mean (knn_embedder.py:125-126,146-147)
  // feat = W, B = outputs per batch; the second row of an output travels in the registers the score modes use for the
  // rows of the other side.  No planes, no table, no LDS.  (The plain gather out[b] = W[ids[b]] was tried here too and
  // is no faster than the grid-stride row_copy_kernel of gather.hip -- 108.2 against 107.9 us for 20 x 65536 rows -- so
  // mi_oov_gather_rows_multi stays there, for every row width.)
  constexpr bool kMean = MODE == kModeMean;
  constexpr bool kMover = kMean;
  constexpr bool kRows = MODE == kModeRows || kMover;  // f32[B,64] rows are written
  constexpr bool kTable = kScore || MODE == kModeRows;  // the 2^H-row table of aggregates is needed
  constexpr bool kPlanes = !kFromCodes && !kMover;
  constexpr int NU = (kScore || kMean) ? 4 : 1;
  const int lane = threadIdx.x & 63, l16 = lane & 15, grp = lane >> 4;
  const unsigned wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned tpb = (B + 15) / 16;
  const unsigned G = gridDim.x * kPWpb;
  const unsigned mean_m = static_cast<unsigned>(n_vocab);  // MODE_MEAN: indices per batch (2 B or 2 B - 1), in the n_vocab slot
#ifdef MI_PSTAMPS  // developer build (tools/multi_bench MB_STAMPS=1): per-wave s_memrealtime stamps behind the bucket table
  uint64_t* stamps = reinterpret_cast<uint64_t*>(const_cast<float*>(buckets) + H * 64) + (blockIdx.x * kPWpb + wv) * 4;
  const uint64_t st0 = __builtin_amdgcn_s_memrealtime();
  uint64_t st1 = 0, st2 = 0;
#define MI_STAMP(v) v = __builtin_amdgcn_s_memrealtime()
#define MI_STAMPS_OUT() do { if (lane == 0) { stamps[0] = st0; stamps[1] = st1; stamps[2] = st2; stamps[3] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define MI_STAMP(v)
#define MI_STAMPS_OUT()
#endif

  auto ids_of = [&](unsigned batch) -> gptr_i64 {
    return TAB ? (gptr_i64) reinterpret_cast<const int64_t* const*>(ids_src)[batch] : (gptr_i64)ids_src;
  };
  auto other_of = [&](unsigned batch) -> const char* {
    return TAB ? reinterpret_cast<const char* const*>(other_src)[batch] : reinterpret_cast<const char*>(other_src);
  };
  auto out_of = [&](unsigned batch) -> void* {
    return TAB ? reinterpret_cast<void* const*>(out_src)[batch] : out_src;
  };
  auto advance = [&](TilePos p) {
    p.local += G;
    while (p.local >= tpb) {
      p.local -= tpb;
      ++p.batch;
    }
    return p;
  };
  // stage A: the ids of a tile.  Lane (grp, l16) asks for the id of row (l16 & 3) * 4 + grp, i.e. lane r of a row
  // holds the id of round r (16 distinct addresses = one 128-B line per instruction).
  auto load_ids = [&](TilePos p) -> int64_t {
    unsigned row = p.local * 16u + (l16 & 3) * 4u + grp;
    row = row < B ? row : B - 1;  // tail tiles recompute the last row
    if constexpr (kFromCodes) return static_cast<int64_t>(((gptr_i32)ids_src)[row]);  // -1 / -2: no answer
    gptr_i64 idp = ids_of(p.batch);
    if constexpr (kMean) {  // lanes 0-3 of a row hold the first index of rounds 0-3, lanes 4-7 the second
      unsigned e = 2u * row + ((l16 >> 2) & 1u);
      e = e < mean_m ? e : mean_m - 1;  // an odd count: the last output has one row (its second slot is not used)
      return idp[e];
    }
    return idp[row];
  };
  // stage B: the 4 gathers first (second hop of the ids -> rows chain), the sequential rows of the other side behind
  auto load_rows = [&](TilePos p, int64_t idv, float4 (&x)[4], float4 (&u)[NU]) {
    int64_t id[4];
    round_ids(idv, id);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if constexpr (kFromCodes) {
        const bool valid = static_cast<uint64_t>(id[r]) < static_cast<uint64_t>(N);
        const v2u c = ((gptr_u2)feat)[valid ? id[r] : 0];  // one address per 16-lane row
        x[r].x = __uint_as_float(c.x);
        x[r].y = __uint_as_float(c.y);
      } else if constexpr (LOOKUP) {
        const bool oov = id[r] >= n_vocab;
        const bool valid = oov ? static_cast<uint64_t>(id[r]) < static_cast<uint64_t>(N) : id[r] >= 0;
        x[r] = gload4<MI_PNT_X != 0>((oov ? feat : vtable) + (valid ? id[r] : 0) * 64 + l16 * 4);
      } else {
        const bool valid = static_cast<uint64_t>(id[r]) < static_cast<uint64_t>(N);
        x[r] = gload4<MI_PNT_X != 0>(feat + (valid ? id[r] : 0) * 64 + l16 * 4);
      }
    }
    if constexpr (kMean) {
      int64_t id2[4];
      id2[0] = bcast_id<4>(idv);
      id2[1] = bcast_id<5>(idv);
      id2[2] = bcast_id<6>(idv);
      id2[3] = bcast_id<7>(idv);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool valid = static_cast<uint64_t>(id2[r]) < static_cast<uint64_t>(N);
        u[r] = gload4<MI_PNT_X != 0>(feat + (valid ? id2[r] : 0) * 64 + l16 * 4);
      }
    }
    if constexpr (kScore) {
      asm volatile("" ::: "memory");
      const char* up = other_of(p.batch);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        unsigned row = p.local * 16u + r * 4u + grp;
        row = row < B ? row : B - 1;
        u[r] = gload4<MI_PNT_U != 0>(up + (row * 256u + l16 * 16u));
      }
    }
  };

  // ---- schedule ------------------------------------------------------------------------------------------------------
  // Per-wave stamps of the statically scheduled kernel (tools/multi_bench MB_STAMPS=1): with every wave given exactly
  // the same 40 tiles, the first wave ends at 86 us and the last at 109 -- some XCDs and some waves of a CU get ~10 % less
  // of the memory system than others, and the launch lasts as long as its slowest wave.  So only the first three
  // quarters of the tiles are dealt statically (wave g takes tiles g, g + G, ...: no counter, and the sequential rows of
  // the other side are swept as one front); the rest is a POOL of tile pairs that waves draw from as they get there.
  // The pool has kPWpb ticket counters, 256 B apart; a wave uses counter (workgroup + wave slot) % kPWpb, so every
  // counter is shared by all eight wave slots and all eight XCDs alike (waves differ in speed along both: shared by wave
  // slot, the counters of the fast slots ran dry 20 us before the others), the fast waves take more, all counters run
  // dry together, and a counter sees one draw per 2 tiles of 1/kPWpb of the chip (~50 per us: an address takes ~90).
  // Ticket t of counter c is pair t * kPWpb + c, so the pairs in flight stay one front.  A draw is a SCALAR atomic
  // (s_atomic_add ... glc, waited for on lgkmcnt in the same asm statement): the wave's vector loads stay in flight
  // across it and their s_waitcnt vmcnt immediates are untouched (a returning vector atomic makes the compiler wait for
  // vmcnt(0) wherever its result is read, which drains the pipeline; tools/probe/satomic.hip checks the instruction).
  // A wave draws the ticket of its NEXT pair just before it would wait for a tile's rows anyway, so the counter's round
  // trip is spent in the shadow of that wait; its first ticket it draws at the very start of the kernel.
  // The last wave of a counter to leave resets it, and the last of those clears the counter set's "busy" word in host
  // memory, which is how the host knows it may hand the set to another launch (launches need no memset, no event).
  // sched == nullptr (a launch being captured into a graph, whose replays the host cannot see; or no free counter set):
  // the same tickets are dealt in a fixed order instead, workgroup b taking tickets b, b + gridDim.x, ... of its counters.
  const unsigned T = K * tpb;                               // tiles of the launch (<= 2^30)
  const unsigned nsp = static_cast<unsigned>((static_cast<uint64_t>(T) * MI_PSTATIC / 100) / (2 * G));  // static pairs per wave
  const unsigned Ts = 2 * G * nsp;                          // tiles dealt statically
  const unsigned Pp = (T - Ts + 1) / 2;                     // pool pairs (the last one may repeat tile T - 1)
  const unsigned shard = (blockIdx.x + wv) % kPWpb;
  unsigned* ticket_ctr = sched + shard * 64;
  unsigned* exit_ctr = ticket_ctr + 32;
  const unsigned gwave = blockIdx.x * kPWpb + wv;
  auto pos_of = [&](unsigned t) {
    TilePos p;
    p.batch = t / tpb;
    p.local = t - p.batch * tpb;
    return p;
  };
  unsigned dealt = blockIdx.x;  // (sched == nullptr)
  auto draw = [&]() -> unsigned {
    unsigned t = 1;
    if (sched) {
      asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(t) : "s"(ticket_ctr) : "memory");
    } else {
      t = dealt;
      dealt += gridDim.x;
    }
    return t;
  };
  auto pool_pair = [&](unsigned ticket, TilePos& p0, TilePos& p1) -> bool {
    const unsigned pair = ticket * kPWpb + shard;
    if (pair >= Pp) return false;
    const unsigned t0 = Ts + 2 * pair;
    p0 = pos_of(t0);
    p1 = pos_of(t0 + 1 < T ? t0 + 1 : T - 1);  // odd T: the very last pair scores tile T - 1 twice (same values)
    return true;
  };
  auto leave = [&]() {  // every wave passes here exactly once, after its last draw
    if (sched && lane == 0) {
      const unsigned done = __hip_atomic_fetch_add(exit_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (done == gridDim.x - 1) {  // the last wave of this counter (every workgroup has one): zero for the next launch
        // (zeroed with read-modify-writes, not stores: the counters are only ever touched by atomics, which all XCDs
        //  perform at one place -- a plain store could sit in this XCD's L2 until the end of the kernel)
        __hip_atomic_fetch_and(ticket_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_and(exit_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // Order: the two stores are acknowledged (vmcnt) before this counter is reported finished, and the busy word is
        // cleared by whoever sees all kPWpb reported.  No release fence: on this chip one is a write-back of the L2,
        // which the end of the kernel is about to do anyway, and all of this sits behind the launch's LAST wave.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned* counters_done = sched + 48;  // (third quarter of counter 0's line)
        if (__hip_atomic_fetch_add(counters_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kPWpb - 1) {
          __hip_atomic_fetch_and(counters_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef MI_PNOFLAG
          __hip_atomic_store(sched_busy, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
        }
      }
    }
  };

  // The ids of the wave's first two tiles go out before anything else: their round trip overlaps the staging of the
  // weights and the table build instead of following them.
  // (A launch with a static part draws its first ticket behind those loads; one without needs it to know its tiles.)
  TilePos pa, pb, pn, pm;
  unsigned tk = 0;
  bool have = true;
  if (nsp > 0) {
    pa = pos_of(gwave);
    pb = advance(pa);
  } else {
    have = pool_pair(draw(), pa, pb);
  }
  int64_t ida = 0, idb = 0, idn, idm;
  if (have) {
    ida = load_ids(pa);
    idb = load_ids(pb);
  }

  // Weights.  Planes and bucket rows are staged global -> LDS first; then -- the ids have landed with them, vmcnt
  // retires in order -- the rows of the wave's FIRST tile are requested, and the 2^H-row table of aggregates (score /
  // rows modes) is built out of LDS while they are in flight: the build issues no vector-memory instruction, so nothing
  // it waits for has the row gathers queued in front of it (MI_PEARLY=0: rows requested after the build, +~2 us per launch).
  // With a PREPARED table (tab_prep) there is nothing to build: its 2^H x 256 bytes are requested right behind the ids
  // (L2 hits after the first workgroups), the first tile's rows behind them, and the LDS copy is written while those
  // rows are in flight -- one barrier instead of two and no fmaf chains at the head of the launch.
#ifndef MI_PEARLY
#define MI_PEARLY 1
#endif
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [kTable ? 2^H : 0][64] table, [H][64] planes, [H][64] buckets
  constexpr int kTabRows = kTable ? (1 << H) : 0;
  float* splanes = sw + kTabRows * 64;
  float* sbuckets = splanes + H * 64;
  const bool prep = kTable && tab_prep != nullptr;  // (uniform)
  float4 xa[4], ua[NU], xb[4], ub[NU];
  {
    // (H * 16 <= 128 float4 per matrix: one per thread of the first two waves.)  The wave's first ticket is drawn
    // between the loads and the LDS stores: the draw waits for its reply, and waits while these loads are in flight.
    static_assert(H * 16 <= kPBlk, "one staging load per thread");
    constexpr int kTabPer = kTable ? ((16 << H) + kPBlk - 1) / kPBlk : 1;  // float4 of a prepared table per thread
    const bool stage = threadIdx.x < H * 16;
    float4 pv = make_float4(0.f, 0.f, 0.f, 0.f), bv = pv;
    float4 tv[kTabPer];
    if constexpr (kPlanes) {
      if (stage) pv = *reinterpret_cast<const float4*>(planes + threadIdx.x * 4);
    }
    if constexpr (kTable) {
      if (prep) {
#pragma unroll
        for (int q = 0; q < kTabPer; ++q) {
          const int i = q * kPBlk + static_cast<int>(threadIdx.x);
          tv[q] = ((16 << H) % kPBlk == 0 || i < (16 << H)) ? gload4(tab_prep + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      } else if (stage) {
        bv = *reinterpret_cast<const float4*>(buckets + threadIdx.x * 4);
      }
    }
    asm volatile("" ::: "memory");
    if constexpr (kTable) {
      // (prepared table: the ids are older than the table's loads, so they are here before those are -- the first tile's
      //  rows go out now, ahead of the ticket draw, and the LDS copy below is written in their shadow)
      if (prep && MI_PEARLY && have) load_rows(pa, ida, xa, ua);
    }
    asm volatile("" ::: "memory");
    if (have) tk = draw();
    asm volatile("" ::: "memory");
    if constexpr (kPlanes) {
      if (stage) *reinterpret_cast<float4*>(splanes + threadIdx.x * 4) = pv;
    }
    if constexpr (kTable) {
      if (prep) {
#pragma unroll
        for (int q = 0; q < kTabPer; ++q) {
          const int i = q * kPBlk + static_cast<int>(threadIdx.x);
          if ((16 << H) % kPBlk == 0 || i < (16 << H)) *reinterpret_cast<float4*>(sw + i * 4) = tv[q];
        }
      } else if (stage) {
        *reinterpret_cast<float4*>(sbuckets + threadIdx.x * 4) = bv;
      }
    }
  }
  if constexpr (!kMover) __syncthreads();
  if (MI_PEARLY && have && !prep) load_rows(pa, ida, xa, ua);
  if constexpr (kTable) {
    if (!prep) {  // table build: see build_code_table
      build_code_table<H>(sbuckets, reinterpret_cast<v4f*>(sw), static_cast<int>(threadIdx.x), kPBlk);
      __syncthreads();
    }
  }
  float4 pw[kPlanes ? H : 1];
  if constexpr (kPlanes) {
#pragma unroll
    for (int h = 0; h < H; ++h) pw[h] = *reinterpret_cast<const float4*>(splanes + (h * 16 + l16) * 4);
  }
  // which two planes the lane's bank holds after rows8_sum (t0: planes {0,2,1,3}[bank], t1: 4 + that)
  const int pl = (((l16 >> 2) & 1) << 1) | (l16 >> 3);
  const unsigned m0 = 1u << pl, m1 = 16u << pl;

  // the H sign bits of one gathered row as an integer, known to all 16 lanes of the row
  auto code_of = [&](const float4& x) -> unsigned {
    unsigned code;
    if constexpr (kFromCodes) {
      // bytes b0..b3 (each 0 / 1) of a word -> b0 + 2 b1 + 4 b2 + 8 b3: the four products land on bits 24..27 and
      // every other partial product on a bit of its own below them (no carries)
      const uint32_t lo = __float_as_uint(x.x), hi = __float_as_uint(x.y);
      code = ((lo * 0x01020408u) >> 24) | (((hi * 0x01020408u) >> 24) << 4);
      code &= 0xFFu;
    } else if constexpr (H == 8) {
      float p[8];
#pragma unroll
      for (int h = 0; h < 8; ++h) p[h] = dot4_fma(x, pw[h], 0.f);
      float t0, t1;
      rows8_sum(p, t0, t1);
      // >= 0, +-0 and NaN -> bit 1 (torch_hash.py:57-59); the bank's two bits, then OR over the four banks
      code = ((t0 < 0.f) ? 0u : m0) | ((t1 < 0.f) ? 0u : m1);
      code |= static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(code), 0x124, 0xF, 0xF, false));
      code |= static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(code), 0x128, 0xF, 0xF, false));
    } else {
      code = 0;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const float sdot = row16_sum(dot4_fma(x, pw[h], 0.f));
        code |= (sdot < 0.f) ? 0u : (1u << h);
      }
    }
    return code;
  };

  // stage C: reduce one tile whose rows were requested a whole iteration ago (`reduce`), and store its result
  // (`commit`).  The two are separate because of WHERE the store may sit in the wave's instruction stream: stores count
  // against vmcnt like loads and vmcnt retires in order, so a wait for a load that was issued behind a store also waits
  // for that store's acknowledgement from memory.  MI_PDEFER: a tile's result is kept in registers and stored right
  // behind the NEXT request of gathered rows -- every load the loop waits for was then issued before the youngest store
  // in flight, and no wait ever covers a store (rows mode: 4 x 16-byte stores per lane and tile).
  struct Res {
    float4 e[kRows ? 4 : 1];
    float sc;
    uint32_t lo, hi;
    void* op;        // the batch's output (wave-uniform)
    unsigned first;  // first row of the tile
    unsigned last;   // last row of the batch (B - 1): rows of a partial tile beyond it are stored ONTO it (see commit)
  };
  auto reduce = [&](TilePos p, int64_t idv, const float4 (&x)[4], const float4 (&u)[NU]) -> Res {
    Res res;
    res.op = out_of(p.batch);
    res.first = p.local * 16u;
    res.last = B - 1u;
    res.sc = 0.f;
    res.lo = res.hi = 0u;
    int64_t id[4];
    round_ids(idv, id);
    // the embedding of round r's lookup: the table row at its code; with LOOKUP the gathered row itself when the id is
    // in the vocabulary; NaN for an id that addresses no row
    auto emb_of = [&](int r) -> float4 {
      if constexpr (kMean) {
        // sum in increasing position from +0, one division by the group's length (torch.split: an odd count leaves the
        // last output a group of one) -- gather_mean_kernel's order, oracle/oov_oracle.c's
        int64_t id2;
        switch (r) {
          case 0: id2 = bcast_id<4>(idv); break;
          case 1: id2 = bcast_id<5>(idv); break;
          case 2: id2 = bcast_id<6>(idv); break;
          default: id2 = bcast_id<7>(idv); break;
        }
        unsigned o = p.local * 16u + r * 4u + grp;
        o = o < B ? o : B - 1u;  // (tail lanes recompute the last output, as their ids do)
        const bool two = 2u * o + 1u < mean_m;
        const bool ok = static_cast<uint64_t>(id[r]) < static_cast<uint64_t>(N) &&
                        (!two || static_cast<uint64_t>(id2) < static_cast<uint64_t>(N));
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        acc.x += x[r].x; acc.y += x[r].y; acc.z += x[r].z; acc.w += x[r].w;
        if (two) { acc.x += u[r].x; acc.y += u[r].y; acc.z += u[r].z; acc.w += u[r].w; }
        const float n = two ? 2.f : 1.f;
        acc.x /= n; acc.y /= n; acc.z /= n; acc.w /= n;
        return ok ? acc : make_float4(qnan(), qnan(), qnan(), qnan());
      } else {
      float4 emb = *reinterpret_cast<const float4*>(sw + (code_of(x[r]) * 16u + l16) * 4u);
      bool bad;
      if constexpr (LOOKUP) {
        const bool oov = id[r] >= n_vocab;
        if (!oov) emb = x[r];
        bad = oov ? !(static_cast<uint64_t>(id[r]) < static_cast<uint64_t>(N)) : id[r] < 0;
      } else {
        bad = !(static_cast<uint64_t>(id[r]) < static_cast<uint64_t>(N));
      }
      if constexpr (kFromCodes)  // a byte above 1 (0xFF): the owner saw an id outside its shard
        bad = bad || ((__float_as_uint(x[r].x) | __float_as_uint(x[r].y)) & 0xFEFEFEFEu) != 0;
      if (bad) emb = make_float4(qnan(), qnan(), qnan(), qnan());
      return emb;
      }
    };
    if constexpr (kScore) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = row16_sum(dot4_muladd(u[r], emb_of(r), 0.f));
        if (l16 == r) res.sc = s;  // lane r of group grp keeps round r's result
      }
    } else if constexpr (kRows) {
#pragma unroll
      for (int r = 0; r < 4; ++r) res.e[r] = emb_of(r);
      if constexpr (kMover) {
        // (three additions and a division: pinned HERE -- left to itself the compiler sinks them to the stores of the
        //  next step, behind a wait for every row request in between)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(res.e[r].x), "+v"(res.e[r].y), "+v"(res.e[r].z), "+v"(res.e[r].w));
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned c = code_of(x[r]);
        // byte h of the little-endian 8-byte code row = bit h (0 / 1), 0xFF bytes for an id outside the table
        uint32_t lo = (c & 1u) | ((c & 2u) << 7) | ((c & 4u) << 14) | ((c & 8u) << 21);
        uint32_t hi = ((c >> 4) & 1u) | ((c & 32u) << 3) | ((c & 64u) << 10) | ((c & 128u) << 17);
        if (!(static_cast<uint64_t>(id[r]) < static_cast<uint64_t>(N))) lo = hi = 0xFFFFFFFFu;
        if (l16 == r) {
          res.lo = lo;
          res.hi = hi;
        }
      }
    }
    return res;
  };
  // Every store of a tile is UNCONDITIONAL: a lane-predicated store of more than a few instructions is compiled into a
  // branch around it (s_cbranch_execz), and a memory instruction that may or may not have been issued makes every
  // s_waitcnt count behind it a lower bound -- the loop then waits for the stores' acknowledgements as well, or for
  // vmcnt(0).  Rows of a partial tile beyond the batch are stored ONTO its last row instead: those lanes were given the
  // last row's id (and row of the other side) by the clamped loads, so they hold the same value bit for bit and the
  // duplicate stores are benign.  The empty result the pipeline starts with goes to a junk sink (g_junk).
  auto commit = [&](const Res& res) {
    unsigned row = res.first + l16 * 4u + grp;
    row = row < res.last ? row : res.last;
    if constexpr (kScore) {
      // the tile's 16 contiguous scores in one store
      if (l16 < 4) ((gptr_f32)res.op)[row] = res.sc;
    } else if constexpr (kRows) {
      // four 256-byte rows per 16-lane group, all four stores together (a store between the rounds of `reduce` would sit
      // in front of the waits that follow it).  Non-temporal: the rows are written once and not read by this launch,
      // and kept out of L2 / Infinity Cache they leave those to the gathers (K = 64: 6.37 -> 5.67 us per batch).
      gptr_v4 op = (gptr_v4)res.op;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        unsigned orow = res.first + r * 4u + grp;
        orow = orow < res.last ? orow : res.last;
        const v4f v = {res.e[r].x, res.e[r].y, res.e[r].z, res.e[r].w};
#if MI_PNT_O
        __builtin_nontemporal_store(v, op + (orow * 16u + l16));
#else
        op[orow * 16u + l16] = v;
#endif
      }
    } else {
      // the tile's 16 code rows (128 contiguous bytes) in one store
      if (l16 < 4) {
        v2u v = {res.lo, res.hi};
        ((gptr_u2)res.op)[row] = v;
      }
    }
  };

  if (!have) {  // no tile for this wave (after the barriers above)
    leave();
    return;
  }
  MI_STAMP(st1);  // weights / table ready
  if (!MI_PEARLY) load_rows(pa, ida, xa, ua);
  // (The ids are in registers by now.  Saying so here -- an empty statement that "rewrites" them -- keeps the compiler's
  // wait-count analysis from carrying "idb may be the youngest load in flight", which it concludes from the branches of
  // the prologue, into the loop, where it would turn the first wait of every iteration into vmcnt(0).)
  asm volatile("" : "+v"(ida), "+v"(idb));
  // One iteration retires the pair (a, b) -- a's rows and b's ids are in flight -- and brings the next pair (n, m) to
  // that state:    ids(n)  rows(b)  reduce(a)      ids(m)  rows(n)  reduce(b)
  // with each result committed behind the rows request that follows its reduction (MI_PDEFER; the pipeline starts with an
  // empty result whose store no lane executes: the steady-state loop holds no conditional memory instruction).
#ifndef MI_PDEFER
#define MI_PDEFER 1
#endif
  Res pend;
  pend.op = g_junk[gwave % kJunkSlots];
  pend.first = 0u;
  pend.last = 15u;
  pend.sc = 0.f;
  pend.lo = pend.hi = 0u;
#pragma unroll
  for (int r = 0; r < (kRows ? 4 : 1); ++r) pend.e[r] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto step_pair = [&](bool pool) {
    idn = load_ids(pn);
    asm volatile("" ::: "memory");
    load_rows(pb, idb, xb, ub);
    asm volatile("" ::: "memory");
    if (MI_PDEFER) commit(pend);
    if (pool) tk = draw();
    // Rows modes: everything the wave has in flight but its youngest request is waited for before a tile is reduced --
    // more than the reduction needs (the compiler's own count would leave the next tile's rows and the last stores in
    // flight: MI_PROWS_WAIT=0), and faster: K = 64, 5.92 -> 5.72 us per batch (the version with lane-predicated stores,
    // whose waits were lower bounds, ran at that speed by accident and is where this was found).  Row stores and gathers
    // of one wave then alternate instead of overlapping; the other waves of the CU cover the wait.  The score modes,
    // which store 4 bytes per lookup, run the same with and without (5.14-5.20 us).
#if MI_PROWS_WAIT
    if constexpr (kRows) __builtin_amdgcn_s_waitcnt(0x0F70 | MI_PROWS_WAIT);
#endif
#if MI_PSCORE_WAIT
    if constexpr (kScore) __builtin_amdgcn_s_waitcnt(0x0F70 | MI_PSCORE_WAIT);
#endif
    const Res ra = reduce(pa, ida, xa, ua);
    if (!MI_PDEFER) commit(ra);
#ifdef MI_PSTAMPS
    if (st2 == 0) MI_STAMP(st2);  // first tile finished
#endif
    idm = load_ids(pm);
    asm volatile("" ::: "memory");
    load_rows(pn, idn, xa, ua);
    asm volatile("" ::: "memory");
    if (MI_PDEFER) commit(ra);
#if MI_PROWS_WAIT
    if constexpr (kRows) __builtin_amdgcn_s_waitcnt(0x0F70 | MI_PROWS_WAIT);
#endif
#if MI_PSCORE_WAIT
    if constexpr (kScore) __builtin_amdgcn_s_waitcnt(0x0F70 | MI_PSCORE_WAIT);
#endif
    pend = reduce(pb, idb, xb, ub);
    if (!MI_PDEFER) commit(pend);
    pa = pn; ida = idn;
    pb = pm; idb = idm;
  };
  // ONE loop for both phases (a second copy of the body costs register moves of rows in flight at its head)
  for (unsigned r = 1;;) {
    const bool pool = r >= nsp;
    if (!pool) {
      pn = advance(pb);
      pm = advance(pn);
      ++r;
    } else if (!pool_pair(tk, pn, pm)) {
      break;
    }
    step_pair(pool);
  }
  // drain: (a, b) is the wave's last pair
  load_rows(pb, idb, xb, ub);
  asm volatile("" ::: "memory");
  if (MI_PDEFER) commit(pend);
  commit(reduce(pa, ida, xa, ua));
  commit(reduce(pb, idb, xb, ub));
  MI_STAMPS_OUT();
  leave();
}

// Counter sets of the tile pool (see the kernel): kSchedSlots sets of kPWpb lines of 64 words (ticket counter, exit
// counter at +32, and in line 0 the count of finished counters at +48) per device, zero at load and left at zero by
// every launch.  A set serves ONE launch at a time: the host marks it busy in a word of pinned host memory when it hands
// it out, the last wave of the launch clears the word.  Nothing is handed out to a launch that is being captured (the
// kernel then deals the tickets in a fixed order), when every set is busy, or when MI_OOV_POOL=0.
constexpr int kSchedSlots = 32, kSchedDevices = 16;
__device__ unsigned g_sched[kSchedSlots][kPWpb * 64];
struct SchedSlot {
  unsigned* counters = nullptr;  // device
  unsigned* busy_dev = nullptr;  // device address of the busy word
  unsigned* busy = nullptr;      // host address of the same word
};
struct SchedPool {
  unsigned* counters = nullptr;
  unsigned* busy = nullptr;
  unsigned* busy_dev = nullptr;
  unsigned turn = 0;
  int state = 0;  // 0 new, 1 ready, -1 unusable
};
static SchedPool g_pools[kSchedDevices];
static std::mutex g_pool_mutex;

// The one allocation this library ever makes: kSchedSlots "busy" words (128 bytes) of pinned, device-mapped host memory
// per device, through which the last wave of a pool-scheduled launch hands its counter set back.  Made by mi_oov_init()
// or, when that was not called, by the first persistent launch on the device (never while a stream is being captured:
// such launches do not use the pool).  Called with g_pool_mutex held.
static bool init_pool_locked(SchedPool& pool) {
  if (pool.state == 0) {
    pool.state = -1;
    void* host = nullptr;
    if (hipGetSymbolAddress(reinterpret_cast<void**>(&pool.counters), HIP_SYMBOL(g_sched)) == hipSuccess &&
        hipHostMalloc(&host, kSchedSlots * sizeof(unsigned), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
      memset(host, 0, kSchedSlots * sizeof(unsigned));
      pool.busy = static_cast<unsigned*>(host);
      if (hipHostGetDevicePointer(reinterpret_cast<void**>(&pool.busy_dev), host, 0) == hipSuccess) pool.state = 1;
    }
    if (pool.state != 1) (void)hipGetLastError();
  }
  return pool.state == 1;
}

static bool pool_enabled() {
  static const bool v = env_knob("MI_OOV_POOL", 1, 0, 1) != 0;  // developer knob: 0 = always deal the tickets in a fixed order
  return v;
}

// A launch of fewer than kPoolMinPairs tile pairs per wave is over before the spread between waves has grown past what
// the pool's own costs are (the draws, and the hand-back after the last wave: ~1.5 us per launch; K = 4 batches of
// 65536: 6.2 us per batch dealt in fixed order, 6.6 from the pool; K = 20: 5.5 and 5.4; K = 64: 5.39 and 5.16).
constexpr int64_t kPoolMinPairs = 12;

static SchedSlot take_sched_slot(hipStream_t st, int64_t tiles, int grid) {
  SchedSlot none;
  if (!pool_enabled() || tiles < kPoolMinPairs * 2 * kPWpb * grid) return none;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return none;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kSchedDevices) return none;
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  SchedPool& pool = g_pools[dev];
  if (!init_pool_locked(pool)) return none;
  for (int i = 0; i < kSchedSlots; ++i) {
    const unsigned s = (pool.turn + i) % kSchedSlots;
    if (__atomic_load_n(&pool.busy[s], __ATOMIC_ACQUIRE) == 0) {
      __atomic_store_n(&pool.busy[s], 1u, __ATOMIC_RELAXED);
      pool.turn = s + 1;
      SchedSlot slot;
      slot.counters = pool.counters + static_cast<size_t>(s) * (kPWpb * 64);
      slot.busy_dev = pool.busy_dev + s;
      slot.busy = pool.busy + s;
      return slot;
    }
  }
  return none;
}
// after a launch: a set whose kernel never started is handed back by the host
static int check_pool_launch(const SchedSlot& slot) {
  const int rc = check_launch();
  if (rc != MI_OOV_OK && slot.busy) __atomic_store_n(slot.busy, 0u, __ATOMIC_RELEASE);
  return rc;
}

// resident workgroups of a persistent kernel on the current device (occupancy x CUs), cached per instantiation
template <typename Kern>
static int resident_blocks(Kern kernel, size_t lds, int& cached, int max_per_cu = 0) {
  if (cached > 0) return cached;
  int dev = 0, per_cu = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kPBlk, lds) != hipSuccess) return 0;
  if (per_cu < 1 || cus < 1) return 0;
  if (max_per_cu > 0 && per_cu > max_per_cu) per_cu = max_per_cu;
  cached = per_cu * cus;
  return cached;
}

static int grid_override() {
  // developer knob: grid size of the persistent launches (0 = occupancy x CUs)
  static const int v = static_cast<int>(env_knob("MI_OOV_MULTI_BLOCKS", 0, 0, 4096));
  return v;
}

// One host launcher for every instantiation.  TAB: a.ids / a.other / a.out are device arrays of K pointers and every
// batch has B lookups; otherwise they are the single batch's own pointers (K = 1), cut into launches of <= 2^23 rows
// (32-bit byte offsets into the batch inside the kernel).
struct PersistArgs {
  const void* ids;    // int64 ids (FROM_CODES: int32 slots)
  const void* other;  // f32[B,64] rows of the other side (score modes)
  void* out;          // f32[B] scores / u8[B,8] codes / f32[B,64] rows
  int64_t K, B;
  const float* feat;  // f32[N,64] (FROM_CODES: u8[N,8] codes)
  int64_t N;
  const float* vtable;  // LOOKUP: f32[n_vocab,64]
  int64_t n_vocab;
  const float *planes, *buckets, *tab;
};

template <int H, int MODE, bool TAB, bool LOOKUP>
static int launch_persistent(const PersistArgs& a, hipStream_t st) {
  constexpr bool kTable = MODE != kModeCodes && MODE != kModeMean;
  constexpr size_t lds = MODE == kModeMean ? 0 : ((kTable ? (size_t(1) << H) : 0) + (kTable ? 2 * H : H)) * 64 * sizeof(float);
  auto kern = lsh64_persistent_kernel<H, MODE, TAB, LOOKUP>;
  if (int rc = set_lds(kern, lds)) return rc;
  static int cached = 0;
  // MI_PROWS_WG workgroups per CU in rows mode (its 97 registers would admit two: measured in DESIGN.md section 5)
  int resident = grid_override() > 0 ? grid_override() : resident_blocks(kern, lds, cached, (MODE == kModeRows || MODE == kModeMean) ? MI_PROWS_WG : 0);
  if (resident <= 0) {
    g_last_hip_error = static_cast<int>(hipGetLastError());
    return MI_OOV_ERR_LAUNCH;
  }
  auto launch = [&](const void* ids, const void* other, void* out, int64_t k, int64_t b) -> int {
    const int64_t tiles = k * ((b + 15) / 16);
    const int64_t blocks_needed = (tiles + kPWpb - 1) / kPWpb;
    const int grid = static_cast<int>(blocks_needed < resident ? blocks_needed : resident);
    const SchedSlot cset = take_sched_slot(st, tiles, grid);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kPBlk), lds, st, ids, other, out, static_cast<unsigned>(k), static_cast<unsigned>(b),
                       a.feat, a.N, a.planes, a.buckets, cset.counters, cset.busy_dev, a.vtable, a.n_vocab, a.tab);
    return check_pool_launch(cset);
  };
  if constexpr (TAB) {
    const int64_t tpb = (a.B + 15) / 16;
    // 32-bit tile cursor inside the kernel: at most 2^30 tiles per launch
    const int64_t kmax = ((int64_t(1) << 30) / tpb) < 1 ? 1 : (int64_t(1) << 30) / tpb;
    for (int64_t k0 = 0; k0 < a.K; k0 += kmax) {
      const int64_t nk = (a.K - k0 < kmax) ? a.K - k0 : kmax;
      if (int rc = launch(static_cast<const char*>(a.ids) + k0 * 8, a.other ? static_cast<const char*>(a.other) + k0 * 8 : nullptr,
                          static_cast<char*>(a.out) + k0 * 8, nk, a.B))
        return rc;
    }
  } else {
    constexpr int64_t kMaxRows = int64_t(1) << 23;
    constexpr int64_t id_bytes = MODE == kModeFromCodes ? 4 : 8;
    constexpr int64_t out_bytes = (MODE == kModeRows || MODE == kModeMean) ? 256 : (MODE == kModeCodes ? 8 : 4);
    for (int64_t b0 = 0; b0 < a.B; b0 += kMaxRows) {
      const int64_t nb = (a.B - b0 < kMaxRows) ? a.B - b0 : kMaxRows;
      if (int rc = launch(static_cast<const char*>(a.ids) + b0 * id_bytes,
                          a.other ? static_cast<const char*>(a.other) + b0 * 256 : nullptr,
                          static_cast<char*>(a.out) + b0 * out_bytes, 1, nb))
        return rc;
    }
  }
  return MI_OOV_OK;
}

template <int MODE, bool TAB, bool LOOKUP>
static int launch_persistent_h(int H, const PersistArgs& a, hipStream_t st) {
  switch (H) {
#define MI_CASE(HV) \
  case HV: return launch_persistent<HV, MODE, TAB, LOOKUP>(a, st);
    MI_CASE(1) MI_CASE(2) MI_CASE(3) MI_CASE(4) MI_CASE(5) MI_CASE(6) MI_CASE(7) MI_CASE(8)
#undef MI_CASE
    default: return MI_OOV_ERR_SHAPE;
  }
}

// scores (`score` given; rows of the other side in `other`) or rows (`out` given), with or without the in-vocabulary
// table, K batches behind pointer tables (tab = true) or one batch
static int launch_persistent_any(bool tab, int H, const PersistArgs& a, bool score, hipStream_t st) {
#define MI_GO(M, T, L) return launch_persistent_h<M, T, L>(H, a, st)
  if (tab) {
    if (score) { if (a.vtable) MI_GO(kModeScore, true, true); MI_GO(kModeScore, true, false); }
    if (a.vtable) MI_GO(kModeRows, true, true);
    MI_GO(kModeRows, true, false);
  }
  if (score) { if (a.vtable) MI_GO(kModeScore, false, true); MI_GO(kModeScore, false, false); }
  if (a.vtable) MI_GO(kModeRows, false, true);
  MI_GO(kModeRows, false, false);
#undef MI_GO
}

// One LARGE batch of scores or rows (F = D = 64, H <= 8): host entry used by launch_lsh64 (lsh64.hip) from
// kPersistMinB lookups on -- below that a wave of this kernel has a tile or two and nothing to pipeline, and the
// per-batch kernel's one-tile-per-wave launch is the faster one (crossover measured in DESIGN.md section 5).
int launch_lsh64_persistent_single(const int64_t* ids, int64_t B, const float* feat, int64_t N, const float* vtable,
                                   int64_t n_vocab, const float* planes, int H, const float* buckets, const float* other,
                                   float* score, float* out, hipStream_t st) {
  if ((score && out) || (!score && !out)) return MI_OOV_ERR_SHAPE;  // one output per launch
  PersistArgs a{ids, other, score ? static_cast<void*>(score) : static_cast<void*>(out), 1, B, feat, N, vtable, n_vocab,
                planes, buckets, nullptr};
  return launch_persistent_any(false, H, a, score != nullptr, st);
}

// The knn aggregate on the persistent pipeline, D = 64, group size 2, K batches of M indices behind pointer tables: host
// entry used by mi_oov_gather_mean_multi (gather.hip).
int launch_gather_mean64_persistent(const int64_t* const* idx_tab, float* const* out_tab, int64_t K, int64_t M,
                                    const float* W, int64_t N, hipStream_t st) {
  const int64_t B = (M + 1) / 2;  // outputs per batch
  if (B > (int64_t(1) << 23) || M >= (int64_t(1) << 31)) return MI_OOV_ERR_SHAPE;
  PersistArgs a{idx_tab, nullptr, const_cast<float**>(out_tab), K, B, W, N, nullptr, M, nullptr, nullptr, nullptr};
  return launch_persistent<1, kModeMean, true, false>(a, st);
}

// Codes of one large batch (H = 8, F = 64): host entry used by launch_lsh64 (lsh64.hip) for codes-only calls of at
// least kCodesMinB lookups.  Below that a wave of the per-batch kernel has a single tile and nothing to pipeline.
int launch_lsh64_codes_persistent(const int64_t* ids, int64_t B, const float* feat, int64_t N, const float* planes,
                                  uint8_t* bits, hipStream_t st) {
  PersistArgs a{ids, nullptr, bits, 1, B, feat, N, nullptr, 0, planes, nullptr, nullptr};
  return launch_persistent<8, kModeCodes, false, false>(a, st);
}

// Requester side of a sharded lookup, D = 64, H = 8, score only: host entry used by mi_oov_lsh_codes_embed
// (exchange.hip).  codes u8[M,8] (8-byte aligned), slot i32[B], other f32[B,64], score f32[B].
int launch_lsh64_from_codes(const uint8_t* codes, int64_t M, const int32_t* slot, int64_t B, const float* buckets,
                            const float* other, float* score, hipStream_t st) {
  PersistArgs a{slot, other, score, 1, B, reinterpret_cast<const float*>(codes), M, nullptr, 0, nullptr, buckets, nullptr};
  return launch_persistent<8, kModeFromCodes, false, false>(a, st);
}

}  // namespace mi_oov

using namespace mi_oov;

extern "C" int mi_oov_init(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) {
    g_last_hip_error = static_cast<int>(hipGetLastError());
    return MI_OOV_ERR_LAUNCH;
  }
  if (dev >= kSchedDevices) return MI_OOV_OK;  // (launches on such a device deal their tiles in a fixed order)
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  return init_pool_locked(g_pools[dev]) ? MI_OOV_OK : MI_OOV_ERR_LAUNCH;
}

extern "C" int64_t mi_oov_lsh_table_bytes(int64_t H, int64_t D) {
  if (H < 1 || H > 8 || D != 64) return 0;
  return (int64_t(1) << H) * D * static_cast<int64_t>(sizeof(float));
}

extern "C" int mi_oov_lsh_table_prepare(const float* buckets, int64_t H, int64_t D, float* table, void* stream) {
  if (H < 1 || H > 8 || D != 64) return MI_OOV_ERR_SHAPE;
  if (!buckets || !table) return MI_OOV_ERR_NULL;
  if (!aligned16(buckets) || !aligned16(table)) return MI_OOV_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (H) {
#define MI_CASE(HV) \
  case HV: hipLaunchKernelGGL(lsh64_table_kernel<HV>, dim3(1), dim3(kPBlk), 0, st, buckets, table); break;
    MI_CASE(1) MI_CASE(2) MI_CASE(3) MI_CASE(4) MI_CASE(5) MI_CASE(6) MI_CASE(7) MI_CASE(8)
#undef MI_CASE
  }
  return check_launch();
}

extern "C" int mi_oov_lsh_multi(int mode, const int64_t* const* ids_tab, const float* const* other_tab, void* const* out_tab,
                                int64_t K, int64_t B, const float* vtable, int64_t n_vocab, const float* feat, int64_t N,
                                int64_t F, const float* planes, int64_t H, const float* buckets, int64_t D,
                                const float* table, void* stream) {
  if (mode != MI_OOV_LSH_SCORE && mode != MI_OOV_LSH_ROWS) return MI_OOV_ERR_KIND;
  if (K < 0 || B < 0 || N <= 0 || (vtable && n_vocab < 0)) return MI_OOV_ERR_SHAPE;
  if (K == 0 || B == 0) return MI_OOV_OK;
  if (!ids_tab || !out_tab || !feat || !planes || (!buckets && !table)) return MI_OOV_ERR_NULL;
  if (mode == MI_OOV_LSH_SCORE && !other_tab) return MI_OOV_ERR_NULL;
  // the persistent kernel exists for the register-resident shape only (callers fall back to K single launches)
  if (F != 64 || D != 64 || H < 1 || H > 8 || B > (int64_t(1) << 23)) return MI_OOV_ERR_SHAPE;
  if (!aligned16(feat) || !aligned16(planes) || (buckets && !aligned16(buckets)) || (table && !aligned16(table)) ||
      (vtable && !aligned16(vtable)) || (reinterpret_cast<uintptr_t>(ids_tab) & 7u) ||
      (reinterpret_cast<uintptr_t>(other_tab) & 7u) || (reinterpret_cast<uintptr_t>(out_tab) & 7u))
    return MI_OOV_ERR_ALIGN;
  PersistArgs a{ids_tab, other_tab, const_cast<void**>(out_tab), K, B, feat, N, vtable, n_vocab, planes, buckets, table};
  return launch_persistent_any(true, static_cast<int>(H), a, mode == MI_OOV_LSH_SCORE, static_cast<hipStream_t>(stream));
}

extern "C" int mi_oov_lsh_embed_score_multi(const int64_t* const* ids_tab, const float* const* other_tab,
                                            float* const* score_tab, int64_t K, int64_t B, const float* feat, int64_t N,
                                            int64_t F, const float* planes, int64_t H, const float* buckets, int64_t D,
                                            void* stream) {
  return mi_oov_lsh_multi(MI_OOV_LSH_SCORE, ids_tab, other_tab, reinterpret_cast<void* const*>(score_tab), K, B, nullptr, 0,
                          feat, N, F, planes, H, buckets, D, nullptr, stream);
}
