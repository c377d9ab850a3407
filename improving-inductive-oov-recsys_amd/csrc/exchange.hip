// Row-sharded tables: the two device-side ends of the owner-computes exchange.
// New functionality -- the reference has no table sharding (its only multi-GPU mode is DDP replicas,
// R/trainer/trainer.py:68-72); the arithmetic at both ends is the reference's lsh arithmetic
// (R/inductive/lsh_embedder.py:133-179), split where the data lives:
//
//   requester                                   owner of the feature row
//   ---------                                   ------------------------
//   bucket_by_owner: ids -> per-owner send      mi_oov_lsh_embed(bits only): feature row -> H sign bits
//   segments of LOCAL row numbers + the slot    (the F-wide row never leaves its GPU; 8 B of id in,
//   each lookup's answer will come back in      H bytes of code out)
//        ... all_to_all(ids) ...                     ... all_to_all(codes) ...
//   lsh_codes_embed: code at slot[b] -> masked mean of the (replicated) bucket rows -> score against the
//   requester's own row of the other side.
//
// Segments have a FIXED capacity, so the collective needs no counts from the device (no host sync anywhere in a
// step); a segment that would overflow drops the lookup (slot -1, NaN result) and leaves the true count in
// counts[w] > cap for the caller to see.  With cap = B nothing can overflow.
#include "common.hpp"

namespace mi_oov {

// A workgroup owns a contiguous chunk of the batch and reserves its share of every owner's segment with ONE global
// atomic per owner (a per-lookup or even per-wave atomic serialises on `world` addresses: 189 us for 1 M lookups).
// Two passes over the chunk (the second one re-reads the ids from L2):
//   pass 1  count the chunk's lookups per owner -> thread w reserves counts[w] += n_w and keeps the base
//   pass 2  hand every lookup a position base + (running count) + (rank inside the wave's ballot)
// The owner of a row is id / per, taken as a double-precision product with one correction step (ids < 2^52): a
// 64-bit integer division is ~40 instructions per lookup and pass.
__device__ __forceinline__ int owner_of(int64_t id, int64_t per, double inv_per, int world) {
  int64_t o = static_cast<int64_t>(static_cast<double>(id) * inv_per);
  if (o * per > id) --o;
  else if ((o + 1) * per <= id) ++o;
  return static_cast<int>(o < world ? o : world - 1);
}

// world <= kSmallWorld (every single-node job): the per-owner counts of a wave live in registers -- `world` ballots
// and popcounts per 64 lookups, no atomics at all inside the loops; a wave takes its range of the chunk's share with
// one LDS atomic per owner, the workgroup its range of the segment with one global atomic per owner.
//
// FUSED (round 4, mi_oov_bucket_by_owner_fused): the whole call in ONE launch.  The three-operation form -- a memset of the
// counts, this kernel, a kernel that fills the segments' tails with -1 -- is 17.7 us for 65536 lookups of which ~4.5 us
// each are the boundaries between the operations.  Here the reservations go to `scratch` (caller-owned words that are zero
// at launch and zero again at the end), and the LAST workgroup to finish publishes the counts, fills the tails (positions
// no other workgroup ever wrote) and resets the scratch.  my_rank >= 0 additionally COMPACTS the lookups this rank owns
// itself into `local_rows` (their answers never cross a link): they get slots world * cap + position, behind the slots
// of the exchanged answers, and the send segment of my_rank stays empty.
constexpr int kSmallWorld = 16, kUnroll = 4, kFillers = 32;
template <bool FUSED>
__global__ __launch_bounds__(256) void bucket_by_owner_small_kernel(const int64_t* __restrict__ ids, int64_t B, int64_t n_rows,
                                                                   int64_t per, int world, int64_t cap, int64_t chunk,
                                                                   int64_t* __restrict__ send, int32_t* __restrict__ slot,
                                                                   int32_t* __restrict__ counts, int32_t* __restrict__ overflow,
                                                                   int my_rank, int64_t* __restrict__ local_rows,
                                                                   unsigned* __restrict__ scratch) {
  __shared__ int32_t s_cnt[kSmallWorld], s_base[kSmallWorld];
  __shared__ unsigned s_last;
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < kSmallWorld) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const double inv_per = 1.0 / static_cast<double>(per);
  const int64_t lo = static_cast<int64_t>(blockIdx.x) * chunk;
  const int64_t hi = (lo + chunk < B) ? lo + chunk : B;
  const int64_t iters = (chunk + 255) / 256;  // (a multiple of kUnroll: the host rounds the chunk to 1024 lookups)
  const uint64_t below = (uint64_t(1) << lane) - 1;
  int32_t wcnt[kSmallWorld];  // wave-uniform
#pragma unroll
  for (int w = 0; w < kSmallWorld; ++w) wcnt[w] = 0;
  // kUnroll ids per thread are requested together (the ballots make the iterations dependent: without this one load
  // round trip per 256 lookups is exposed)
  const bool one_round = iters <= kUnroll;  // (a 65536-lookup call: 1024 per workgroup) the ids stay in registers for pass 2
  int64_t idkeep[kUnroll];
  for (int64_t it = 0; it < iters; it += kUnroll) {
    int64_t idv[kUnroll];
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) {
      const int64_t b = lo + (it + j) * 256 + threadIdx.x;
      idv[j] = (b < hi) ? ids[b] : -1;
      idkeep[j] = idv[j];
    }
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) {
      const bool valid = static_cast<uint64_t>(idv[j]) < static_cast<uint64_t>(n_rows);
      const int owner = valid ? owner_of(idv[j], per, inv_per, world) : -1;
#pragma unroll
      for (int w = 0; w < kSmallWorld; ++w)
        if (w < world) wcnt[w] += static_cast<int32_t>(__builtin_popcountll(__ballot(owner == w)));
    }
  }
  // lane w of the wave takes the wave's range inside the chunk's share of owner w
  int32_t mine_cnt = 0;
#pragma unroll
  for (int w = 0; w < kSmallWorld; ++w)
    if (lane == w) mine_cnt = wcnt[w];
  int32_t wave_base = 0;
  if (lane < world && mine_cnt) wave_base = atomicAdd(s_cnt + lane, mine_cnt);
  __syncthreads();
  if (threadIdx.x < world) {
    const int32_t n = s_cnt[threadIdx.x];
    int32_t base = 0;
    if (n) base = FUSED ? static_cast<int32_t>(atomicAdd(scratch + threadIdx.x, static_cast<unsigned>(n))) : atomicAdd(counts + threadIdx.x, n);
    s_base[threadIdx.x] = base;
    // the largest excess of any segment over its capacity, kept across calls (the caller reads it when it likes)
    if (overflow && static_cast<int64_t>(base) + n > cap) atomicMax(overflow, static_cast<int32_t>(base + n - cap));
  }
  __syncthreads();
  int64_t run[kSmallWorld];  // wave-uniform: next free position of this wave in segment w
#pragma unroll
  for (int w = 0; w < kSmallWorld; ++w)
    run[w] = (w < world) ? static_cast<int64_t>(s_base[w]) + __builtin_amdgcn_readlane(wave_base, w) : 0;
  for (int64_t it = 0; it < iters; it += kUnroll) {
    int64_t idv[kUnroll];
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) {
      const int64_t b = lo + (it + j) * 256 + threadIdx.x;
      idv[j] = one_round ? idkeep[j] : ((b < hi) ? ids[b] : -1);
    }
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) {
      const int64_t b = lo + (it + j) * 256 + threadIdx.x;
      const bool live = b < hi;
      const int64_t id = idv[j];
      const bool valid = static_cast<uint64_t>(id) < static_cast<uint64_t>(n_rows);
      const int owner = valid ? owner_of(id, per, inv_per, world) : -1;
      int64_t pos = 0;
#pragma unroll
      for (int w = 0; w < kSmallWorld; ++w) {
        if (w < world) {
          const uint64_t m = __ballot(owner == w);
          if (owner == w) pos = run[w] + __builtin_popcountll(m & below);
          run[w] += __builtin_popcountll(m);
        }
      }
      if (live) {
        int32_t my_slot = -2;  // invalid id: never sent, NaN at the requester (as the single-GPU kernel)
        if (valid) {
          if (pos < cap) {
            const int64_t local = id - static_cast<int64_t>(owner) * per;  // the owner's LOCAL row
            if (FUSED && owner == my_rank) {
              local_rows[pos] = local;
              my_slot = static_cast<int32_t>(static_cast<int64_t>(world) * cap + pos);
            } else {
              send[static_cast<int64_t>(owner) * cap + pos] = local;
              my_slot = static_cast<int32_t>(static_cast<int64_t>(owner) * cap + pos);
            }
          } else {
            my_slot = -1;  // dropped: counts[w] > cap tells the caller
          }
        }
        slot[b] = my_slot;
      }
    }
  }
  if constexpr (FUSED) {
    // The last nfill workgroups to get here fill the tails together, once EVERY workgroup has made its reservations (they
    // are atomics, performed in one place for all XCDs; at most nfill - 1 workgroups wait, for workgroups that are running
    // or still to be dispatched); one workgroup alone took 5 us for the 64 K tail entries of a 1 M-lookup call.
    __syncthreads();
    const unsigned G = gridDim.x;
    const unsigned nfill = G < static_cast<unsigned>(kFillers) ? G : static_cast<unsigned>(kFillers);
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(scratch + kSmallWorld, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned ticket = s_last;
    if (ticket + nfill < G) return;
    const unsigned f = ticket - (G - nfill);
    if (threadIdx.x == 0)
      while (__hip_atomic_load(scratch + kSmallWorld, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < G) __builtin_amdgcn_s_sleep(1);
    __syncthreads();
    if (threadIdx.x < kSmallWorld)
      s_cnt[threadIdx.x] = threadIdx.x < world ? static_cast<int32_t>(__hip_atomic_load(scratch + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0;
    __syncthreads();
    if (f == 0 && threadIdx.x < world) counts[threadIdx.x] = s_cnt[threadIdx.x];
    const int64_t step = static_cast<int64_t>(nfill) * 256;
    for (int w = 0; w < world; ++w) {  // unused entries of a segment are -1: positions nobody wrote
      const int64_t from = (w == my_rank) ? 0 : (s_cnt[w] < cap ? s_cnt[w] : cap);
      int64_t* seg = send + static_cast<int64_t>(w) * cap;
      for (int64_t q = from + static_cast<int64_t>(f) * 256 + threadIdx.x; q < cap; q += step) seg[q] = -1;
    }
    if (my_rank >= 0 && my_rank < world) {
      const int64_t from = s_cnt[my_rank] < cap ? s_cnt[my_rank] : cap;
      for (int64_t q = from + static_cast<int64_t>(f) * 256 + threadIdx.x; q < cap; q += step) local_rows[q] = -1;
    }
    __syncthreads();  // (this workgroup has read the counters)
    if (threadIdx.x == 0 &&
        __hip_atomic_fetch_add(scratch + kSmallWorld + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nfill - 1)
      for (int i = 0; i < kSmallWorld + 2; ++i)  // every filler is past the counters: zero for the next launch
        __hip_atomic_fetch_and(scratch + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// any world size: the counts live in LDS (one LDS atomic per wave, owner and iteration in both passes)
__global__ __launch_bounds__(256) void bucket_by_owner_kernel(const int64_t* __restrict__ ids, int64_t B, int64_t n_rows,
                                                             int64_t per, int world, int64_t cap, int64_t chunk,
                                                             int64_t* __restrict__ send, int32_t* __restrict__ slot,
                                                             int32_t* __restrict__ counts, int32_t* __restrict__ overflow) {
  extern __shared__ int32_t sh[];  // [world] chunk counts -> running positions, [world] global bases
  int32_t* s_cnt = sh;
  int32_t* s_base = sh + world;
  const int lane = threadIdx.x & 63;
  for (int w = threadIdx.x; w < world; w += 256) s_cnt[w] = 0;
  __syncthreads();
  const double inv_per = 1.0 / static_cast<double>(per);
  const int64_t lo = static_cast<int64_t>(blockIdx.x) * chunk;
  const int64_t hi = (lo + chunk < B) ? lo + chunk : B;
  const int64_t iters = (chunk + 255) / 256;  // every wave runs every iteration: the ballots need all 64 lanes
  for (int pass = 0; pass < 2; ++pass) {
    for (int64_t it = 0; it < iters; ++it) {
      const int64_t b = lo + it * 256 + threadIdx.x;
      const bool live = b < hi;
      const int64_t id = live ? ids[b] : -1;
      const bool valid = live && static_cast<uint64_t>(id) < static_cast<uint64_t>(n_rows);
      const int owner = valid ? owner_of(id, per, inv_per, world) : 0;
      int32_t my_slot = -2;  // invalid id: never sent, NaN at the requester (as the single-GPU kernel)
      uint64_t todo = __ballot(valid);
      while (todo) {
        const int leader = __builtin_amdgcn_readfirstlane(__builtin_ctzll(todo));
        const int w = __builtin_amdgcn_readlane(owner, leader);
        const uint64_t mine = __ballot(valid && owner == w);
        int32_t base = 0;
        if (lane == leader) base = atomicAdd(s_cnt + w, static_cast<int32_t>(__builtin_popcountll(mine)));
        if (pass == 1) {
          base = __builtin_amdgcn_readlane(base, leader);
          if (valid && owner == w) {
            const int64_t pos = static_cast<int64_t>(s_base[w]) + base + __builtin_popcountll(mine & ((uint64_t(1) << lane) - 1));
            if (pos < cap) {
              send[static_cast<int64_t>(w) * cap + pos] = id - static_cast<int64_t>(w) * per;  // the owner's LOCAL row
              my_slot = static_cast<int32_t>(static_cast<int64_t>(w) * cap + pos);
            } else {
              my_slot = -1;  // dropped: counts[w] > cap tells the caller
            }
          }
        }
        todo &= ~mine;
      }
      if (pass == 1 && live) slot[b] = my_slot;
    }
    __syncthreads();
    if (pass == 0) {
      for (int w = threadIdx.x; w < world; w += 256) {
        const int32_t n = s_cnt[w];
        const int32_t base = n ? atomicAdd(counts + w, n) : 0;
        s_base[w] = base;
        s_cnt[w] = 0;
        if (overflow && static_cast<int64_t>(base) + n > cap) atomicMax(overflow, static_cast<int32_t>(base + n - cap));
      }
      __syncthreads();
    }
  }
}

// Unused entries of a segment are -1 (the owner's kernel answers them with a 0xFF code without touching its table).
// Written AFTER the bucketing, and only where nothing was placed -- positions counts[w] .. cap of segment w (grid.y = w):
// with ids spread evenly and cap = 1.25 x the expected share that is a fifth of the buffer, and no byte of it is written
// twice.  (Round 2 cleared the whole buffer first: hipMemsetAsync's fill kernel, 5.4 us for the 10.5 MB of a 20-batch
// exchange, one more launch in a chain whose gaps -- 77 of 266 us on one GPU -- are host launch latency.)
__global__ __launch_bounds__(256) void fill_segment_tails_kernel(int64_t* __restrict__ send, const int32_t* __restrict__ counts,
                                                                int64_t cap) {
  const int w = blockIdx.y;  // segment
  const int64_t from = counts[w];
  int64_t* seg = send + static_cast<int64_t>(w) * cap;
  for (int64_t p = from + static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; p < cap; p += static_cast<int64_t>(gridDim.x) * 256) seg[p] = -1;
}

// Requester side: codes u8[M,H] as they came back, slot i32[B] -> emb f32[B,D] and / or score f32[B].
// A 16-lane row owns a lookup; lane l holds 4-float chunks l, l+16, ... of the output row (the canonical order of
// DESIGN.md section 4): acc = fmaf(bit_h, W[h][d], acc) for h = 0..H-1 from +0, one IEEE division by the exact count,
// score = 16-lane tree over the lanes' multiply-then-add chains.  Same operations as lsh_fused_kernel (lsh.hip).
// CHUNK (round 4): more bucket rows than the LDS holds (a model with thousands of OOV buckets) are staged hc at a time, the
// four waves in lock step; the chain over h runs on in order, so the bits are those of the resident form.
template <int DC, bool CHUNK = false>
__global__ __launch_bounds__(256) void lsh_codes_embed_kernel(const uint8_t* __restrict__ codes, int64_t M,
                                                             const int32_t* __restrict__ slot, int64_t B, int H,
                                                             const float* __restrict__ buckets, int D,
                                                             const float* __restrict__ other, float* __restrict__ score,
                                                             float* __restrict__ out, int hc) {
  extern __shared__ __attribute__((aligned(16))) float sW[];  // [H or hc][DC*64], zero padded
  constexpr int DP = DC * 64;
  auto stage = [&](int h0, int hn) {
    for (int i = threadIdx.x; i < hn * DP; i += 256) {
      const int h = i / DP, e = i - h * DP;
      sW[i] = (e < D) ? buckets[static_cast<int64_t>(h0 + h) * D + e] : 0.f;
    }
  };
  if constexpr (!CHUNK) {
    stage(0, H);
    __syncthreads();
  }
  const int HC = CHUNK ? hc : H;
  const int lane = threadIdx.x & 63, l16 = lane & 15, grp = lane >> 4, wv = threadIdx.x >> 6;
  const bool vec = (D % 4) == 0;
  const int64_t ngroups = (B + 3) / 4;  // a wave takes 4 lookups per pass
  for (int64_t g = static_cast<int64_t>(blockIdx.x) * 4 + wv; (CHUNK ? g - wv : g) < ngroups; g += static_cast<int64_t>(gridDim.x) * 4) {
    const int64_t b = g * 4 + grp;
    const bool live = b < B;
    const int32_t s = live ? slot[b] : -1;
    const bool have = s >= 0 && s < M;
    const uint8_t* crow = codes + static_cast<int64_t>(have ? s : 0) * H;
    float4 acc[DC];
#pragma unroll
    for (int c = 0; c < DC; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    float cnt = 0.f;
    bool ok = have;
    for (int h = 0; h < H; ++h) {
      if constexpr (CHUNK) {
        if (h % HC == 0) {  // (uniform)
          __syncthreads();
          stage(h, (H - h < HC) ? H - h : HC);
          __syncthreads();
        }
      }
      const uint8_t cb = crow[h];
      if (cb > 1) ok = false;  // 0xFF: the owner saw an id outside its shard
      const float bit = (cb == 1) ? 1.f : 0.f;
      cnt = cnt + bit;
#pragma unroll
      for (int c = 0; c < DC; ++c) {
        const float4 w = *reinterpret_cast<const float4*>(sW + (CHUNK ? h % HC : h) * DP + (c * 16 + l16) * 4);
        acc[c].x = __builtin_fmaf(bit, w.x, acc[c].x);
        acc[c].y = __builtin_fmaf(bit, w.y, acc[c].y);
        acc[c].z = __builtin_fmaf(bit, w.z, acc[c].z);
        acc[c].w = __builtin_fmaf(bit, w.w, acc[c].w);
      }
    }
    float sp = 0.f;
#pragma unroll
    for (int c = 0; c < DC; ++c) {
      const int e = (c * 16 + l16) * 4;
      float4 emb;
      emb.x = acc[c].x / cnt;  // 0/0 -> NaN row, as lsh_embedder.py:178
      emb.y = acc[c].y / cnt;
      emb.z = acc[c].z / cnt;
      emb.w = acc[c].w / cnt;
      if (!ok) emb = make_float4(qnan(), qnan(), qnan(), qnan());
      if (out && live) {
        if (vec) store4<true>(out + b * D, e, D, emb);
        else store4<false>(out + b * D, e, D, emb);
      }
      if (score) {
        const float* orow = other + (live ? b : 0) * D;
        float4 o = vec ? load4<true>(orow, e, D) : load4<false>(orow, e, D);
        if (e + 0 >= D) emb.x = 0.f;  // padded tail: exact zeros, as lsh_fused_kernel
        if (e + 1 >= D) emb.y = 0.f;
        if (e + 2 >= D) emb.z = 0.f;
        if (e + 3 >= D) emb.w = 0.f;
        sp = dot4_muladd(o, emb, sp);
      }
    }
    if (score) {
      const float tot = row16_sum(sp);
      if (l16 == 0 && live) score[b] = tot;
    }
  }
}

// lsh64p.hip: the persistent, software-pipelined requester kernel for D = 64, H = 8, score only
int launch_lsh64_from_codes(const uint8_t* codes, int64_t M, const int32_t* slot, int64_t B, const float* buckets,
                            const float* other, float* score, hipStream_t st);

}  // namespace mi_oov

// Workgroups a bucketing launch aims at (chunks are a multiple of 1024 lookups, so small calls get fewer): every workgroup makes one
// reservation per owner counter, ~11 ns each at the memory side whoever issues it, against which stands the parallelism of
// the two passes.  Round 4 sweep, one launch alone on the GPU, us: 1 M ids 512 / 2048 workgroups 24.0 / 28.9; 4 M ids 512 /
// 1024 / 2048 / 4096: 57.7 / 47.3 / 58.5 / 78.8.  MI_OOV_BUCKET_WGS (developer knob) forces a count.
static int64_t bucket_wgs(int64_t B) {
  static const int64_t v = mi_oov::env_knob("MI_OOV_BUCKET_WGS", 0, 0, 8192);
  return v > 0 ? v : (B >= (int64_t(1) << 21) ? 1024 : 512);
}
#define kBucketWgs bucket_wgs(B)

extern "C" int mi_oov_bucket_by_owner(const int64_t* ids, int64_t B, int64_t n_rows, int64_t rows_per_rank, int64_t world,
                                      int64_t cap, int64_t* send, int32_t* slot, int32_t* counts, int32_t* overflow,
                                      void* stream) {
  using namespace mi_oov;
  if (B < 0 || n_rows <= 0 || n_rows >= (int64_t(1) << 52) || rows_per_rank <= 0 || world <= 0 || world > 1024 || cap <= 0)
    return MI_OOV_ERR_SHAPE;
  if (world * cap > (int64_t(1) << 31) - 1) return MI_OOV_ERR_SHAPE;  // slots are int32
  if (!send || !counts) return MI_OOV_ERR_NULL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(counts, 0, static_cast<size_t>(world) * sizeof(int32_t), st) != hipSuccess) {
    g_last_hip_error = static_cast<int>(hipGetLastError());
    return MI_OOV_ERR_LAUNCH;
  }
  const int64_t fill_units = (cap + 255) / 256;  // (a segment's tail is walked by at most 2048 / world workgroups)
  const int64_t fill_max = kMaxGrid / world > 0 ? kMaxGrid / world : 1;
  const dim3 fill_grid(static_cast<unsigned>(fill_units < fill_max ? fill_units : fill_max), static_cast<unsigned>(world));
  if (B == 0) {
    hipLaunchKernelGGL(fill_segment_tails_kernel, fill_grid, dim3(256), 0, st, send, counts, cap);
    return check_launch();
  }
  if (!ids || !slot) return MI_OOV_ERR_NULL;
  // chunks of a multiple of 1024 lookups, at most ~bucket_wgs() workgroups (reservations per owner counter)
  int64_t chunk = (B + kBucketWgs - 1) / kBucketWgs;
  chunk = (chunk < 1024) ? 1024 : (chunk + 1023) / 1024 * 1024;
  const int grid = static_cast<int>((B + chunk - 1) / chunk);
  if (world <= kSmallWorld)
    hipLaunchKernelGGL(bucket_by_owner_small_kernel<false>, dim3(grid), dim3(256), 0, st, ids, B, n_rows, rows_per_rank,
                       static_cast<int>(world), cap, chunk, send, slot, counts, overflow, -1, static_cast<int64_t*>(nullptr),
                       static_cast<unsigned*>(nullptr));
  else
    hipLaunchKernelGGL(bucket_by_owner_kernel, dim3(grid), dim3(256), 2 * world * sizeof(int32_t), st, ids, B, n_rows,
                       rows_per_rank, static_cast<int>(world), cap, chunk, send, slot, counts, overflow);
  if (int rc = check_launch()) return rc;
  hipLaunchKernelGGL(fill_segment_tails_kernel, fill_grid, dim3(256), 0, st, send, counts, cap);
  return check_launch();
}

extern "C" int64_t mi_oov_bucket_by_owner_scratch(void) { return mi_oov::kSmallWorld + 2; }  // 32-bit words

extern "C" int mi_oov_bucket_by_owner_fused(const int64_t* ids, int64_t B, int64_t n_rows, int64_t rows_per_rank, int64_t world,
                                            int64_t cap, int64_t my_rank, int64_t* send, int32_t* slot, int32_t* counts,
                                            int32_t* overflow, int64_t* local_rows, uint32_t* scratch, void* stream) {
  using namespace mi_oov;
  if (B <= 0 || n_rows <= 0 || n_rows >= (int64_t(1) << 52) || rows_per_rank <= 0 || world <= 0 || world > kSmallWorld || cap <= 0)
    return MI_OOV_ERR_SHAPE;  // (B = 0 and larger worlds: mi_oov_bucket_by_owner)
  if ((world + 1) * cap > (int64_t(1) << 31) - 1 || my_rank >= world) return MI_OOV_ERR_SHAPE;  // slots are int32
  if (!ids || !send || !slot || !counts || !scratch || (my_rank >= 0 && !local_rows)) return MI_OOV_ERR_NULL;
  int64_t chunk = (B + kBucketWgs - 1) / kBucketWgs;
  chunk = (chunk < 1024) ? 1024 : (chunk + 1023) / 1024 * 1024;
  const int grid = static_cast<int>((B + chunk - 1) / chunk);
  hipLaunchKernelGGL(bucket_by_owner_small_kernel<true>, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), ids, B, n_rows,
                     rows_per_rank, static_cast<int>(world), cap, chunk, send, slot, counts, overflow,
                     static_cast<int>(my_rank < 0 ? -1 : my_rank), local_rows, scratch);
  return check_launch();
}

extern "C" int mi_oov_lsh_codes_embed(const uint8_t* codes, int64_t M, const int32_t* slot, int64_t B, int64_t H,
                                      const float* buckets, int64_t D, const float* other, float* score, float* out,
                                      void* stream) {
  using namespace mi_oov;
  if (B < 0 || M < 0 || H <= 0 || D <= 0 || D > 256) return MI_OOV_ERR_SHAPE;
  if (B == 0) return MI_OOV_OK;
  if (!codes || !slot || !buckets || (!score && !out) || (score && !other)) return MI_OOV_ERR_NULL;
  if (D % 4 == 0 && ((out && !aligned16(out)) || (other && !aligned16(other)))) return MI_OOV_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (D == 64 && H == 8 && score && !out && M > 0 && (reinterpret_cast<uintptr_t>(codes) & 7u) == 0 && aligned16(buckets))
    return launch_lsh64_from_codes(codes, M, slot, B, buckets, other, score, st);
  const int dc = static_cast<int>((D + 63) / 64);
  const int dcq = dc <= 1 ? 1 : (dc <= 2 ? 2 : 4);
  size_t lds = static_cast<size_t>(H) * dcq * 64 * sizeof(float);
  const bool chunked = static_cast<int64_t>(lds) > 64 * 1024;  // (beyond two workgroups per CU: bucket rows a chunk at a time)
  int hc = 0;
  if (chunked) {
    hc = static_cast<int>(32 * 1024 / (dcq * 64 * sizeof(float)));  // 32 KB per chunk
    lds = static_cast<size_t>(hc) * dcq * 64 * sizeof(float);
  }
  const int grid = grid_for(B, 16);
#define MI_GO(DCV, CH)                                                                                                \
  {                                                                                                                   \
    auto k = lsh_codes_embed_kernel<DCV, CH>;                                                                         \
    if (int rc = set_lds(k, lds)) return rc;                                                                          \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, codes, M, slot, B, static_cast<int>(H), buckets,            \
                       static_cast<int>(D), other, score, out, hc);                                                   \
    return check_launch();                                                                                            \
  }
  if (chunked) {
    if (dcq == 1) MI_GO(1, true)
    if (dcq == 2) MI_GO(2, true)
    MI_GO(4, true)
  }
  if (dcq == 1) MI_GO(1, false)
  if (dcq == 2) MI_GO(2, false)
  MI_GO(4, false)
#undef MI_GO
}
