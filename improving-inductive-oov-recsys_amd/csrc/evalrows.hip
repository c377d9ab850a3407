// Sampled-ranking evaluation, the rows either side of model.predict:
//
//   mi_oov_eval_rows_build   the (user, item) rows of NegSampleEvalDataLoader's batches
//                            (R/data/dataloader/general_dataloader.py:157-190, abstract_dataloader.py:227-235): per user
//                            its positives first, then `n_neg` sampled items per positive -- for a whole GROUP of
//                            consecutive batches in one pass, from the per-user positive counts (CSR), the positives and
//                            the negatives as they were drawn batch by batch.
//   mi_oov_segment_dedup     what the reference's scatter `scores[row_idx, col_idx] = origin_scores`
//                            (R/inductive/evaluator.py:118-134) does to a candidate that occurs twice for one user: ONE
//                            entry survives.  Here the first occurrence (a positive, if the item is one) keeps its column,
//                            later ones get column -1, which mi_oov_segment_topk skips when it is given a column range.
//
// Both are integer / index work, bit-exact against oracle/oov_oracle.c (oov_eval_rows_build, oov_segment_dedup).
#include "common.hpp"

namespace mi_oov {

// One wave per user: rows [seg_ptr[u], seg_ptr[u+1]) of the group, seg_ptr[u] = pos_ptr[u] * (1 + n_neg).
__global__ __launch_bounds__(kBlock) void eval_rows_build_kernel(const int64_t* __restrict__ pos_ptr, int64_t n_users,
                                                                 const int64_t* __restrict__ user_ids,
                                                                 const int64_t* __restrict__ pos_items,
                                                                 const int64_t* __restrict__ neg_items, int64_t n_neg,
                                                                 int64_t* __restrict__ row_user, int64_t* __restrict__ row_item,
                                                                 int64_t* __restrict__ seg_ptr, int64_t* __restrict__ pos_user) {
  const int lane = threadIdx.x & 63;
  const int64_t waves = static_cast<int64_t>(gridDim.x) * (kBlock / 64);
  for (int64_t u = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6); u < n_users; u += waves) {
    const int64_t p0 = pos_ptr[u], p1 = pos_ptr[u + 1];
    const int64_t c = p1 - p0, r0 = p0 * (1 + n_neg), n = c * (1 + n_neg);
    const int64_t uid = user_ids[u];
    for (int64_t o = lane; o < n; o += 64) {
      row_user[r0 + o] = uid;
      row_item[r0 + o] = o < c ? pos_items[p0 + o] : neg_items[p0 * n_neg + (o - c)];
    }
    if (pos_user)
      for (int64_t o = lane; o < c; o += 64) pos_user[p0 + o] = u;
    if (lane == 0) {
      seg_ptr[u] = r0;
      if (u == n_users - 1) seg_ptr[n_users] = p1 * (1 + n_neg);
    }
  }
}

// ---- first occurrence of every column inside a segment ---------------------------------------------------------------
// One workgroup per segment, an open-addressing table of positions in LDS: slot = the SMALLEST position seen so far of
// the column that hashed there (CAS on an empty slot, atomicMin on a slot that holds the same column, next slot on another
// column).  A position is a first occurrence iff the slot of its column holds it.  Segments of more than kDedupSlots / 4
// candidates are walked in P passes, pass p taking the columns whose hash falls into partition p (P a power of two with an
// expected load of <= 1/4 per pass); a pass whose table fills up is repeated with twice the partitions, so the result
// never depends on the distribution.
constexpr int kDedupSlots = 8192;  // 32 KiB of LDS: five workgroups per CU
constexpr uint64_t kDedupMaxSplit = 16;  // a partition is split at most this many times over (expected load then 1/64 of the slots)

__device__ __forceinline__ uint64_t dedup_mix(uint64_t x) {  // murmur3's finalizer: a bijection on 64 bits
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

__global__ __launch_bounds__(kBlock) void segment_dedup_kernel(const int64_t* __restrict__ cols, const int64_t* __restrict__ seg_ptr,
                                                               int64_t S, int64_t* __restrict__ out) {
  __shared__ int table[kDedupSlots];
  __shared__ int s_full;
  for (int64_t seg = blockIdx.x; seg < S; seg += gridDim.x) {
    const int64_t lo = seg_ptr[seg];
    const int64_t n64 = seg_ptr[seg + 1] - lo;
    const int n = n64 > 0x7FFFFFF0 ? 0x7FFFFFF0 : static_cast<int>(n64 < 0 ? 0 : n64);  // (positions are ints)
    const int64_t* c = cols + lo;
    int64_t* o = out + lo;
    // table size for this segment: a power of two >= 4 n, at most kDedupSlots (smaller tables are cheaper to clear)
    int slots = 256;
    while (slots < 4 * n && slots < kDedupSlots) slots *= 2;
    const int mask = slots - 1;
    uint64_t parts = 1;  // partitions: expected distinct columns per pass <= slots / 4
    while (static_cast<uint64_t>(n) > parts * (kDedupSlots / 4)) parts *= 2;
    const uint64_t max_parts = parts * kDedupMaxSplit;
    for (uint64_t p = 0; p < parts;) {
      for (int i = threadIdx.x; i < slots; i += kBlock) table[i] = -1;
      if (threadIdx.x == 0) s_full = 0;
      __syncthreads();
      for (int i = threadIdx.x; i < n; i += kBlock) {
        const int64_t col = c[i];
        const uint64_t hsh = dedup_mix(static_cast<uint64_t>(col));
        if (((hsh >> 20) & (parts - 1)) != p) continue;
        int h = static_cast<int>(hsh) & mask;
        bool placed = false;
        for (int probe = 0; probe < slots; ++probe) {
          int cur = table[h];
          if (cur < 0) {
            cur = atomicCAS(&table[h], -1, i);
            if (cur < 0) { placed = true; break; }
          }
          if (c[cur] == col) {  // (whatever position the slot holds later, it is one of this column)
            atomicMin(&table[h], i);
            placed = true;
            break;
          }
          h = (h + 1) & mask;
        }
        if (!placed) s_full = 1;
      }
      __syncthreads();
      if (s_full) {  // more distinct columns in this partition than slots (uniform decision)
        __syncthreads();
        if (parts < max_parts) {
          // split every partition in two: partition p keeps its index (the new bit is a higher one); the partitions
          // q + old parts, q < p, are walked again further on, which rewrites the same values
          parts *= 2;
          continue;
        }
        // thousands of distinct columns in a partition that should hold a hundred are not data, they are columns built to
        // collide: this partition is settled by comparing every candidate of it with its predecessors (terminates whatever
        // the input; tests/test_eval_rows.py builds such columns by inverting the mix)
        for (int i = threadIdx.x; i < n; i += kBlock) {
          const int64_t col = c[i];
          if (((dedup_mix(static_cast<uint64_t>(col)) >> 20) & (parts - 1)) != p) continue;
          bool first = true;
          for (int j = 0; j < i && first; ++j) first = c[j] != col;
          o[i] = first ? col : -1;
        }
        __syncthreads();
        ++p;
        continue;
      }
      for (int i = threadIdx.x; i < n; i += kBlock) {
        const int64_t col = c[i];
        const uint64_t hsh = dedup_mix(static_cast<uint64_t>(col));
        if (((hsh >> 20) & (parts - 1)) != p) continue;
        int h = static_cast<int>(hsh) & mask;
        for (;;) {  // the column is in the table: the walk ends at its slot
          const int cur = table[h];
          if (cur < 0 || c[cur] == col) {  // (cur < 0 cannot happen -- no slot is ever emptied -- and must not index c)
            o[i] = cur == i ? col : -1;
            break;
          }
          h = (h + 1) & mask;
        }
      }
      __syncthreads();
      ++p;
    }
  }
}

// ---- the TopkMetric family on the device (R/evaluator/metrics.py:36-235, base_metric.py:60-84) -----------------------------
// The reference turns the collected rec.topk block int[U, K+1] into per-user curves for k = 1..K in float64 NumPy and averages
// them over the users whose curve holds no NaN.  On the host that arithmetic was what an evaluation run waited for once the
// kernels around it were quick (36 000 users x 9 collectors x 5 metrics: 0.1 s of NumPy).  Here:
//   phase 1, one THREAD per user: the six curves exactly as NumPy computes a row -- sequential cumsums over k, int -> double
//            true divisions, the discount table 1 / log2(rank + 1) and its cumsum handed over FROM the host's NumPy (so
//            not an ulp of libm differs), -ffp-contract=off;
//   phase 2, one WAVE per (user side, metric), lane = rank k: the column sums over the selected users added ONE USER AFTER
//            THE OTHER in user order -- the order of NumPy's reduction over the leading axis -- so the sums, and with them
//            the means, are the float64 values NumPy returns, bit for bit (tests/test_eval_rows.py checks exactly that).
// Metric ids: 0 recall, 1 hit, 2 precision, 3 ndcg, 4 mrr, 5 map.  Only recall has NaN rows (no positive: 0 / 0).
constexpr int kNumMetrics = 6;

__global__ __launch_bounds__(kBlock) void topk_metric_curves_kernel(const int32_t* __restrict__ rec, int64_t U, int K,
                                                                    const double* __restrict__ disc, const double* __restrict__ idcg_base,
                                                                    const int64_t* __restrict__ uids, int64_t n_old_users,
                                                                    double* __restrict__ val, uint8_t* __restrict__ nanrow) {
  for (int64_t u = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; u < U; u += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int32_t* r = rec + u * (K + 1);
    const int64_t pos_len = r[K];
    const int64_t n = pos_len < K ? pos_len : K;  // np.minimum(pos_len, K)
    nanrow[u] = (pos_len == 0 ? 1 : 0) | ((uids && uids[u] < n_old_users) ? 2 : 0);  // flags: no positive | old user
    int64_t cum = 0;
    int first = -1;
    double dcg = 0.0, sum_pre = 0.0;
    for (int k = 0; k < K; ++k) {
      const bool h = r[k] != 0;  // rec[:, :-1].astype(bool)
      cum += h ? 1 : 0;
      if (h && first < 0) first = k;
      const double prec = static_cast<double>(cum) / static_cast<double>(k + 1);
      dcg = dcg + (h ? disc[k] : 0.0);
      sum_pre = sum_pre + prec * (h ? 1.0 : 0.0);
      const int cap = static_cast<int>(n <= 0 ? K - 1 : (k < n - 1 ? k : n - 1));  // NumPy's index -1 wraps to the last entry
      double* v = val + u * K + k;
      const int64_t plane = U * K;
      v[0 * plane] = static_cast<double>(cum) / static_cast<double>(pos_len);  // 0 / 0 -> NaN: the row is dropped (nanrow)
      v[1 * plane] = cum > 0 ? 1.0 : 0.0;
      v[2 * plane] = prec;
      v[3 * plane] = dcg / idcg_base[cap];
      v[4 * plane] = first >= 0 ? 1.0 / static_cast<double>(first + 1) : 0.0;
      v[5 * plane] = sum_pre / static_cast<double>(cap + 1);
    }
  }
}

// side 0: every user; 1: users with id < n_old_users (flag bit 1 of phase 1); 2: the others.
// One workgroup per (side, metric): wave 0 is the CHAIN -- user after user, one float64 addition per user and rank, that is
// the point -- and waves 1-3 feed it: they copy the curves of the next chunk of users (up to 32 KB, contiguous in memory) into
// the other LDS buffer, all of a thread's loads in flight together, a user that does not count landing as +0.0 (x + 0.0 == x
// for these sums: no curve value is negative or -0 and a sum that starts at +0.0 stays >= +0.0), so the chain reads one double
// per user and nothing else.  History of this kernel at 36 000 users x 10 ranks (one evaluation calls it three times): eight
// users' loads per round trip 10.7 ms; chunks staged through LDS 5.3; no branch per user 2.1; everything in ONE wave --
// requesting, landing and adding each cost ~0.4 ms of a lone wave's issue slots -- 1.2; feeders beside the chain: 0.40 ms (the chain itself: ~25 cycles per user).
constexpr int kSumChunk = 4096;   // doubles per LDS buffer
constexpr int kFeeders = 192;     // threads of waves 1-3
__global__ __launch_bounds__(256) void topk_metric_sums_kernel(const double* __restrict__ val, const uint8_t* __restrict__ flags,
                                                               int64_t U, int K, double* __restrict__ sums, int64_t* __restrict__ counts) {
  __shared__ double buf[2][kSumChunk];
  __shared__ int s_cnt;
  const int side = blockIdx.x / kNumMetrics, m = blockIdx.x % kNumMetrics;
  const int tid = threadIdx.x, lane = tid & 63;
  const bool feeder = tid >= 64;
  const int p = tid - 64;  // feeder index
  const double* v = val + static_cast<int64_t>(m) * U * K;
  const int CH = kSumChunk / K;  // users per chunk (K <= 256: at least 16)
  constexpr int R = (kSumChunk + kFeeders - 1) / kFeeders;  // elements of a chunk per feeder
  if (tid == 0) s_cnt = 0;
  auto counts_user = [&](uint8_t f) {
    bool s_ = true;
    if (side != 0) s_ = ((f & 2) != 0) == (side == 1);
    if (m == 0) s_ = s_ && (f & 1) == 0;
    return s_;
  };
  int my_cnt = 0;
  auto stage = [&](int64_t u0, int b) {  // feeders: users [u0, u0 + CH) -> buf[b]
    const int64_t nu = U - u0 < CH ? U - u0 : CH;
    const int64_t n = nu * K;
    const double* src = v + u0 * K;
    double x[R];
    uint8_t f[R];
    // element i = j kFeeders + p belongs to user i / K, followed incrementally
    int uu = p / K, rr = p % K;
    const int du = kFeeders / K, dr = kFeeders % K;
    int uj = uu, rj = rr;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int64_t i = static_cast<int64_t>(j) * kFeeders + p;
      x[j] = i < n ? src[i] : 0.0;
      f[j] = i < n ? flags[u0 + uj] : 0;
      uj += du;
      rj += dr;
      if (rj >= K) { rj -= K; ++uj; }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int64_t i = static_cast<int64_t>(j) * kFeeders + p;
      const bool s_ = i < n && counts_user(f[j]);
      if (i < kSumChunk) buf[b][i] = s_ ? x[j] : 0.0;
      if (s_ && rr == 0) ++my_cnt;  // (the element of rank 0 stands for its user)
      uu += du;
      rr += dr;
      if (rr >= K) { rr -= K; ++uu; }
    }
  };
  if (feeder && U > 0) stage(0, 0);
  __syncthreads();
  double acc[4] = {0.0, 0.0, 0.0, 0.0};  // ranks lane, lane + 64, lane + 128, lane + 192
  int b = 0;
  for (int64_t u0 = 0; u0 < U; u0 += CH, b ^= 1) {
    if (feeder) {
      if (u0 + CH < U) stage(u0 + CH, b ^ 1);
    } else {
      const int64_t nu = U - u0 < CH ? U - u0 : CH;
      if (K <= 64) {
        const int kl = lane < K ? lane : 0;
        constexpr int G = 16;  // users whose LDS reads are issued together, then added in order
        int64_t j = 0;
        for (; j + G <= nu; j += G) {
          double xv[G];
#pragma unroll
          for (int t = 0; t < G; ++t) xv[t] = buf[b][(j + t) * K + kl];
#pragma unroll
          for (int t = 0; t < G; ++t) acc[0] = acc[0] + xv[t];
        }
        for (; j < nu; ++j) acc[0] = acc[0] + buf[b][j * K + kl];
      } else {
        for (int64_t j = 0; j < nu; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (lane + 64 * q < K) acc[q] = acc[q] + buf[b][j * K + lane + 64 * q];
      }
    }
    __syncthreads();
  }
  if (feeder && my_cnt) atomicAdd(&s_cnt, my_cnt);
  __syncthreads();
  if (!feeder) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (lane + 64 * q < K) sums[(static_cast<int64_t>(side) * kNumMetrics + m) * K + lane + 64 * q] = acc[q];
    if (lane == 0) counts[side * kNumMetrics + m] = s_cnt;
  }
}

}  // namespace mi_oov

using namespace mi_oov;

extern "C" int64_t mi_oov_topk_metric_sums_workspace(int64_t U, int64_t K) {
  if (U <= 0 || K <= 0) return 0;
  return (kNumMetrics * U * K * 8 + U + 255) / 256 * 256;
}

extern "C" int mi_oov_topk_metric_sums(const int32_t* rec, int64_t U, int64_t K, const double* disc, const double* idcg_base,
                                       const int64_t* uids, int64_t n_old_users, int n_sides, double* sums, int64_t* counts,
                                       void* workspace, void* stream) {
  if (U < 0 || K <= 0 || K > 256 || (n_sides != 1 && n_sides != 3)) return MI_OOV_ERR_SHAPE;
  if (!sums || !counts) return MI_OOV_ERR_NULL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (U == 0) {
    if (hipMemsetAsync(sums, 0, static_cast<size_t>(n_sides) * kNumMetrics * K * 8, st) != hipSuccess ||
        hipMemsetAsync(counts, 0, static_cast<size_t>(n_sides) * kNumMetrics * 8, st) != hipSuccess) {
      check_launch();
      return MI_OOV_ERR_LAUNCH;
    }
    return MI_OOV_OK;
  }
  if (!rec || !disc || !idcg_base || !workspace || (n_sides == 3 && !uids)) return MI_OOV_ERR_NULL;
  if ((reinterpret_cast<uintptr_t>(workspace) & 7u) != 0) return MI_OOV_ERR_ALIGN;
  double* val = static_cast<double*>(workspace);
  uint8_t* nanrow = reinterpret_cast<uint8_t*>(val + kNumMetrics * U * K);
  hipLaunchKernelGGL(topk_metric_curves_kernel, dim3(grid_for(U, kBlock)), dim3(kBlock), 0, st, rec, U, static_cast<int>(K), disc, idcg_base,
                     n_sides == 3 ? uids : static_cast<const int64_t*>(nullptr), n_old_users, val, nanrow);
  if (int rc = check_launch()) return rc;
  hipLaunchKernelGGL(topk_metric_sums_kernel, dim3(static_cast<unsigned>(n_sides * kNumMetrics)), dim3(256), 0, st, val, nanrow, U,
                     static_cast<int>(K), sums, counts);
  return check_launch();
}

extern "C" int mi_oov_eval_rows_build(const int64_t* pos_ptr, int64_t n_users, const int64_t* user_ids, const int64_t* pos_items,
                                      const int64_t* neg_items, int64_t n_neg, int64_t* row_user, int64_t* row_item,
                                      int64_t* seg_ptr, int64_t* pos_user, void* stream) {
  if (n_users < 0 || n_neg < 0) return MI_OOV_ERR_SHAPE;
  if (!seg_ptr) return MI_OOV_ERR_NULL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n_users == 0) return hipMemsetAsync(seg_ptr, 0, sizeof(int64_t), st) == hipSuccess ? MI_OOV_OK : MI_OOV_ERR_LAUNCH;
  if (!pos_ptr || !user_ids || !pos_items || !row_user || !row_item || (n_neg > 0 && !neg_items)) return MI_OOV_ERR_NULL;
  hipLaunchKernelGGL(eval_rows_build_kernel, dim3(grid_for(n_users, kBlock / 64)), dim3(kBlock), 0, st, pos_ptr, n_users, user_ids,
                     pos_items, neg_items, n_neg, row_user, row_item, seg_ptr, pos_user);
  return check_launch();
}

extern "C" int mi_oov_segment_dedup(const int64_t* cols, const int64_t* seg_ptr, int64_t S, int64_t* out, void* stream) {
  if (S < 0) return MI_OOV_ERR_SHAPE;
  if (S == 0) return MI_OOV_OK;
  if (!cols || !seg_ptr || !out) return MI_OOV_ERR_NULL;
  if (cols == out) return MI_OOV_ERR_ALIAS;  // later passes of a long segment re-read the columns the earlier ones marked
  hipLaunchKernelGGL(segment_dedup_kernel, dim3(grid_for(S, 1)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), cols, seg_ptr, S, out);
  return check_launch();
}
