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

}  // namespace mi_oov

using namespace mi_oov;

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
