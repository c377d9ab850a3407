/*
 * mi_oov.h -- C ABI of libmi_oov.so, the MI355X (gfx950) inductive-OOV embedding hot path.
 *
 * The reference (snap-research/improving-inductive-oov-recsys) is pure Python and has no FFI;
 * each entry point below replaces the torch-op sequence of ONE reference function, cited as
 * R/ = RecBole/recbole/ in the reference tree.  A reference maintainer binds them with the
 * ctypes stubs shown in INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in _host;
 *   - matrices are dense row-major, float = IEEE binary32, ids = int64 (torch.long);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls only
 *     ENQUEUE work and never synchronise (graph-capture safe).  They never allocate or free
 *     either, with ONE exception: the persistent launches (mi_oov_lsh_multi and the calls that
 *     route to the same kernel) hand their tile-pool counters back through 128 bytes of pinned
 *     host memory per device, allocated once -- by mi_oov_init(), or by the first such launch on
 *     the device when mi_oov_init() was not called (never during stream capture);
 *   - return value: 0 = MI_OOV_OK, negative = error code (mi_oov_strerror); no C++
 *     exception crosses the boundary;
 *   - ids that do not address a row (id < 0 or id >= N) never fault: the row's outputs are
 *     NaN (floats) / -1 (indices) / 0xFF (bits).  The reference raises IndexError there
 *     (R/inductive/lsh_embedder.py:129 `feature_mat[nodes]`); the Python host mirror offers
 *     strict=True to reproduce the exception.
 *
 * Deterministic summation order ("canonical order"): a length-L dot
 * product is split over 16 lanes, lane l owning elements e with (e/4)%16 == l, accumulated
 * in increasing e; the 16 partials are then summed by a stride-halving tree (l with l+8, then
 * stride 4, 2, 1).  The
 * oracle (oracle/oov_oracle.c) uses the same order, so HIP results are BIT-EXACT against it.
 */
#ifndef MI_OOV_H
#define MI_OOV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_OOV_VERSION 100 /* 0.1.0 */

enum {
  MI_OOV_OK = 0,
  MI_OOV_ERR_NULL = -1,      /* required pointer is NULL                        */
  MI_OOV_ERR_SHAPE = -2,     /* negative / zero / unsupported dimension         */
  MI_OOV_ERR_KIND = -3,      /* unknown enum value (hash function ...)          */
  MI_OOV_ERR_LAUNCH = -4,    /* hipLaunchKernel / hipMemsetAsync reported error */
  MI_OOV_ERR_ALIGN = -5,     /* pointer not aligned as documented               */
  MI_OOV_ERR_WORKSPACE = -6, /* workspace too small                             */
  MI_OOV_ERR_ALIAS = -7      /* an output buffer is also an input               */
};

int mi_oov_version(void);
/* Optional, once per device (the current one) before the first launch: makes the library's one allocation up front
 * (see Conventions).  Idempotent; MI_OOV_ERR_LAUNCH if the pinned allocation fails (launches then still work: they deal
 * their tiles in a fixed order).                                                                                  */
int mi_oov_init(void);
const char* mi_oov_strerror(int code);
/* last hipError_t (as int) seen by a failing launch on this thread, 0 if none */
int mi_oov_last_hip_error(void);

/* ------------------------------------------------------------------------------------------
 * lsh: fused gather -> H sign-random-projections -> masked mean of bucket rows.
 * Replaces LSHInductiveEmbedder._hash_node + embed_user_ids/embed_item_ids
 *   (R/inductive/lsh_embedder.py:116-179) and TorchLSHash.hash_points (R/inductive/torch_hash.py:55-60):
 *     x = feat[ids];  bits = !(x @ planes^T < 0);  out = (bits @ buckets) / bits.sum(1)
 *   bit is 1 for projections that are >= 0, +-0 or NaN (torch_hash.py:57-59);
 *   an all-zero code gives 0/0 = NaN rows (lsh_embedder.py:178), reproduced.
 *   ids    i64[B]        feat    f32[N,F]      planes  f32[H,F]    buckets f32[H,D]
 *   out    f32[B,D] or NULL (hash only; buckets may then be NULL)
 *   bits   u8[B,H] or NULL  (0/1; 0xFF for invalid ids)
 * Shapes: any F, H >= 1 and D the reference takes.  F = D = 64 with H <= 64 are the hot tiles (a feature matrix
 * narrower than 64 columns can be given zero-padded to 64 with planes padded to match: same bits).
 * H is the model's number of OOV buckets (lsh_embedder.py:108-114) and may be in the thousands: planes and bucket rows
 * that do not fit the LDS are staged a chunk at a time.  D > 256: the rows are written one window of 256 columns per
 * launch; the FUSED score entries return MI_OOV_ERR_SHAPE there (compose mi_oov_lsh_embed / _lookup + mi_oov_rowdot, as
 * the Python mirror does).  Refused: F > ~5000 together with more planes than fit the LDS.
 * ------------------------------------------------------------------------------------------ */
int mi_oov_lsh_embed(const int64_t* ids, int64_t B,
                     const float* feat, int64_t N, int64_t F,
                     const float* planes, int64_t H,
                     const float* buckets, int64_t D,
                     float* out, uint8_t* bits, void* stream);

/* Backward of mi_oov_lsh_embed with respect to the bucket table (the only input of the lsh plugin that
 * carries a gradient: planes are read through `.data`, lsh_embedder.py:129):
 *     grad_buckets[h,d] = sum_b bits[b,h] * (grad_out[b,d] / popcount(bits[b,:]))
 *   bits u8[B,H] as returned by the forward; grad_out f32[B,D]; grad_buckets f32[H,D] (overwritten);
 *   workspace f32[mi_oov_lsh_backward_workspace(B,H,D)].  Deterministic (fixed two-pass order).      */
int64_t mi_oov_lsh_backward_workspace(int64_t B, int64_t H, int64_t D); /* number of floats */
int mi_oov_lsh_embed_backward(const uint8_t* bits, const float* grad_out, int64_t B, int64_t H, int64_t D,
                              float* grad_buckets, float* workspace, void* stream);

/* Backward of mi_oov_slsh_embed w.r.t. the bucket table: grad_buckets[idx[b],:] += grad_out[b,:]
 * (autograd of the `.weight[idx]` gather, single_lsh_embedder.py:87,109).  grad_buckets f32[n_buckets,D]
 * is overwritten.  n_buckets <= 64: the deterministic two-pass reduction of the lsh backward (workspace
 * f32[mi_oov_lsh_backward_workspace(B,n_buckets,D)]); larger tables: memset + float atomics.            */
int mi_oov_slsh_embed_backward(const int64_t* idx, const float* grad_out, int64_t B, int64_t n_buckets, int64_t D,
                               float* grad_buckets, float* workspace, void* stream);

/* The same two gradients in ONE launch (round 4): a workgroup writes the partial sums of its row partitions for every
 * plane, and the last workgroups to arrive add the partials up -- the same values in the same order, so the results are
 * bit-identical to the two-launch entry points above (and to oracle/oov_oracle.c).
 *   workspace f32[mi_oov_lsh_backward_fused_workspace(B, H or n_buckets, D)];
 *   counters u32[mi_oov_lsh_backward_fused_counters()], caller-owned device memory that is ZERO at the first launch and is
 *   left zero by every launch (the last workgroup resets it): allocate once, zero once, reuse -- also across HIP-graph
 *   replays.  Two launches that may run at the same time (different streams) need different counters.  counters == NULL:
 *   the same partials for every plane in one launch, then the final reduction as a second launch (16 consecutive columns per
 *   workgroup, 64 partials per thread in flight) -- two launches, no counters, the same bits.
 * slsh with more than 64 buckets or D > 256: as mi_oov_slsh_embed_backward (memset + float atomics; counters unused). */
int64_t mi_oov_lsh_backward_fused_workspace(int64_t B, int64_t H, int64_t D); /* number of floats */
int64_t mi_oov_lsh_backward_fused_counters(void);                            /* number of 32-bit words */
int mi_oov_lsh_embed_backward_fused(const uint8_t* bits, const float* grad_out, int64_t B, int64_t H, int64_t D,
                                    float* grad_buckets, float* workspace, uint32_t* counters, void* stream);
int mi_oov_slsh_embed_backward_fused(const int64_t* idx, const float* grad_out, int64_t B, int64_t n_buckets, int64_t D,
                                     float* grad_buckets, float* workspace, uint32_t* counters, void* stream);

/* out[idx[m],:] += g[m,:] with hardware float atomics (summation order not fixed): the backward of every
 * row gather on the path (mi_oov_gather_rows, mi_oov_splice_rows, knn's gather_mean after scaling).
 *   idx i64[M] (entries outside [0,N) are skipped); g f32[M,D]; out f32[N,D] (accumulated into).       */
int mi_oov_scatter_add_rows(const int64_t* idx, int64_t M, const float* g, int64_t N, int64_t D, float* out,
                            void* stream);

/* As mi_oov_lsh_embed, fused with the pairwise score of BPR.predict
 * (R/model/general_recommender/bpr.py:145-149): score[b] = sum_d other[b,d] * emb[b,d],
 * computed in canonical order with a separate multiply and add (torch.mul(...).sum(1)).
 *   other  f32[B,D]   the already-embedded opposite side (user rows for item lookups)
 *   score  f32[B]
 *   out    f32[B,D] or NULL: the embedding is not materialised when NULL            */
int mi_oov_lsh_embed_score(const int64_t* ids, int64_t B,
                           const float* feat, int64_t N, int64_t F,
                           const float* planes, int64_t H,
                           const float* buckets, int64_t D,
                           const float* other, float* score,
                           float* out, void* stream);

/* ---- row-sharded feature tables: the two device-side ends of the owner-computes exchange --------------------
 * (csrc/exchange.hip; the collective between them is the host's: torch.distributed all_to_all_single = RCCL).
 * The reference never shards a table (its multi-GPU mode is DDP replicas, R/trainer/trainer.py:68-72); the
 * arithmetic is R/inductive/lsh_embedder.py:133-179 split where the data lives.
 *
 * mi_oov_bucket_by_owner: lookups -> per-owner send segments of FIXED capacity (no counts leave the device, so a
 * step needs no host synchronisation).  Rank w owns global rows [w*rows_per_rank, (w+1)*rows_per_rank) (the last
 * rank up to n_rows).
 *   ids     i64[B]  global row numbers
 *   send    i64[world*cap]  (written) segment w = the owner-LOCAL rows asked of rank w, unused entries -1
 *   slot    i32[B]  (written) index into the [world*cap] answer array where lookup b's answer will be;
 *                   -2: id outside [0, n_rows) (never sent, NaN at the requester); -1: segment full (dropped)
 *   counts  i32[world]  (written) lookups owned by each rank; counts[w] > cap means lookups were dropped
 *   overflow i32[1] or NULL  (updated, never reset here) max over all calls of counts[w] - cap when positive: the
 *                   caller zeroes it once and reads it whenever it chooses to synchronise
 * Order inside a segment is unspecified (atomics); world*cap must be < 2^31.                              */
int mi_oov_bucket_by_owner(const int64_t* ids, int64_t B, int64_t n_rows, int64_t rows_per_rank, int64_t world,
                           int64_t cap, int64_t* send, int32_t* slot, int32_t* counts, int32_t* overflow,
                           void* stream);

/* The same in ONE launch (round 4; world <= 16, B > 0): no memset, no second kernel for the segments' tails -- the
 * reservations are made in `scratch` (u32[mi_oov_bucket_by_owner_scratch()], caller-owned, ZERO at the first launch and left
 * zero by every launch: allocate once per stream), and the last workgroups to finish write counts[], fill the unused
 * entries with -1 and reset the scratch.  my_rank >= 0 also COMPACTS the lookups whose row this rank owns itself:
 * local_rows i64[cap] receives their local row numbers (-1 behind the last), their slots are world * cap + position -- a
 * caller that appends the codes it computes for local_rows behind the world * cap exchanged ones hands the whole array to
 * mi_oov_lsh_codes_embed (M = (world + 1) * cap) -- and the send segment of my_rank stays empty (-1).  my_rank < 0:
 * exactly mi_oov_bucket_by_owner's outputs.  MI_OOV_ERR_SHAPE for world > 16 or B = 0 (use mi_oov_bucket_by_owner).   */
int64_t mi_oov_bucket_by_owner_scratch(void); /* number of 32-bit words */
int mi_oov_bucket_by_owner_fused(const int64_t* ids, int64_t B, int64_t n_rows, int64_t rows_per_rank, int64_t world,
                                 int64_t cap, int64_t my_rank, int64_t* send, int32_t* slot, int32_t* counts,
                                 int32_t* overflow, int64_t* local_rows, uint32_t* scratch, void* stream);

/* Requester side of a sharded lsh lookup: the owners returned codes u8[M,H] (mi_oov_lsh_embed with bits only,
 * 0xFF rows for ids outside their shard); lookup b's code sits at row slot[b].
 *   emb[b] = (code @ buckets) / popcount  -- same fmaf chain and division as mi_oov_lsh_embed, bit-identical --
 *   score[b] = sum_d other[b,d] * emb[b,d] (as mi_oov_lsh_embed_score).  NaN where slot[b] < 0 or the code is 0xFF.
 *   buckets f32[H,D], D <= 256; other f32[B,D] (needed with score); score f32[B] or NULL; out f32[B,D] or NULL.  */
int mi_oov_lsh_codes_embed(const uint8_t* codes, int64_t M, const int32_t* slot, int64_t B, int64_t H,
                           const float* buckets, int64_t D, const float* other, float* score, float* out,
                           void* stream);

/* K batches of mi_oov_lsh_embed_score in ONE persistent launch (csrc/lsh64p.hip): what a serving / evaluation
 * loop that has K batches queued calls instead of K launches.  Replaces K times the reference's op sequence
 * LSHInductiveEmbedder.embed_item_ids + BPR.predict (R/inductive/lsh_embedder.py:161-179,
 * R/model/general_recommender/bpr.py:145-149); batch k gives exactly the scores mi_oov_lsh_embed_score gives
 * for (ids_tab[k], other_tab[k]) -- same arithmetic, same order, bit-identical.
 *   ids_tab    DEVICE array of K device pointers, each int64[B] (8-byte aligned)
 *   other_tab  DEVICE array of K device pointers, each f32[B,D] (16-byte aligned rows)
 *   score_tab  DEVICE array of K device pointers, each f32[B] (written)
 *   feat, planes, buckets as in mi_oov_lsh_embed (16-byte aligned); every batch has the same B.
 * Only the register-resident shape is served: F == D == 64, 1 <= H <= 8, B <= 2^23; any other shape returns
 * MI_OOV_ERR_SHAPE and the caller issues K single launches.  The pointer tables are read by the kernel: they
 * must stay valid (and unchanged) until the launch has completed on `stream`.                          */
int mi_oov_lsh_embed_score_multi(const int64_t* const* ids_tab, const float* const* other_tab,
                                 float* const* score_tab, int64_t K, int64_t B,
                                 const float* feat, int64_t N, int64_t F,
                                 const float* planes, int64_t H,
                                 const float* buckets, int64_t D, void* stream);

/* The aggregate (bits @ buckets) / popcount takes only 2^H values: mi_oov_lsh_table_prepare writes them once --
 * table f32[2^H, D], row c = the embedding of code c (bit h of c = plane h), made with exactly the per-lookup arithmetic
 * (fmaf chain over the bucket rows in plane order from +0, one correctly rounded division; row 0 = 0/0 = NaN,
 * lsh_embedder.py:178) -- and the persistent launches below load it instead of rebuilding it at their head.  The table
 * is valid for the bucket values it was made from (the Python mirror re-prepares when the bucket tensor's version
 * counter moves).  mi_oov_lsh_table_bytes: size of `table` in bytes, 0 for shapes the persistent kernel does not take
 * (it takes D == 64, 1 <= H <= 8).  buckets, table 16-byte aligned.                                               */
int64_t mi_oov_lsh_table_bytes(int64_t H, int64_t D);
int mi_oov_lsh_table_prepare(const float* buckets, int64_t H, int64_t D, float* table, void* stream);

/* K queued batches of ANY of the four lsh per-batch calls in ONE persistent launch (csrc/lsh64p.hip):
 *   mode MI_OOV_LSH_SCORE, vtable NULL   K x mi_oov_lsh_embed_score   (= mi_oov_lsh_embed_score_multi)
 *   mode MI_OOV_LSH_SCORE, vtable given  K x mi_oov_lsh_lookup_score  (BPR.predict, bpr.py:145-149 over :94-125)
 *   mode MI_OOV_LSH_ROWS,  vtable NULL   K x mi_oov_lsh_embed         (LSHInductiveEmbedder.embed_*_ids,
 *                                                                      lsh_embedder.py:141-179: the [B,D] rows)
 *   mode MI_OOV_LSH_ROWS,  vtable given  K x mi_oov_lsh_lookup        (BPR.get_*_embedding, bpr.py:48-125)
 * batch k gives exactly what the per-batch call gives for (ids_tab[k], other_tab[k]): same arithmetic, same order.
 *   ids_tab    DEVICE array of K device pointers, each int64[B]
 *   other_tab  DEVICE array of K device pointers, each f32[B,D] (score mode; NULL in rows mode)
 *   out_tab    DEVICE array of K device pointers, each f32[B] (scores) or f32[B,D] (rows; 16-byte aligned), written
 *   vtable     f32[n_vocab,D] or NULL;  table: mi_oov_lsh_table_prepare's output for `buckets`, or NULL (built per launch;
 *              `buckets` may be NULL when a table is given)
 * Shapes and pointer-table lifetime as for mi_oov_lsh_embed_score_multi.                                          */
enum { MI_OOV_LSH_SCORE = 0, MI_OOV_LSH_ROWS = 1 };
int mi_oov_lsh_multi(int mode, const int64_t* const* ids_tab, const float* const* other_tab, void* const* out_tab,
                     int64_t K, int64_t B, const float* vtable, int64_t n_vocab,
                     const float* feat, int64_t N, int64_t F,
                     const float* planes, int64_t H,
                     const float* buckets, int64_t D, const float* table, void* stream);

/* BPR.get_user_embedding / get_item_embedding with an lsh plugin, one launch
 * (R/model/general_recommender/bpr.py:48-125): rows with id < n_vocab are copied from
 * `table`, the others take the lsh path above on feat[id].
 *   table  f32[n_vocab,D]                                                            */
int mi_oov_lsh_lookup(const int64_t* ids, int64_t B,
                      const float* table, int64_t n_vocab,
                      const float* feat, int64_t N, int64_t F,
                      const float* planes, int64_t H,
                      const float* buckets, int64_t D,
                      float* out, void* stream);

/* BPR.predict with an lsh plugin on this side (bpr.py:145-149 over :94-125): the lookup above fused
 * with the row dot against the already-embedded other side; `out` may be NULL.            */
int mi_oov_lsh_lookup_score(const int64_t* ids, int64_t B,
                            const float* table, int64_t n_vocab,
                            const float* feat, int64_t N, int64_t F,
                            const float* planes, int64_t H,
                            const float* buckets, int64_t D,
                            const float* other, float* score,
                            float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * slsh: SingleLSHInductiveEmbedder._hash_node + embed_* (R/inductive/single_lsh_embedder.py:82-109)
 *     bits as above with H = bits_req planes;  idx = (sum_h 2**bits[h]) % n_buckets
 *     = (H + popcount(bits)) % n_buckets  (reference quirk, :86);  out = buckets[idx]
 *   buckets f32[n_buckets,D];  idx i64[B] or NULL;  out f32[B,D] or NULL
 *   H = 0 (n_buckets = 1: bits_req = ceil(log2(1)) = 0; planes may be NULL) puts every lookup in bucket 0.
 * ------------------------------------------------------------------------------------------ */
int mi_oov_slsh_embed(const int64_t* ids, int64_t B,
                      const float* feat, int64_t N, int64_t F,
                      const float* planes, int64_t H,
                      const float* buckets, int64_t n_buckets, int64_t D,
                      float* out, int64_t* idx, void* stream);

/* K queued batches of mi_oov_slsh_embed in ONE launch (every batch B ids).  Hot tile only: F == 64, H <= 32 and D 64 or
 * 128 when rows are wanted; any other shape returns MI_OOV_ERR_SHAPE (callers issue K single launches).
 * ids_tab / out_tab / idx_tab: DEVICE arrays of K device pointers; out_tab or idx_tab may be NULL.                */
int mi_oov_slsh_embed_multi(const int64_t* const* ids_tab, float* const* out_tab, int64_t* const* idx_tab, int64_t K,
                            int64_t B, const float* feat, int64_t N, int64_t F,
                            const float* planes, int64_t H,
                            const float* buckets, int64_t n_buckets, int64_t D, void* stream);

/* ------------------------------------------------------------------------------------------
 * dhe: DeepHashEmbedder._get_hashes / _hash_ids (R/inductive/dh_embedder.py:140-170):
 *     out[b,j] = float( SipHash-2-4(key_j, LE64(ids[b])) mod mod )      (mod = 16777216)
 *   keys u8[K,16];  out f32[B,K].  SipHash-2-4 is the third-party csiphash==0.0.5
 *   (RecBole/setup.py:23), restated from the published algorithm.  Bit-exact.
 *   mod must be a power of two <= 2^24 so the float conversion is exact.
 * ------------------------------------------------------------------------------------------ */
int mi_oov_siphash24_mod(const int64_t* ids, int64_t B,
                         const uint8_t* keys, int64_t K, uint32_t mod,
                         float* out, void* stream);
/* the same into rows of ld >= K floats (out f32[B,ld], columns K .. ld untouched): fdhe writes its hashes straight into
 * the [B, K + F (+ padding)] input of its net (feat_dh_embedder.py:164-172) instead of concatenating afterwards */
int mi_oov_siphash24_mod_ld(const int64_t* ids, int64_t B, const uint8_t* keys, int64_t K, uint32_t mod,
                            float* out, int64_t ld, void* stream);

/* ------------------------------------------------------------------------------------------
 * random mapper: RandomOOVInductiveMapper (R/inductive/random_mapper.py:70-130).
 *   kind: 0 'mod', 1 'fast' (:70-76), 2 '3round' (:78-86), 3 '64bit' (:95-102)
 *   mi_oov_mapper_hash : out[b] = hash_kind(ids[b])            (raw signed-int64 mixer, no modulo;
 *                        for '64bit' and 'mod' the raw value before `% n_buckets`)
 *   mi_oov_mapper_map  : map_user_ids/map_item_ids (:116-130):
 *        out[b] = ids[b]                                     if ids[b] <  n_orig
 *               = pymod(hash(ids[b]-n_orig), n_buckets)+n_orig otherwise
 *        (signed mixers with arithmetic >>, Python-style non-negative %; '64bit' is uint64)
 * ------------------------------------------------------------------------------------------ */
enum { MI_OOV_HASH_MOD = 0, MI_OOV_HASH_FAST = 1, MI_OOV_HASH_3ROUND = 2, MI_OOV_HASH_64BIT = 3 };
int mi_oov_mapper_hash(const int64_t* ids, int64_t B, int kind, int64_t* out, void* stream);
int mi_oov_mapper_map(const int64_t* ids, int64_t B, int kind,
                      int64_t n_orig, int64_t n_buckets, int64_t* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * knn aggregate: KNNInductiveEmbedder.embed_* tail (R/inductive/knn_embedder.py:125-126,146-147):
 *     rows = W[idx.ravel()];  out = vstack(chunk.mean(0) for chunk in rows.split(g))
 *   idx i64[M] (flattened [B,k]);  W f32[N,D];  out f32[ceil(M/g), D]; the last group may be
 *   short and is averaged over its own length (torch.split semantics).  The reference fixes
 *   g = 2.  Sum in increasing position, then one division.
 * ------------------------------------------------------------------------------------------ */
int mi_oov_gather_mean(const int64_t* idx, int64_t M, int64_t g,
                       const float* W, int64_t N, int64_t D,
                       float* out, void* stream);

/* K queued batches of mi_oov_gather_mean in ONE launch (every batch M indices, same g): idx_tab / out_tab are DEVICE
 * arrays of K device pointers (int64[M] / f32[ceil(M/g), D]); D a multiple of 4, W and the output rows 16-byte aligned.
 * Batch k gives exactly what mi_oov_gather_mean gives for idx_tab[k].                                            */
int mi_oov_gather_mean_multi(const int64_t* const* idx_tab, float* const* out_tab, int64_t K, int64_t M, int64_t g,
                             const float* W, int64_t N, int64_t D, void* stream);

/* nn.Embedding forward (bpr.py:77-81 _user_id_lookup/_item_id_lookup; slsh bucket gather):
 *   out[b,:] = W[ids[b],:]                                                              */
int mi_oov_gather_rows(const int64_t* ids, int64_t B,
                       const float* W, int64_t N, int64_t D,
                       float* out, void* stream);

/* K queued batches of mi_oov_gather_rows in ONE launch (every batch B ids): ids_tab / out_tab are DEVICE arrays of K
 * device pointers (int64[B] / f32[B,D]); D a multiple of 4, W and the output rows 16-byte aligned.                  */
int mi_oov_gather_rows_multi(const int64_t* const* ids_tab, float* const* out_tab, int64_t K, int64_t B,
                             const float* W, int64_t N, int64_t D, void* stream);

/* BPR.get_*_embedding splice for plugins other than lsh (bpr.py:62-76,108-123):
 *   out[b,:] = table[ids[b],:]              if ids[b] < n_vocab
 *            = oov_rows[rank(b),:]          otherwise, rank(b) = number of OOV ids before b
 *   oov_rank i64[B] supplies rank(b) (host mirror computes it with an exclusive scan) */
int mi_oov_splice_rows(const int64_t* ids, const int64_t* oov_rank, int64_t B,
                       const float* table, int64_t n_vocab,
                       const float* oov_rows, int64_t n_oov, int64_t D,
                       float* out, void* stream);

/* Context models (DCNV2 / WideDeep / xDeepFM): the fused-table token gather with the OOV splice.
 *   InductiveContextRecommender.embed_token_fields   (R/model/abstract_recommender.py:794-842)
 *   InductiveFMFirstOrderLinear.embed_token_fields   (R/model/layers.py:1634-1693)
 *   out[b,f,:] = table[tokens[b,f] + offsets[f], :], except
 *       f == 0 and tokens[b,0] >= n_users  ->  oov_user_rows[user_rank[b], :]
 *       f == 1 and tokens[b,1] >= n_items  ->  oov_item_rows[item_rank[b], :]
 *   (rank = number of OOV users / items before row b; the rows come from the plugin's
 *   embed_*_ids or from the OOV bucket tables via the mapper, as in the reference).
 *   sum_fields != 0: the first-order form, out[b,:] = sum_f of those rows in field order ([B,D]).
 *   tokens i64[B,nf]  offsets i64[nf]  table f32[T,D]  out f32[B,nf,D] or f32[B,D]              */
int mi_oov_token_fields_embed(const int64_t* tokens, int64_t B, int64_t nf, const int64_t* offsets,
                              const float* table, int64_t T, int64_t D, int64_t n_users, int64_t n_items,
                              const float* oov_user_rows, const int64_t* user_rank, int64_t n_oov_users,
                              const float* oov_item_rows, const int64_t* item_rank, int64_t n_oov_items,
                              int sum_fields, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * mean / zero: MeanEmbedder (R/inductive/mean_embedder.py:41-87), ZeroEmbedder (zero_embedder.py:36-60)
 *   mi_oov_col_mean      : mean[d] = (1/N) sum_n W[n,d]   (two-pass, deterministic:
 *                          fixed row partition, partials summed in partition order)
 *                          workspace f32[mi_oov_col_mean_workspace(N,D)] scratch
 *   mi_oov_broadcast_rows: out[b,:] = vec[:]  (vec NULL -> zeros)
 * ------------------------------------------------------------------------------------------ */
int64_t mi_oov_col_mean_workspace(int64_t N, int64_t D); /* number of floats */
int mi_oov_col_mean(const float* W, int64_t N, int64_t D,
                    float* mean, float* workspace, void* stream);
int mi_oov_broadcast_rows(const float* vec, int64_t B, int64_t D, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * scoring: BPR.predict (bpr.py:145-149)  score[b] = sum_d (u[b,d]*e[b,d])  (mul, then sum)
 *          BPR.full_sort_predict / ind_full_sort_predict (bpr.py:151-163)
 *              scores[b,n] = sum_d U[b,d]*E[n,d]    -> f32[B,N] (caller views it [B*N])
 *          computed on the f32 MFMA (v_mfma_f32_32x32x2_f32): per element an fmaf chain in
 *          increasing d starting from +0, d zero-padded to a multiple of 32 (exact f32, no
 *          reduced precision).
 * ------------------------------------------------------------------------------------------ */
int mi_oov_rowdot(const float* U, const float* E, int64_t B, int64_t D, float* score, void* stream);
int mi_oov_full_sort_scores(const float* U, int64_t B, const float* E, int64_t N, int64_t D,
                            float* scores, void* stream);

/* Linear layer of the dhe / fdhe / dnn hash nets (R/inductive/dh_embedder.py:70-89,
 * feat_dh_embedder.py:108-127, dnn_embedder.py:65-90):  Y = act(X W^T + bias)
 *   X f32[B,K]   W f32[N_out,K] (nn.Linear.weight)   bias f32[N_out]   Y f32[B,N_out]
 *   act: 0 identity, 1 nn.GELU() (erf form), 2 nn.Sigmoid()
 * Same f32-MFMA tiling as the scoring kernel: per element an fmaf chain over increasing k from +0
 * (k zero-padded to a multiple of 32), then + bias, then the activation.                      */
enum { MI_OOV_ACT_NONE = 0, MI_OOV_ACT_GELU = 1, MI_OOV_ACT_SIGMOID = 2 };
int mi_oov_linear_act(const float* X, int64_t B, int64_t K, const float* W, const float* bias,
                      int64_t N_out, int act, float* Y, void* stream);

/* The same layer on the bf16 matrix cores at f32 accuracy (csrc/linear3.hip; what the hash nets run on,
 * dh_embedder.py:140-170 -> 70-89).  Every f32 operand is held as three bf16 values h + m + l (= the f32 value exactly)
 * and six of the nine cross products are accumulated in f32 on v_mfma_f32_16x16x32_bf16 (two planes side by side in
 * the instruction's 32 k); the three left out are below 2^-24 |x||w| each.  Error against the exact dot product: that of an f32 accumulation -- measured not larger than the
 * f32 chain's of mi_oov_linear_act on zero-mean, same-sign and wide-range operands -- but NOT the
 * oracle's summation order: parity is within the tolerances written in tests/test_gpu_parity.py::test_linear_x3_*
 * (8 * 2^-24 * (sum|x||w| + |b|) against the oracle on zero-mean operands), not bit for bit.  Finite operands only: where an
 * operand is infinite, NaN or above the largest bf16 (3.39e38) the result is NaN (the f32 product: +-inf or NaN).
 *   mi_oov_linear_x3_weights_bytes(N_out, K)   bytes of the split weights (three planes, rows padded to 256, K to 16)
 *   mi_oov_linear_x3_prepare(W, N_out, K, wsplit, stream)     W f32[N_out,K] -> wsplit (16-byte aligned); once per
 *                                                             weight update (a few microseconds)
 *   mi_oov_linear_x3(X, B, K, wsplit, bias, N_out, act, Y, stream)   Y f32[B,N_out] = act(X W^T + bias); X is split
 *                                                             inside the kernel while it is staged.
 * K here is the row length of X: the K the weights were prepared with, or that rounded up to a multiple of 16 -- the
 * columns beyond the weights' K meet the zeros the split is padded with (they must hold finite values).  Rows of a
 * multiple of 16 floats, 16-byte aligned, K >= 32, N_out > 128 and enough rows to fill half of the CUs with 256 x 256
 * tiles take the pipelined kernel (one persistent workgroup per CU, weights by LDS-DMA); everything else a generic tile
 * kernel of the same arithmetic in the same order: a row's result does not depend on the batch it is computed in.   */
int64_t mi_oov_linear_x3_weights_bytes(int64_t N_out, int64_t K);
int mi_oov_linear_x3_prepare(const float* W, int64_t N_out, int64_t K, void* wsplit, void* stream);
/* the same split from the TRANSPOSE of the weights, Wt f32[K,N_out] row-major (training: the operands of dW = dZ^T X,
 * db = 1^T dZ and dX = dZ W lie that way; no transposed copy is made first) */
int mi_oov_linear_x3_prepare_t(const float* Wt, int64_t N_out, int64_t K, void* wsplit, void* stream);
int mi_oov_linear_x3(const float* X, int64_t B, int64_t K, const void* wsplit, const float* bias,
                     int64_t N_out, int act, float* Y, void* stream);
/* The same product with K cut into ksplit shares (1 .. 65535) that run side by side -- for the shapes of the hash nets'
 * TRAINING (dh_embedder.py:191-217 under autograd): dW = dZ^T X has a few output tiles and K = the batch, one workgroup
 * per tile would walk it alone.  Each share leaves its sums in a slab of workspace
 * (mi_oov_linear_x3_splitk_workspace(B, N_out, ksplit) bytes, 16-byte aligned), a second kernel adds the slabs in
 * order, then bias and activation: deterministic, but NOT the un-split entry's rounding (a result depends on ksplit). */
int64_t mi_oov_linear_x3_splitk_workspace(int64_t B, int64_t N_out, int64_t ksplit);
int mi_oov_linear_x3_splitk(const float* X, int64_t B, int64_t K, const void* wsplit, const float* bias,
                            int64_t N_out, int act, float* Y, int64_t ksplit, void* workspace, void* stream);

/* Training of the hash nets on this library's GEMM (csrc/mlp.hip): what torch autograd does for the reference's
 * nn.Sequential(Linear, GELU, ..., Linear, Sigmoid) (dh_embedder.py:70-89,191-217, dnn_embedder.py:65-109).
 *   forward   Z = mi_oov_linear_act(X, W, b, MI_OOV_ACT_NONE);  Y = mi_oov_act_forward(Z, act)      (Z is kept)
 *   backward  dZ = mi_oov_act_backward(dY, Z, act)                       dY * act'(Z)
 *             dX = mi_oov_full_sort_scores(dZ,   W^T)                    [B,out] x [in,out]^T -> [B,in]
 *             dW = mi_oov_full_sort_scores(dZ^T, X^T)                    [out,B] x [in,B]^T   -> [out,in]
 *             db = mi_oov_full_sort_scores(1[1,B], dZ^T)                 column sums of dZ
 *   with the layouts made by mi_oov_transpose (At[c,r] = A[r,c], A f32[R,C]).
 * n = number of elements; pointers 16-byte aligned; act as in mi_oov_linear_act.                        */
int mi_oov_act_forward(const float* Z, int64_t n, int act, float* Y, void* stream);
int mi_oov_act_backward(const float* dY, const float* Z, int64_t n, int act, float* dZ, void* stream);
int mi_oov_transpose(const float* A, int64_t R, int64_t C, float* At, void* stream);

/* Fused full-sort score + per-row top-k (the [B,N] matrix is never written).  Serves
 *   - the evaluator's torch.topk(scores, k) (R/evaluator/collector.py:158-167), and
 *   - the exact kNN search standing in for ScaNN (R/inductive/knn_embedder.py:100-102).
 *   vals f32[B,k] descending, idx i64[B,k]; ties broken towards the LOWER index.
 *   n_skip_low: columns [0, n_skip_low) are excluded (the evaluator masks padding item 0).
 *   workspace: mi_oov_score_topk_workspace(B,N,k) bytes.                               */
int64_t mi_oov_score_topk_workspace(int64_t B, int64_t N, int64_t k);
/* the same for a known row width D (<= 64: one k-half of bf16 copies instead of the two the query above keeps room for;
 * at 10 M rows the difference is 1.28 GB) -- sufficient for mi_oov_score_topk with that D */
int64_t mi_oov_score_topk_workspace_d(int64_t B, int64_t N, int64_t D, int64_t k);
int mi_oov_score_topk(const float* U, int64_t B, const float* E, int64_t N, int64_t D,
                      int64_t k, int64_t n_skip_low,
                      float* vals, int64_t* idx, void* workspace, void* stream);

/* Evaluation (R/inductive/evaluator.py:118-134 neg_sample_batch_eval -> R/evaluator/collector.py:158-167): the
 * reference scatters the sampled scores of a batch into a dense [users, items] matrix of -inf and calls
 * torch.topk; here segment s (one user of the batch) owns candidates [seg_ptr[s], seg_ptr[s+1]) given as
 * (scores[i], cols[i]) and the dense matrix never exists.  Best k (<= 256) candidates with col_lo <= column <
 * col_hi, larger score first, NaN highest, ties -> earlier candidate; (-inf, -1) where fewer than k qualify.
 *   scores f32[M], cols i64[M], seg_ptr i64[S+1] -> vals f32[S,k], idx i64[S,k] (item columns).
 * Columns are item ids (>= 0).  With the whole range (col_lo <= 0 and col_hi >= 2^62) only the winners' columns
 * are read: 4 instead of 12 bytes per candidate.                                                                 */
int mi_oov_segment_topk(const float* scores, const int64_t* cols, const int64_t* seg_ptr, int64_t S, int64_t k,
                        int64_t col_lo, int64_t col_hi, float* vals, int64_t* idx, void* stream);

/* The collector's "rec.topk" block (collector.py:161-166): out[s,j] = 1 iff idx[s,j] is among the positives of
 * segment s (CSR pos_ptr i64[S+1], pos_cols i64[nnz]); out[s,k] = number of positives.  out i32[S,k+1].        */
int mi_oov_topk_hits(const int64_t* idx, int64_t S, int64_t k, const int64_t* pos_ptr, const int64_t* pos_cols,
                     int32_t* out, void* stream);

/* ... with a column range on the POSITIVES: positives outside [col_lo, col_hi) neither count in out[s,k] nor match -- the
 * old-item / new-item slices of the filtered collectors (R/inductive/collector_filter.py:128-256) without compacting the
 * positive lists.  mi_oov_topk_hits = the whole range.                                                             */
int mi_oov_topk_hits_range(const int64_t* idx, int64_t S, int64_t k, const int64_t* pos_ptr, const int64_t* pos_cols,
                           int64_t col_lo, int64_t col_hi, int32_t* out, void* stream);

/* The rows of NegSampleEvalDataLoader's batches (R/data/dataloader/general_dataloader.py:157-190,
 * abstract_dataloader.py:227-235: per user its positives first, then n_neg sampled items per positive) for a GROUP of
 * consecutive batches, in one pass: user u of the group (u < n_users) has positives pos_items[pos_ptr[u] .. pos_ptr[u+1])
 * and negatives neg_items[pos_ptr[u] * n_neg .. pos_ptr[u+1] * n_neg) (the negatives of a batch are drawn as one array in
 * user order, so consecutive batches concatenate to exactly this layout); its rows are
 * [seg_ptr[u], seg_ptr[u+1]), seg_ptr[u] = pos_ptr[u] * (1 + n_neg).
 *   pos_ptr i64[n_users+1] (pos_ptr[0] = 0), user_ids i64[n_users], pos_items i64[P], neg_items i64[P * n_neg]
 *   -> row_user i64[M], row_item i64[M] (M = P (1 + n_neg): what model.predict is handed), seg_ptr i64[n_users+1],
 *      pos_user i64[P] or NULL (index of the user of every positive).                                               */
int mi_oov_eval_rows_build(const int64_t* pos_ptr, int64_t n_users, const int64_t* user_ids, const int64_t* pos_items,
                           const int64_t* neg_items, int64_t n_neg, int64_t* row_user, int64_t* row_item,
                           int64_t* seg_ptr, int64_t* pos_user, void* stream);

/* The reference scatters a batch's scores into a dense matrix, `scores[row_idx, col_idx] = origin_scores`
 * (R/inductive/evaluator.py:118-134): a (user, item) pair that occurs twice keeps ONE entry.  out[i] = cols[i] where
 * candidate i is the first of its segment with that column, -1 where an earlier one has it (mi_oov_segment_topk skips
 * negative columns whenever it is given a column range, e.g. [0, 2^62 - 1)).  cols i64[M], seg_ptr i64[S+1] -> out i64[M];
 * out must not be cols (MI_OOV_ERR_ALIAS).  Any segment length, no workspace.                                      */
int mi_oov_segment_dedup(const int64_t* cols, const int64_t* seg_ptr, int64_t S, int64_t* out, void* stream);

/* The TopkMetric family (R/evaluator/metrics.py:36-235, base_metric.py:60-84) on a rec.topk block rec i32[U, K+1]
 * (mi_oov_topk_hits): per-user curves for k = 1..K in float64, summed over users IN USER ORDER -- the order of NumPy's
 * reduction over the leading axis, so sums[...] / counts[...] are the float64 means the reference's Evaluator computes, bit
 * for bit.  disc f64[K] = 1 / log2(rank + 1) and idcg_base f64[K] = its cumulative sum, both computed by the caller with
 * NumPy (the reference's expressions) so that no libm ulp differs.  Metric m: 0 recall, 1 hit, 2 precision, 3 ndcg, 4 mrr,
 * 5 map.  n_sides 1: all users; 3: all users, users with uids[u] < n_old_users, the others (uids i64[U]).
 *   -> sums f64[n_sides, 6, K], counts i64[n_sides, 6] (users whose curve holds no NaN: recall drops users without a positive)
 *   workspace: mi_oov_topk_metric_sums_workspace(U, K) bytes, 8-byte aligned.  K <= 256.                           */
int64_t mi_oov_topk_metric_sums_workspace(int64_t U, int64_t K);
int mi_oov_topk_metric_sums(const int32_t* rec, int64_t U, int64_t K, const double* disc, const double* idcg_base,
                            const int64_t* uids, int64_t n_old_users, int n_sides, double* sums, int64_t* counts,
                            void* workspace, void* stream);

/* mi_oov_score_topk with per-user exclusions: what the collector's topk sees after InductiveEvaluator.eval_batch has
 * set scores[:,0] and scores[history_index] to -inf (R/inductive/evaluator.py:92-95).  excl_ptr i64[B+1] / excl_cols
 * i64[nnz] is the CSR of excluded columns, ASCENDING within a row; h_max >= the longest row; k + h_max <= 256
 * (longer histories: score in chunks of users, or materialise with mi_oov_full_sort_scores).  Top-(k + h_max) through
 * the fused two-pass kernel, then one filter pass; nothing [B,N]-sized is written.
 *   workspace: mi_oov_score_topk_excl_workspace(B, N, k, h_max) bytes, 16-byte aligned.                         */
int64_t mi_oov_score_topk_excl_workspace(int64_t B, int64_t N, int64_t k, int64_t h_max);
int mi_oov_score_topk_excl(const float* U, int64_t B, const float* E, int64_t N, int64_t D, int64_t k,
                           int64_t n_skip_low, const int64_t* excl_ptr, const int64_t* excl_cols, int64_t h_max,
                           float* vals, int64_t* idx, void* workspace, void* stream);

/* The same result for EVERY shape (any D, any k, any history length, catalogues too small for the fused path): scores of
 * a chunk of users materialised in the workspace (mi_oov_full_sort_scores' kernel), the exclusion bitmap built beside
 * them, an exact radix select that skips excluded columns -- the route of last resort, all inside the library and without
 * a host synchronisation.  excl_cols need not be sorted; entries outside [0, N) are ignored; missing entries (-inf, -1).
 *   workspace: mi_oov_score_topk_excl_dense_workspace(B, N) bytes, 16-byte aligned.                                 */
int64_t mi_oov_score_topk_excl_dense_workspace(int64_t B, int64_t N);
int mi_oov_score_topk_excl_dense(const float* U, int64_t B, const float* E, int64_t N, int64_t D, int64_t k,
                                 int64_t n_skip_low, const int64_t* excl_ptr, const int64_t* excl_cols,
                                 float* vals, int64_t* idx, void* workspace, void* stream);

/* The same result with exclusion lists of ANY length, still without a [B,N] matrix: the CSR becomes a bitmap of
 * B x ceil(N/64) words inside the workspace; the first pass of the fused kernel leaves excluded columns out of its tile
 * maxima (so its bound is the k-th best ALLOWED one and the candidate count does not grow with the histories), the
 * second pass drops them when it emits a candidate.  Rows of D <= 128 floats (16-byte aligned when D = 64 or 128; other widths
 * are zero-padded to 64 or 128 in the library's bf16 copies), k <= 256, N >= 128 k (the fused bf16 path); the workspace query returns 0 for shapes it does not take (use mi_oov_score_topk_excl or materialise).
 * excl_cols need not be sorted; entries outside [0, N) are ignored.
 *   workspace: mi_oov_score_topk_masked_workspace(B, N, D, k) bytes, 16-byte aligned.                            */
int64_t mi_oov_score_topk_masked_workspace(int64_t B, int64_t N, int64_t D, int64_t k);
int mi_oov_score_topk_masked(const float* U, int64_t B, const float* E, int64_t N, int64_t D, int64_t k,
                             int64_t n_skip_low, const int64_t* excl_ptr, const int64_t* excl_cols,
                             float* vals, int64_t* idx, void* workspace, void* stream);

/* A catalogue E that many user batches are scored against (the knn search's feature table -- the reference builds a ScaNN
 * searcher over it at construction, knn_embedder.py:84-93 -- or the item table of an evaluation run) can be prepared once:
 * the buffer receives what the fused bf16 path otherwise derives from E on every call.  D <= 128 (E 16-byte aligned when D = 64 or 128);
 * mi_oov_topk_catalogue_bytes returns 0 for other shapes.  The catalogue is valid for exactly the E it was made from.
 * mi_oov_score_topk_prepared = mi_oov_score_topk (excl_ptr NULL; workspace mi_oov_score_topk_workspace) or
 * mi_oov_score_topk_masked (excl_ptr given; workspace mi_oov_score_topk_masked_workspace) with the per-call pass over E
 * left out; results are identical.                                                                                */
int64_t mi_oov_topk_catalogue_bytes(int64_t N, int64_t D);
int mi_oov_topk_catalogue_prepare(const float* E, int64_t N, int64_t D, void* catalogue, void* stream);
int mi_oov_score_topk_prepared(const float* U, int64_t B, const float* E, int64_t N, int64_t D, int64_t k,
                               int64_t n_skip_low, const int64_t* excl_ptr, const int64_t* excl_cols,
                               const void* catalogue, float* vals, int64_t* idx, void* workspace, void* stream);
/* bytes of workspace for mi_oov_score_topk_prepared (masked: with an exclusion list): the lists for this row width only
 * -- the catalogue holds the bf16 copy of E, so a 10 M-row search of 4096 users fits one call (0 where the shape is not
 * taken, as mi_oov_score_topk_masked_workspace) */
int64_t mi_oov_score_topk_prepared_workspace(int64_t B, int64_t N, int64_t D, int64_t k, int masked);

#ifdef __cplusplus
}
#endif
#endif /* MI_OOV_H */
