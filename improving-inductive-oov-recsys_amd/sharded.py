"""Row-sharded tables across the GPUs of one node: new functionality, the
reference has no table sharding (its only multi-GPU mode is DDP replicas, R/trainer/trainer.py:68-72).

One process per GPU, `torch.distributed` ("nccl" IS RCCL on ROCm; xGMI is point-to-point, an all-to-all drives
all 7 links of a GPU at once, no ring).  Rank r owns the contiguous row block shard_bounds(N, world, r) of the big
table; the small state (planes, lsh bucket table, the reachable window of an slsh bucket table) is replicated.

A sharded lookup is OWNER-COMPUTES with the smallest possible payload:

    requester   mi_oov_bucket_by_owner   ids -> fixed-capacity per-owner segments of local row numbers + slot[b]
                all_to_all_single        8 B per lookup out
    owner       mi_oov_lsh_embed (bits)  feature row -> H sign bits          (slsh: mi_oov_slsh_embed -> bucket id)
                all_to_all_single        H B (slsh: 8 B) per lookup back -- never the F-wide row, never the D-wide one
    requester   mi_oov_lsh_codes_embed   code at slot[b] -> masked mean of the REPLICATED bucket rows (+ score)

Every lookup's code is computed by exactly one owner with the single-GPU kernel and aggregated with the same
arithmetic, so results are bit-identical to the unsharded kernels.  Segments have a fixed capacity: the collectives
take no sizes from the device and a step contains NO host synchronisation (tests/test_gpu_sharded.py runs one under
torch.cuda.set_sync_debug_mode("error")).  capacity = B (the default) can never overflow; a tighter
`cap_factor` (bench.py) drops lookups of an overfull segment (NaN) and records it in `overflow` for the caller.

`ShardedEmbeddingTable` shards the D-wide tables themselves (item / user embedding tables, big slsh bucket tables): the
owner gathers the rows it holds and the D-wide rows DO cross the links -- there is nothing smaller to send when the row
is the answer -- for BPR's in-vocabulary gather, the knn aggregate and the full-catalogue top-k (per-shard fused top-k
with global item numbers + `merge_topk`).  Lookups that the requester owns itself never enter an exchange (local fast
path): `ShardedEmbeddingTable.gather` serves them from its own block inside the same splice kernel, and
`ShardedLSHTable.embed_score(..., local_fast=True)` answers them with the fused single-GPU kernel while the exchange of the
remote ones is in flight.

`LshPipeline` issues consecutive steps software-pipelined so that both exchanges of a step hide under the kernels of
its neighbours.  The local compute is injectable (`prims`): the CPU tests (gloo, world 2) run the exchange logic
with the oracle in its place; the product default is the HIP kernels and nothing else.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_rows: int, world: int, rank: int):
    """Contiguous block partition: rank r owns rows [r*per, min(n, (r+1)*per)), per = ceil(n/world)."""
    per = (n_rows + world - 1) // world
    lo = min(n_rows, rank * per)
    hi = min(n_rows, lo + per)
    return lo, hi, per


class HipPrims:
    """The local compute of the exchange: the HIP kernels (the only product implementation)."""

    @staticmethod
    def bucket(ids, n_rows, per, world, cap, overflow):
        from . import ops
        return ops.bucket_by_owner(ids, n_rows, per, world, cap, overflow)

    @staticmethod
    def bucket_local(ids, n_rows, per, world, cap, overflow, my_rank):
        """bucket + the lookups of my_rank compacted into local_rows (slots world * cap + position): one launch."""
        from . import ops
        return ops.bucket_by_owner(ids, n_rows, per, world, cap, overflow, my_rank=my_rank)

    @staticmethod
    def codes(local_ids, feat_local, planes, out=None):
        from . import ops
        if feat_local.shape[0] == 0:  # more ranks than rows: this rank owns nothing and is never asked
            return torch.full((local_ids.numel(), planes.shape[0]), 255, dtype=torch.uint8, device=local_ids.device)
        return ops.lsh_bits(local_ids, feat_local, planes, out=out)

    @staticmethod
    def codes_embed(codes, slot, buckets, other, want_emb, score_out=None):
        from . import ops
        return ops.lsh_codes_embed(codes, slot, buckets, other, want_emb=want_emb, score_out=score_out)

    @staticmethod
    def slsh_index(local_ids, feat_local, planes, n_buckets):
        from . import ops
        if feat_local.shape[0] == 0:
            return torch.full((local_ids.numel(),), -1, dtype=torch.int64, device=local_ids.device)
        return ops.slsh_index(local_ids, feat_local, planes, n_buckets)

    @staticmethod
    def gather_rows(idx, table):
        from . import ops
        if table.shape[0] == 0:  # a rank that owns nothing answers NaN rows (it is never asked for a valid one)
            return torch.full((idx.numel(), table.shape[1]), float("nan"), dtype=torch.float32, device=idx.device)
        return ops.gather_rows(idx, table)

    @staticmethod
    def splice(key, slot, table_local, back):
        """out[b] = table_local[key[b]] where 0 <= key[b] < n_local (the requester's own rows), else back[slot[b]] (the
        owners' answers), NaN where neither exists: mi_oov_splice_rows with the answers in the place of the OOV rows."""
        from . import ops
        return ops._splice_forward(key, slot, table_local, back)

    @staticmethod
    def gather_mean(rows, g):
        """mean over consecutive groups of g rows, summed in position order (the knn aggregate on gathered rows)."""
        from . import ops
        idx = torch.arange(rows.shape[0], dtype=torch.int64, device=rows.device)
        return ops.gather_mean(idx, rows, g)

    @staticmethod
    def score_topk(U, E_local, k, n_skip_low):
        from . import ops
        return ops.score_topk(U, E_local, k, n_skip_low)

    @staticmethod
    def lsh_embed_score(ids_local, feat_local, planes, buckets, other, score_out=None):
        from . import ops
        return ops.lsh_embed_score(ids_local, feat_local, planes, buckets, other, score_out=score_out)


def _backend(group):
    return dist.get_backend(group)


def _all_to_all(out, inp, group, async_op=False):
    """Equal-split all_to_all_single.  RCCL takes device tensors; gloo (CPU tests, 1-GPU rehearsals) takes host
    tensors only, so device tensors are staged through the host there -- a rehearsal path, never the product's."""
    if inp.is_cuda and _backend(group) != "nccl":
        h_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(h_out, inp.cpu(), group=group)
        out.copy_(h_out)
        return None
    return dist.all_to_all_single(out, inp, group=group, async_op=async_op)


class _Pending:
    """One lookup batch on its way through the exchange."""
    __slots__ = ("B", "cap", "slot", "counts", "recv", "back", "w_ids", "w_back", "width", "dtype", "back_ext")


class _ShardedBase:
    def __init__(self, feat_local, n_rows_global, group=None, prims=None, cap_factor=None, max_batch=None,
                 uniform_batches=False):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_rows = n_rows_global
        self.lo, self.hi, self.per = shard_bounds(n_rows_global, self.world, self.rank)
        if feat_local.shape[0] != self.hi - self.lo:
            raise ValueError(f"rank {self.rank} must hold rows [{self.lo},{self.hi}), got {feat_local.shape[0]}")
        self.feat_local = feat_local
        self.prims = prims or HipPrims
        self.cap_factor = cap_factor
        self.max_batch = max_batch
        self.uniform_batches = uniform_batches
        # largest excess of a segment over its capacity so far (updated by the bucket kernel, read by check_overflow)
        self.overflow = torch.zeros((1,), dtype=torch.int32, device=feat_local.device)

    def capacity(self, B):
        """Entries per (requester, owner) segment for batches of up to B lookups per rank: B by default (cannot
        overflow).  With cap_factor -- for id streams known to be spread evenly over the table (bench.py) -- the
        expected share B / world times that factor plus eight standard deviations of a binomial share, rounded up
        to 64."""
        if self.cap_factor is None or self.world == 1:
            return max(1, B)
        share = -(-B // self.world)
        want = int(share * self.cap_factor + 8.0 * share ** 0.5)
        return max(64, min(B, -(-want // 64) * 64))

    def _agreed_capacity(self, B):
        """Every rank must use the same segment size.  `uniform_batches` (every rank passes the same number of lookups
        to the same call: a benchmark, a lock-step serving loop) or `max_batch` (an upper bound of the per-rank batch
        that all ranks were constructed with; segments are then sized for it) make that a local computation and the
        step stays free of host synchronisation; without either the ranks agree on max(B) with one tiny all_reduce
        whose result the host reads."""
        if self.uniform_batches:
            return self.capacity(B)
        if self.max_batch is not None:
            if B > self.max_batch:
                raise ValueError(f"batch of {B} lookups exceeds max_batch={self.max_batch}")
            return self.capacity(self.max_batch)
        if self.world == 1:
            return self.capacity(B)
        t = torch.tensor([B], dtype=torch.int64, device=self.feat_local.device if _backend(self.group) == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return self.capacity(int(t.item()))

    def can_compact_local(self, ids):
        """The one-launch bucketing with the local share compacted (mi_oov_bucket_by_owner_fused) serves single-node
        worlds; a rank that owns no rows, or an empty batch, has nothing to compact."""
        return hasattr(self.prims, "bucket_local") and self.world <= 16 and self.hi > self.lo and ids.numel() > 0

    def begin(self, ids, async_op=False, local_planes=None):
        """Bucket `ids` (global rows, int64[B]) by owner and send every owner its segment.
        local_planes (lsh, `can_compact_local`): the lookups whose row this rank owns are NOT sent -- the bucketing kernel
        compacts them, their codes are computed here (1 / world of a uniform batch, while the remote ids travel) and stored
        behind the world * cap exchanged codes, where their slots point."""
        p = _Pending()
        p.B = ids.numel()
        p.cap = self._agreed_capacity(p.B)
        p.back_ext = None
        if local_planes is not None:
            send, p.slot, p.counts, local_rows = self.prims.bucket_local(ids, self.n_rows, self.per, self.world, p.cap, self.overflow,
                                                                         self.rank)
        else:
            send, p.slot, p.counts = self.prims.bucket(ids, self.n_rows, self.per, self.world, p.cap, self.overflow)
        p.recv = torch.empty_like(send)
        p.w_ids = _all_to_all(p.recv, send, self.group, async_op)
        p.w_back = None
        if local_planes is not None:
            H = local_planes.shape[0]
            p.back_ext = torch.empty(((self.world + 1) * p.cap, H), dtype=torch.uint8, device=send.device)
            tail = p.back_ext[self.world * p.cap:]
            got = self.prims.codes(local_rows, self.feat_local, local_planes, out=tail)
            if got.data_ptr() != tail.data_ptr():  # (a prims without an `out`: the oracle of the CPU tests)
                tail.copy_(got)
        return p

    def _reply(self, p, payload, async_op):
        """Send the owners' answers (one fixed-width record per received id) back to the requesters."""
        p.back = torch.empty_like(payload)
        p.w_back = _all_to_all(p.back, payload, self.group, async_op)

    @staticmethod
    def _wait(work):
        if work is not None:
            work.wait()

    def check_overflow(self):
        """Host check (synchronises): raises if any segment ever overflowed its capacity."""
        n = int(self.overflow.item())
        if n:
            raise RuntimeError(f"sharded exchange dropped lookups: a segment exceeded its capacity by {n}; "
                               "use cap_factor=None (capacity = batch) or a larger factor")


class ShardedLSHTable(_ShardedBase):
    """The local block of a row-sharded lsh feature matrix; planes and the lsh bucket table are replicated.

    feat_local: float32[hi-lo, F] rows [lo, hi) of the global matrix, resident on this rank's GPU.
    embed(ids, planes, buckets) == ops.lsh_embed on the unsharded matrix, bit for bit (global ids, any owner);
    embed_score(...) == ops.lsh_embed_score."""

    def owner(self, p, planes, async_op=False):
        """Owner side: sign bits of the rows this rank was asked for (0xFF rows for the -1 padding), sent back."""
        self._wait(p.w_ids)
        codes = self.prims.codes(p.recv.view(-1), self.feat_local, planes)
        if p.back_ext is None:
            self._reply(p, codes, async_op)
        else:  # the exchanged codes land in front of the local share's (begin)
            p.back = p.back_ext[: self.world * p.cap]
            p.w_back = _all_to_all(p.back, codes, self.group, async_op)

    def finish(self, p, buckets, other=None, want_emb=True, score_out=None):
        self._wait(p.w_back)
        return self.prims.codes_embed(p.back if p.back_ext is None else p.back_ext, p.slot, buckets, other, want_emb, score_out)

    def embed(self, ids, planes, buckets):
        p = self.begin(ids)
        self.owner(p, planes)
        return self.finish(p, buckets)[1]

    def embed_score(self, ids, planes, buckets, other, score_out=None, local_fast=False):
        """local_fast: lookups whose row this rank owns skip both collectives -- they are answered by the fused
        single-GPU kernel on the local block (ids outside it give NaN there and are filled in from the exchange), which
        runs while the ids of the remote lookups are on the wire; remote and local answers are the same arithmetic on the
        same bits, so the result is unchanged.  1 / world of a uniform batch is local."""
        if not local_fast or self.hi == self.lo:
            p = self.begin(ids)
            self.owner(p, planes)
            return self.finish(p, buckets, other, want_emb=False, score_out=score_out)[0]
        if self.can_compact_local(ids):  # round 4: the local share compacted by the bucketing kernel (1 / world of the batch)
            p = self.begin(ids, async_op=_backend(self.group) == "nccl", local_planes=planes)
            self.owner(p, planes)
            return self.finish(p, buckets, other, want_emb=False, score_out=score_out)[0]
        local = (ids >= self.lo) & (ids < self.hi)
        p = self.begin(torch.where(local, torch.full_like(ids, -1), ids), async_op=_backend(self.group) == "nccl")
        sc_local = self.prims.lsh_embed_score(ids - self.lo, self.feat_local, planes, buckets, other)  # NaN off the block
        self.owner(p, planes)
        sc = self.finish(p, buckets, other, want_emb=False)[0]
        out = torch.where(local, sc_local, sc)
        if score_out is not None:
            score_out.copy_(out)
            return score_out
        return out


class LshPipeline:
    """Consecutive sharded embed_score steps, three in flight, so that neither exchange is exposed:

        step t:   wait ids_t -> owner_t -> send codes_t | finish_{t-1} (codes_{t-1} arrived long ago)
                  | bucket_{t+2} -> send ids_{t+2}
    The collectives are asynchronous (RCCL's own stream) and a kernel only ever waits for an exchange issued a whole
    step earlier.

    `streams=True` (the default on a ROCm device with the RCCL backend) puts the three stages on three streams: the
    owner kernel of step t (random 256-B rows of the local shard), the requester kernel of step t-1 (the sequential
    rows of the other side) and the bucketing of step t+2 are independent, and run side by side they share HBM the way
    the fused single-GPU kernel's gather and stream do -- the owner kernel keeps 2 waves per SIMD at 111 VGPRs, the
    requester kernel 2 at 92, 72 KiB of LDS together, so both are resident on every CU.  Tensors that cross streams
    are recorded on the stream that reads them (torch's allocator then keeps their memory until that stream has
    passed); the caller's stream waits for all three at the end of `run`."""

    def __init__(self, table, planes, buckets, streams=None, local_fast=False):
        """local_fast: the lookups of a step whose row this rank owns (1 / world of a uniform batch) skip both collectives:
        they are masked out of the exchange and scored by the fused single-GPU kernel on the local block (issued with
        the step's bucketing, so it runs while the remote ids travel); the requester stage selects per lookup.  Same
        arithmetic on the same bits either way.  The fused kernel walks the WHOLE batch to find its share, so this pays
        in latency (a short run is one serial chain of exchanges), not in throughput."""
        self.table, self.planes, self.buckets = table, planes, buckets
        self.local_fast = bool(local_fast) and table.hi > table.lo
        dev = table.feat_local.device
        if streams is None:
            streams = dev.type == "cuda" and _backend(table.group) == "nccl"
        self.streams = [torch.cuda.Stream(dev) for _ in range(3)] if streams else None

    def run(self, ids, others, scores):
        """ids[t] int64[M], others[t] f32[M,D], scores[t] f32[M] (written).  No host synchronisation."""
        t_, n = self.table, len(ids)
        if n == 0:
            return
        def begin_step(t):
            if not self.local_fast:
                return t_.begin(ids[t], async_op=True), None, None
            if t_.can_compact_local(ids[t]):  # the bucketing kernel compacts the local share; its codes join the exchanged ones
                return t_.begin(ids[t], async_op=True, local_planes=self.planes), None, None
            local = (ids[t] >= t_.lo) & (ids[t] < t_.hi)
            p = t_.begin(torch.where(local, torch.full_like(ids[t], -1), ids[t]), async_op=True)
            return p, local, t_.prims.lsh_embed_score(ids[t] - t_.lo, t_.feat_local, self.planes, self.buckets, others[t])

        def finish_step(p, local, sc_local, t):
            t_.finish(p, self.buckets, others[t], want_emb=False, score_out=scores[t])
            if local is not None:
                torch.where(local, sc_local, scores[t], out=scores[t])

        if self.streams is None:
            pend = [begin_step(t) for t in range(min(2, n))]
            prev = None
            for t in range(n):
                st = pend.pop(0)
                t_.owner(st[0], self.planes, async_op=True)
                if prev is not None:
                    finish_step(*prev)
                if t + 2 < n:
                    pend.append(begin_step(t + 2))
                prev = (*st, t)
            finish_step(*prev)
            return
        s_bucket, s_owner, s_req = self.streams
        caller = torch.cuda.current_stream(t_.feat_local.device)
        for s in self.streams:  # the inputs were produced on the caller's stream
            s.wait_stream(caller)

        def begin(t):
            with torch.cuda.stream(s_bucket):
                return begin_step(t)

        def owner(p):
            with torch.cuda.stream(s_owner):
                p.recv.record_stream(s_owner)  # allocated on the bucket stream, read here
                if p.back_ext is not None:     # (allocated and partly written on the bucket stream; the exchange fills the rest)
                    s_owner.wait_stream(s_bucket)
                    p.back_ext.record_stream(s_owner)
                t_.owner(p, self.planes, async_op=True)

        def finish(p, local, sc_local, t):
            with torch.cuda.stream(s_req):
                p.slot.record_stream(s_req)  # bucket stream -> here
                t_._wait(p.w_back)
                p.w_back = None
                p.back.record_stream(s_req)  # owner stream -> here
                if p.back_ext is not None:
                    p.back_ext.record_stream(s_req)
                if local is not None:
                    s_req.wait_stream(s_bucket)  # the local share was scored on the bucket stream
                    local.record_stream(s_req)
                    sc_local.record_stream(s_req)
                finish_step(p, local, sc_local, t)

        pend = [begin(t) for t in range(min(2, n))]
        prev = None
        for t in range(n):
            st = pend.pop(0)
            owner(st[0])
            if prev is not None:
                finish(*prev)
            if t + 2 < n:
                pend.append(begin(t + 2))
            prev = (*st, t)
        finish(*prev)
        for s in self.streams:  # results (and the overflow counter) are visible to the caller's stream
            caller.wait_stream(s)


class ShardedSLSHTable(_ShardedBase):
    """slsh over a row-sharded feature table (BASELINE config 4: 1e8 x 64 features + a 1e8 x 128 bucket/item table).

    The owner of the FEATURE row computes the bucket id (single_lsh_embedder.py:82-87; 8 B out, 8 B back).  The
    reference's arithmetic, idx = (bits_req + popcount) % n_buckets, reaches only rows [bits_req, 2*bits_req] of
    the bucket table, so that window (at most 65 rows) is REPLICATED on every rank and the row gather
    (single_lsh_embedder.py:100,108) is local: one exchange per batch, no D-wide row on the wire.  `window` holds
    rows [win_lo, win_lo + len) of the global bucket table (the whole table when n_buckets <= 2*bits_req + 1 and the
    ids wrap); `gather_window` builds it from row-sharded bucket blocks with one all_gather at construction."""

    def __init__(self, feat_local, n_feat_rows, window, win_lo, n_buckets, group=None, prims=None, cap_factor=None,
                 max_batch=None, uniform_batches=False):
        super().__init__(feat_local, n_feat_rows, group, prims, cap_factor, max_batch, uniform_batches)
        self.window, self.win_lo, self.n_buckets = window, win_lo, n_buckets

    @staticmethod
    def window_bounds(n_planes, n_buckets):
        """Rows of the bucket table a lookup can reach: (n_planes + popcount) % n_buckets, popcount in [0, n_planes]."""
        if 2 * n_planes < n_buckets:
            return n_planes, 2 * n_planes + 1
        return 0, n_buckets

    @staticmethod
    def gather_window(buckets_local, n_buckets, n_planes, group=None):
        """Replicate the reachable window from row-sharded bucket blocks (shard_bounds over n_buckets)."""
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        blo, bhi, _ = shard_bounds(n_buckets, world, rank)
        if buckets_local.shape[0] != bhi - blo:
            raise ValueError(f"rank {rank} must hold bucket rows [{blo},{bhi})")
        lo, hi = ShardedSLSHTable.window_bounds(n_planes, n_buckets)
        mine = torch.zeros((hi - lo, buckets_local.shape[1]), dtype=buckets_local.dtype, device=buckets_local.device)
        a, b = max(lo, blo), min(hi, bhi)
        if a < b:
            mine[a - lo:b - lo] = buckets_local[a - blo:b - blo]
        parts = [torch.empty_like(mine) for _ in range(world)]
        if mine.is_cuda and _backend(group) != "nccl":  # gloo rehearsal: through the host
            host = [torch.empty(mine.shape, dtype=mine.dtype) for _ in range(world)]
            dist.all_gather(host, mine.cpu(), group=group)
            parts = [h.to(mine.device) for h in host]
        else:
            dist.all_gather(parts, mine, group=group)
        window = torch.empty_like(mine)
        for r in range(world):  # row i of the window comes from its owner's copy (no arithmetic: bits preserved)
            rlo, rhi, _ = shard_bounds(n_buckets, world, r)
            a, b = max(lo, rlo), min(hi, rhi)
            if a < b:
                window[a - lo:b - lo] = parts[r][a - lo:b - lo]
        return window, lo

    def embed(self, ids, planes):
        """-> (rows f32[B,D], idx int64[B]); idx -1 and a NaN row for ids outside the table, as ops.slsh_embed."""
        p = self.begin(ids)
        self._wait(p.w_ids)
        idx_owner = self.prims.slsh_index(p.recv.view(-1), self.feat_local, planes, self.n_buckets)
        self._reply(p, idx_owner.view(self.world, -1), False)
        self._wait(p.w_back)
        slot = p.slot.to(torch.int64)
        idx = torch.where(slot >= 0, p.back.view(-1)[slot.clamp_min(0)], torch.full_like(slot, -1))
        local = torch.where(idx >= 0, idx - self.win_lo, idx)  # -1 stays -1: the gather answers with a NaN row
        return self.prims.gather_rows(local, self.window), idx


class ShardedEmbeddingTable(_ShardedBase):
    """The local block of a row-sharded [N, D] embedding table: `item_embedding.weight` / `user_embedding.weight` of BPR
    (bpr.py:77-81,103-125), the table the knn aggregate averages over (knn_embedder.py:117-147), a bucket table too big to
    replicate.  table_local: float32[hi-lo, D], rows [lo, hi) of the global table.

        gather(ids)            == ops.gather_rows(ids, table)            rows are copies: identical bits, NaN off the table
        gather_mean(idx, g)    == ops.gather_mean(idx, table, g)         same sums in the same order on the gathered rows
        topk(U, k, n_skip_low) == ops.score_topk(U, table, k, n_skip_low)  per-shard fused top-k + merge, users replicated

    The D-wide row is the answer, so D * 4 bytes per REMOTE lookup cross the links on the way back (8 on the way out);
    lookups this rank owns are served from its block by the same splice kernel that places the remote answers and never
    enter the exchange."""

    def __init__(self, table_local, n_rows_global, group=None, prims=None, cap_factor=None, max_batch=None,
                 uniform_batches=False):
        super().__init__(table_local, n_rows_global, group, prims, cap_factor, max_batch, uniform_batches)
        self.table_local = table_local

    def gather(self, ids):
        ids = ids.reshape(-1)
        local = (ids >= self.lo) & (ids < self.hi)
        p = self.begin(torch.where(local, torch.full_like(ids, -1), ids))  # local lookups are not sent anywhere
        self._wait(p.w_ids)
        rows = self.prims.gather_rows(p.recv.view(-1), self.table_local)   # NaN rows for the -1 padding
        self._reply(p, rows.view(self.world, -1), False)
        self._wait(p.w_back)
        # own rows by key = id - lo (anything else keyed past the block), remote ones by slot; one kernel for both
        key = torch.where(local, ids - self.lo, torch.full_like(ids, self.hi - self.lo))
        return self.prims.splice(key, p.slot.to(torch.int64), self.table_local, p.back.view(-1, self.table_local.shape[1]))

    def gather_mean(self, idx, g=2):
        """vstack(chunk.mean(0) for chunk in W[idx.ravel()].split(g)) over the sharded W: the rows are gathered through the
        exchange and averaged on the requester in position order -- what the owner could pre-add (two neighbours of one
        lookup in one shard, 1 / world of the pairs) is not worth a second record format."""
        return self.prims.gather_mean(self.gather(idx.reshape(-1)), g)

    def topk(self, U, k, n_skip_low=0):
        """Per-row top-k of U @ table.T over the sharded catalogue: U replicated on every rank (all_gather it first if the
        user batch itself is split), each rank runs the fused top-k on its block -- the [B, N] scores exist nowhere --
        and the [B, k] candidates with GLOBAL row numbers are merged (larger score first, ties to the lower row)."""
        n_local = self.hi - self.lo
        skip = min(max(int(n_skip_low) - self.lo, 0), n_local)
        kk = min(k, n_local)
        vals = torch.full((U.shape[0], k), float("-inf"), dtype=torch.float32, device=U.device)
        idx = torch.full((U.shape[0], k), -1, dtype=torch.int64, device=U.device)
        if kk > 0:
            v, i = self.prims.score_topk(U, self.table_local, kk, skip)
            vals[:, :kk] = v
            idx[:, :kk] = torch.where(i >= 0, i + self.lo, i)
        return merge_topk(vals, idx, k, self.group)


def merge_topk(vals, idx, k, group=None):
    """Full-catalogue scoring over an item-sharded table: every rank holds its local top-k
    (vals, idx with GLOBAL item numbers) for the same replicated users; all_gather the [B,k]
    candidates and keep the k best (larger value first, ties to the lower item id)."""
    world = dist.get_world_size(group)
    if vals.is_cuda and _backend(group) != "nccl":  # gloo rehearsal on a device: through the host
        hv = [torch.empty(vals.shape, dtype=vals.dtype) for _ in range(world)]
        hi_ = [torch.empty(idx.shape, dtype=idx.dtype) for _ in range(world)]
        dist.all_gather(hv, vals.cpu().contiguous(), group=group)
        dist.all_gather(hi_, idx.cpu().contiguous(), group=group)
        gv, gi = [t.to(vals.device) for t in hv], [t.to(idx.device) for t in hi_]
    else:
        gv = [torch.empty_like(vals) for _ in range(world)]
        gi = [torch.empty_like(idx) for _ in range(world)]
        dist.all_gather(gv, vals.contiguous(), group=group)
        dist.all_gather(gi, idx.contiguous(), group=group)
    v = torch.cat(gv, dim=1)
    i = torch.cat(gi, dim=1)
    # (-inf, -1) fillers of shards with fewer than k admissible rows go behind every real candidate, -inf ones included
    i = torch.where(i < 0, torch.full_like(i, 1 << 62), i)
    # sort by (value desc, index asc): stable sort on index first, then stable sort on value
    o1 = torch.argsort(i, dim=1, stable=True)
    v, i = torch.gather(v, 1, o1), torch.gather(i, 1, o1)
    o2 = torch.argsort(v, dim=1, descending=True, stable=True)
    v, i = torch.gather(v, 1, o2)[:, :k], torch.gather(i, 1, o2)[:, :k]
    return v, torch.where(i == (1 << 62), torch.full_like(i, -1), i)
