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
    def codes(local_ids, feat_local, planes):
        from . import ops
        if feat_local.shape[0] == 0:  # more ranks than rows: this rank owns nothing and is never asked
            return torch.full((local_ids.numel(), planes.shape[0]), 255, dtype=torch.uint8, device=local_ids.device)
        return ops.lsh_bits(local_ids, feat_local, planes)

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
        return ops.gather_rows(idx, table)


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
    __slots__ = ("B", "cap", "slot", "counts", "recv", "back", "w_ids", "w_back", "width", "dtype")


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

    def begin(self, ids, async_op=False):
        """Bucket `ids` (global rows, int64[B]) by owner and send every owner its segment."""
        p = _Pending()
        p.B = ids.numel()
        p.cap = self._agreed_capacity(p.B)
        send, p.slot, p.counts = self.prims.bucket(ids, self.n_rows, self.per, self.world, p.cap, self.overflow)
        p.recv = torch.empty_like(send)
        p.w_ids = _all_to_all(p.recv, send, self.group, async_op)
        p.w_back = None
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
        self._reply(p, self.prims.codes(p.recv.view(-1), self.feat_local, planes), async_op)

    def finish(self, p, buckets, other=None, want_emb=True, score_out=None):
        self._wait(p.w_back)
        return self.prims.codes_embed(p.back, p.slot, buckets, other, want_emb, score_out)

    def embed(self, ids, planes, buckets):
        p = self.begin(ids)
        self.owner(p, planes)
        return self.finish(p, buckets)[1]

    def embed_score(self, ids, planes, buckets, other, score_out=None):
        p = self.begin(ids)
        self.owner(p, planes)
        return self.finish(p, buckets, other, want_emb=False, score_out=score_out)[0]


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

    def __init__(self, table, planes, buckets, streams=None):
        self.table, self.planes, self.buckets = table, planes, buckets
        dev = table.feat_local.device
        if streams is None:
            streams = dev.type == "cuda" and _backend(table.group) == "nccl"
        self.streams = [torch.cuda.Stream(dev) for _ in range(3)] if streams else None

    def run(self, ids, others, scores):
        """ids[t] int64[M], others[t] f32[M,D], scores[t] f32[M] (written).  No host synchronisation."""
        t_, n = self.table, len(ids)
        if n == 0:
            return
        if self.streams is None:
            pend = [t_.begin(ids[t], async_op=True) for t in range(min(2, n))]
            prev = None
            for t in range(n):
                p = pend.pop(0)
                t_.owner(p, self.planes, async_op=True)
                if prev is not None:
                    t_.finish(prev[0], self.buckets, others[prev[1]], want_emb=False, score_out=scores[prev[1]])
                if t + 2 < n:
                    pend.append(t_.begin(ids[t + 2], async_op=True))
                prev = (p, t)
            t_.finish(prev[0], self.buckets, others[prev[1]], want_emb=False, score_out=scores[prev[1]])
            return
        s_bucket, s_owner, s_req = self.streams
        caller = torch.cuda.current_stream(t_.feat_local.device)
        for s in self.streams:  # the inputs were produced on the caller's stream
            s.wait_stream(caller)

        def begin(t):
            with torch.cuda.stream(s_bucket):
                return t_.begin(ids[t], async_op=True)

        def owner(p):
            with torch.cuda.stream(s_owner):
                p.recv.record_stream(s_owner)  # allocated on the bucket stream, read here
                t_.owner(p, self.planes, async_op=True)

        def finish(p, t):
            with torch.cuda.stream(s_req):
                p.slot.record_stream(s_req)  # bucket stream -> here
                t_._wait(p.w_back)
                p.w_back = None
                p.back.record_stream(s_req)  # owner stream -> here
                t_.finish(p, self.buckets, others[t], want_emb=False, score_out=scores[t])

        pend = [begin(t) for t in range(min(2, n))]
        prev = None
        for t in range(n):
            p = pend.pop(0)
            owner(p)
            if prev is not None:
                finish(*prev)
            if t + 2 < n:
                pend.append(begin(t + 2))
            prev = (p, t)
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


def merge_topk(vals, idx, k, group=None):
    """Full-catalogue scoring over an item-sharded table: every rank holds its local top-k
    (vals, idx with GLOBAL item numbers) for the same replicated users; all_gather the [B,k]
    candidates and keep the k best (larger value first, ties to the lower item id)."""
    world = dist.get_world_size(group)
    gv = [torch.empty_like(vals) for _ in range(world)]
    gi = [torch.empty_like(idx) for _ in range(world)]
    dist.all_gather(gv, vals.contiguous(), group=group)
    dist.all_gather(gi, idx.contiguous(), group=group)
    v = torch.cat(gv, dim=1)
    i = torch.cat(gi, dim=1)
    # sort by (value desc, index asc): stable sort on index first, then stable sort on value
    o1 = torch.argsort(i, dim=1, stable=True)
    v, i = torch.gather(v, 1, o1), torch.gather(i, 1, o1)
    o2 = torch.argsort(v, dim=1, descending=True, stable=True)
    return torch.gather(v, 1, o2)[:, :k], torch.gather(i, 1, o2)[:, :k]
