"""Row-sharded tables across the GPUs of one node: new functionality, the
reference has no table sharding (its only multi-GPU mode is DDP replicas, R/trainer/trainer.py:68-72).

Placement rule: tables that fit one GPU's 288 GB HBM are REPLICATED and
ranks process independent batches with no data-path collective (bench.py --gpus N).  Only tables
too large for one GPU are row-sharded; then a lookup batch takes one exchange each way:

    ids --bucket by owner--> all_to_all(counts) -> all_to_all(ids)          8 B per lookup out
    owner runs the fused kernel on its local rows (replicated planes/buckets)
    all_to_all(rows) back, un-permute                                       4*D B per lookup back

The F-wide feature row never crosses xGMI, only the D-wide result.  xGMI is point-to-point, so an
all-to-all uses all 7 links of a GPU at once; no ring is involved.  One process per GPU;
`torch.distributed` backend "nccl" is RCCL on ROCm, "gloo" is used by the CPU tests of the
exchange logic (tests/test_sharded_gloo.py) with the local compute injected.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_rows: int, world: int, rank: int):
    """Contiguous block partition: rank r owns rows [r*per, min(n, (r+1)*per)), per = ceil(n/world)."""
    per = (n_rows + world - 1) // world
    lo = min(n_rows, rank * per)
    hi = min(n_rows, lo + per)
    return lo, hi, per


def exchange_lookup(ids, rows_per_rank, width, local_fn, group=None):
    """Owner-computes lookup of `ids` (global row numbers, int64[B]) against a row-sharded table.

    local_fn(local_ids) -> float32[len(local_ids), width] is evaluated on the owning rank with
    local_ids = global id - rank*rows_per_rank.  Returns float32[B, width] in the order of `ids`.
    Ids outside every shard are sent to the last rank, whose kernel marks them NaN.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = ids.device
    owner = torch.clamp(torch.div(ids, max(1, rows_per_rank), rounding_mode="floor"), 0, world - 1)
    order = torch.argsort(owner, stable=True)
    send_ids = ids[order].contiguous()
    send_counts = torch.bincount(owner, minlength=world)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    sc, rc = send_counts.tolist(), recv_counts.tolist()
    recv_ids = torch.empty((sum(rc),), dtype=torch.int64, device=dev)
    dist.all_to_all_single(recv_ids, send_ids, output_split_sizes=rc, input_split_sizes=sc, group=group)
    rows = local_fn(recv_ids - rank * rows_per_rank)
    if rows.shape != (recv_ids.numel(), width):
        raise RuntimeError(f"local_fn returned {tuple(rows.shape)}, expected {(recv_ids.numel(), width)}")
    back = torch.empty((ids.numel(), width), dtype=rows.dtype, device=dev)
    dist.all_to_all_single(back, rows.contiguous(), output_split_sizes=sc, input_split_sizes=rc, group=group)
    out = torch.empty_like(back)
    out[order] = back
    return out


class ShardedLSHTable:
    """The local block of a row-sharded lsh/slsh feature matrix plus the replicated small state.

    feat_local: float32[hi-lo, F] rows [lo, hi) of the global matrix, resident on this rank's GPU.
    embed(ids, planes, buckets) returns (bits @ buckets)/popcount for global ids, bit-identical to
    the single-GPU kernel on the unsharded matrix (each row is computed by exactly one owner with
    the same kernel)."""

    def __init__(self, feat_local, n_rows_global, group=None, local_embed=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_rows = n_rows_global
        self.lo, self.hi, self.per = shard_bounds(n_rows_global, self.world, self.rank)
        if feat_local.shape[0] != self.hi - self.lo:
            raise ValueError(f"rank {self.rank} must hold rows [{self.lo},{self.hi}), got {feat_local.shape[0]}")
        self.feat_local = feat_local
        if local_embed is None:
            from . import ops
            local_embed = ops.lsh_embed
        self._local_embed = local_embed

    def embed(self, ids, planes, buckets):
        D = buckets.shape[1]

        def local_fn(local_ids):
            if local_ids.numel() == 0:
                return torch.empty((0, D), dtype=torch.float32, device=local_ids.device)
            return self._local_embed(local_ids, self.feat_local, planes, buckets)

        return exchange_lookup(ids, self.per, D, local_fn, self.group)


class ShardedSLSHTable:
    """slsh over row-sharded tables (BASELINE config 4: 1e8 x 64 features + a 1e8 x 128 bucket/item table, 8 ways).

    Two owner-computes exchanges per batch, neither moving an F-wide row:
        ids  -> owner of the FEATURE row: popcount bucket id (single_lsh_embedder.py:82-87)   8 B out, 8 B back
        idx  -> owner of the BUCKET row:  row gather (single_lsh_embedder.py:100,108)         8 B out, 4*D B back
    feat_local / buckets_local are this rank's contiguous row blocks (shard_bounds).  With the reference's
    popcount ids every lookup lands in rows [bits_req, 2*bits_req] of the bucket table, i.e. on rank 0 -- the
    second exchange is then an all-to-one; that is the reference's arithmetic, not a property of the exchange."""

    def __init__(self, feat_local, n_feat_rows, buckets_local, n_buckets, group=None, local_index=None, local_gather=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_feat_rows, self.n_buckets = n_feat_rows, n_buckets
        flo, fhi, self.per_feat = shard_bounds(n_feat_rows, self.world, self.rank)
        blo, bhi, self.per_bucket = shard_bounds(n_buckets, self.world, self.rank)
        if feat_local.shape[0] != fhi - flo or buckets_local.shape[0] != bhi - blo:
            raise ValueError(f"rank {self.rank} must hold feature rows [{flo},{fhi}) and bucket rows [{blo},{bhi})")
        self.feat_local, self.buckets_local = feat_local, buckets_local
        if local_index is None or local_gather is None:
            from . import ops
            local_index = local_index or (lambda ids, feat, planes, nb: ops.slsh_index(ids, feat, planes, nb))
            local_gather = local_gather or ops.gather_rows
        self._local_index, self._local_gather = local_index, local_gather

    def embed(self, ids, planes):
        D = self.buckets_local.shape[1]

        def index_fn(local_ids):
            if local_ids.numel() == 0:
                return torch.empty((0, 1), dtype=torch.int64, device=local_ids.device)
            return self._local_index(local_ids, self.feat_local, planes, self.n_buckets).view(-1, 1)

        def gather_fn(local_idx):
            if local_idx.numel() == 0:
                return torch.empty((0, D), dtype=torch.float32, device=local_idx.device)
            return self._local_gather(local_idx, self.buckets_local)

        idx = exchange_lookup(ids, self.per_feat, 1, index_fn, self.group).view(-1)  # -1 for ids outside the table
        # invalid lookups (-1) go to rank 0 as local row -1, which the gather kernel answers with a NaN row
        return exchange_lookup(idx, self.per_bucket, D, gather_fn, self.group), idx


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
