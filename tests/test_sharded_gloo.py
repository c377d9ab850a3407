"""CPU, world_size 2 (gloo): the N > 1 exchange logic of mi_oov.sharded with the per-rank compute
injected (the oracle), checked against the unsharded oracle.  Covers ragged splits, ids that all
live on one rank, empty local batches, tight capacities that overflow, the software-pipelined
sequence of steps, the replicated slsh window and the top-k merge."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class OraclePrims:
    """mi_oov.sharded's local compute with the CPU oracle in the place of the HIP kernels (tests only)."""

    @staticmethod
    def bucket(ids, n_rows, per, world, cap, overflow):
        from oracle import oov_oracle as oracle
        send, slot, counts = oracle.bucket_by_owner(ids.numpy(), n_rows, per, world, cap)
        overflow[0] = max(int(overflow[0]), int(counts.max()) - cap, 0)
        return torch.from_numpy(send), torch.from_numpy(slot), torch.from_numpy(counts)

    @staticmethod
    def bucket_local(ids, n_rows, per, world, cap, overflow, my_rank):
        """mi_oov_bucket_by_owner_fused with my_rank, restated on the oracle's bucketing: the lookups of my_rank leave their
        send segment for local_rows and take slots world * cap + position."""
        from oracle import oov_oracle as oracle
        send, slot, counts = oracle.bucket_by_owner(ids.numpy(), n_rows, per, world, cap)
        overflow[0] = max(int(overflow[0]), int(counts.max()) - cap, 0)
        local_rows = send[my_rank].copy()
        send[my_rank] = -1
        mine = (slot >= my_rank * cap) & (slot < (my_rank + 1) * cap)
        slot = slot.copy()
        slot[mine] += (world - my_rank) * cap
        return torch.from_numpy(send), torch.from_numpy(slot), torch.from_numpy(counts), torch.from_numpy(local_rows)

    @staticmethod
    def codes(local_ids, feat_local, planes, out=None):
        from oracle import oov_oracle as oracle
        H = planes.shape[0]
        if feat_local.shape[0] == 0:
            return torch.full((local_ids.numel(), H), 255, dtype=torch.uint8)
        _, bits = oracle.lsh_embed(local_ids.numpy(), feat_local.numpy(), planes.numpy(), np.zeros((H, 1), np.float32),
                                   want_bits=True)
        return torch.from_numpy(bits)

    @staticmethod
    def codes_embed(codes, slot, buckets, other, want_emb, score_out=None):
        from oracle import oov_oracle as oracle
        score, emb = oracle.lsh_codes_embed(codes.numpy().reshape(-1, codes.shape[-1]), slot.numpy(), buckets.numpy(),
                                            None if other is None else other.numpy())
        if score is not None and score_out is not None:
            score_out.copy_(torch.from_numpy(score))
            score = score_out
        elif score is not None:
            score = torch.from_numpy(score)
        return score, torch.from_numpy(emb)

    @staticmethod
    def slsh_index(local_ids, feat_local, planes, n_buckets):
        from oracle import oov_oracle as oracle
        if feat_local.shape[0] == 0:
            return torch.full((local_ids.numel(),), -1, dtype=torch.int64)
        _, idx = oracle.slsh_embed(local_ids.numpy(), feat_local.numpy(), planes.numpy(), np.zeros((n_buckets, 1), np.float32))
        return torch.from_numpy(idx)

    @staticmethod
    def gather_rows(idx, table):
        li, rows = idx.numpy(), table.numpy()
        out = np.full((len(li), rows.shape[1]), np.nan, np.float32)
        okm = (li >= 0) & (li < rows.shape[0])
        out[okm] = rows[li[okm]]
        return torch.from_numpy(out)


    @staticmethod
    def splice(key, slot, table_local, back):
        k, sl, tl, bk = key.numpy(), slot.numpy(), table_local.numpy(), back.numpy()
        out = np.full((len(k), bk.shape[1]), np.nan, np.float32)
        own = (k >= 0) & (k < tl.shape[0])
        out[own] = tl[k[own]]
        rem = ~own & (k >= tl.shape[0]) & (sl >= 0) & (sl < bk.shape[0])
        out[rem] = bk[sl[rem]]
        return torch.from_numpy(out)

    @staticmethod
    def gather_mean(rows, g):
        from oracle import oov_oracle as oracle
        r = rows.numpy()
        return torch.from_numpy(oracle.gather_mean(np.arange(r.shape[0], dtype=np.int64), r, g))

    @staticmethod
    def score_topk(U, E_local, k, n_skip_low):
        from oracle import oov_oracle as oracle
        v, i = oracle.score_topk(U.numpy(), E_local.numpy(), k, n_skip_low)
        return torch.from_numpy(v), torch.from_numpy(i)

    @staticmethod
    def lsh_embed_score(ids_local, feat_local, planes, buckets, other, score_out=None):
        from oracle import oov_oracle as oracle
        sc, _ = oracle.lsh_embed_score(ids_local.numpy(), feat_local.numpy(), planes.numpy(), buckets.numpy(), other.numpy())
        return torch.from_numpy(sc)


def _check_embedding_table(sharded, oracle, rank, world, chk, seed=0):
    """ShardedEmbeddingTable (row e'): D-wide gathers, the knn aggregate and the sharded full-catalogue top-k against the
    unsharded oracle, on every rank with its own ids; plus the local fast path of the lsh exchange."""
    T = torch.from_numpy
    rng = np.random.default_rng(100 + seed)  # same tables on every rank
    N, D = 1003, 20
    W = rng.standard_normal((N, D), dtype=np.float32)
    W[5] = -0.0  # bits must survive the wire
    lo, hi, _ = sharded.shard_bounds(N, world, rank)
    tab = sharded.ShardedEmbeddingTable(T(W[lo:hi].copy()), N, prims=OraclePrims)
    r = np.random.default_rng(200 + rank)
    cases = {
        "ragged": r.integers(0, N, size=150 + 61 * rank),
        "all_local": r.integers(lo, max(lo + 1, hi), size=40) if hi > lo else np.zeros((0,), np.int64),
        "all_remote": r.integers(0, N, size=300),
        "edges": np.array([0, lo, max(lo, hi - 1), hi % N, N - 1, N, -1, 5, 5]),
        "empty": np.zeros((0,), np.int64) if rank == 0 else np.arange(0, N, 13),
    }
    cases["all_remote"] = cases["all_remote"][(cases["all_remote"] < lo) | (cases["all_remote"] >= hi)][:120]
    n_remote = len(cases["all_remote"])
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([n_remote]))
    for name, ids in cases.items():
        ids = ids.astype(np.int64)
        chk(f"gather {name}", _same(tab.gather(T(ids)).numpy().view(np.uint32), oracle.gather_rows(ids, W).view(np.uint32)))
        for g in (2, 3):
            m = len(ids) // g * g if name != "ragged" else len(ids)  # ragged keeps a short last group
            chk(f"gather_mean {name} g={g}", _same(tab.gather_mean(T(ids[:m]), g).numpy(), oracle.gather_mean(ids[:m], W, g)))
    chk("emb overflow", int(tab.overflow) == 0)
    # sharded full-catalogue top-k: users replicated, ties across shards, padding column skipped, NaN score first
    U = rng.standard_normal((11, D), dtype=np.float32)
    E = W.copy()
    E[N - 2] = E[3]  # a tie across the first and the last shard -> the lower global row wins
    E[7, 0] = np.nan
    tabE = sharded.ShardedEmbeddingTable(T(E[lo:hi].copy()), N, prims=OraclePrims)
    for k, skip in ((5, 0), (5, 1), (20, 1), (3, 400)):
        gv, gi = tabE.topk(T(U), k, skip)
        wv, wi = oracle.score_topk(U, E, k, skip)
        chk(f"topk k={k} skip={skip}", np.array_equal(gi.numpy(), wi) and _same(gv.numpy(), wv))
    # a catalogue smaller than k * world: shards with fewer than k rows pad with (-inf, -1)
    small = sharded.ShardedEmbeddingTable(T(E[:7][slice(*sharded.shard_bounds(7, world, rank)[:2])].copy()), 7, prims=OraclePrims)
    gv, gi = small.topk(T(U), 5, 1)
    wv, wi = oracle.score_topk(U, E[:7], 5, 1)
    chk("topk tiny catalogue", np.array_equal(gi.numpy(), wi) and _same(gv.numpy(), wv))
    # lsh: lookups this rank owns skip the exchange (fused kernel on the local block), same scores
    F, H = 12, 5
    feat = rng.standard_normal((N, F), dtype=np.float32)
    planes = rng.standard_normal((H, F), dtype=np.float32)
    buckets = rng.standard_normal((H, D), dtype=np.float32)
    lt = sharded.ShardedLSHTable(T(feat[lo:hi].copy()), N, prims=OraclePrims)
    for name, ids in cases.items():
        ids = ids.astype(np.int64)
        other = np.random.default_rng(300 + rank).standard_normal((len(ids), D), dtype=np.float32)
        want = oracle.lsh_embed_score(ids, feat, planes, buckets, other)[0]
        got = lt.embed_score(T(ids), T(planes), T(buckets), T(other), local_fast=True).numpy()
        chk(f"lsh local_fast {name}", _same(got, want))
        buf = torch.empty(len(ids))
        lt.embed_score(T(ids), T(planes), T(buckets), T(other), score_out=buf, local_fast=True)
        chk(f"lsh local_fast score_out {name}", _same(buf.numpy(), want))
    # the pipelined steps with the local share on the fused kernel
    r3 = np.random.default_rng(400 + rank)
    ids_l = [r3.integers(-2, N + 2, size=90).astype(np.int64) for _ in range(4)]
    oth_l = [r3.standard_normal((90, D), dtype=np.float32) for _ in range(4)]
    sc_l = [torch.empty(90) for _ in range(4)]
    sharded.LshPipeline(lt, T(planes), T(buckets), local_fast=True).run([T(i) for i in ids_l], [T(o) for o in oth_l], sc_l)
    for t in range(4):
        chk(f"pipeline local_fast step {t}", _same(sc_l[t].numpy(), oracle.lsh_embed_score(ids_l[t], feat, planes, buckets, oth_l[t])[0]))


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(np.nan_to_num(a, nan=7.0), np.nan_to_num(b, nan=7.0))


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mi_oov  # noqa: F401
        from mi_oov import sharded
        from oracle import oov_oracle as oracle
        rng = np.random.default_rng(0)  # same data on every rank
        N, F, H, D = 1001, 24, 6, 12
        feat = rng.standard_normal((N, F), dtype=np.float32)
        planes = rng.standard_normal((H, F), dtype=np.float32)
        buckets = rng.standard_normal((H, D), dtype=np.float32)
        lo, hi, per = sharded.shard_bounds(N, world, rank)
        assert per == 501 and (hi - lo) == (501 if rank == 0 else 500)
        T = torch.from_numpy
        table = sharded.ShardedLSHTable(T(feat[lo:hi]), N, prims=OraclePrims)
        cases = {
            "ragged": np.random.default_rng(10 + rank).integers(0, N, size=300 + 77 * rank),
            "all_on_rank0": np.random.default_rng(20 + rank).integers(0, 400, size=64),
            "all_on_rank1": np.random.default_rng(30 + rank).integers(600, N, size=50),
            "empty_here": np.zeros((0,), np.int64) if rank == 0 else np.arange(900, 1001),
            "edges": np.array([0, 500, 501, 1000, 500, 0, N + 3, -2]),  # the last two: outside the table -> NaN rows
        }
        fails = []

        def chk(line, cond):
            if not cond:
                fails.append(line)

        for name, ids in cases.items():
            ids = ids.astype(np.int64)
            got = table.embed(T(ids), T(planes), T(buckets)).numpy()
            chk(102, _same(got, oracle.lsh_embed(ids, feat, planes, buckets)))
            other = np.random.default_rng(40 + rank).standard_normal((len(ids), D), dtype=np.float32)
            sc = table.embed_score(T(ids), T(planes), T(buckets), T(other)).numpy()
            chk(105, _same(sc, oracle.lsh_embed_score(ids, feat, planes, buckets, other)[0]))
        chk(106, int(table.overflow) == 0)
        table.check_overflow()

        # a tight capacity with every id on rank 0: the segment overflows, the dropped lookups are NaN and flagged
        tight = sharded.ShardedLSHTable(T(feat[lo:hi]), N, prims=OraclePrims, cap_factor=1.0, max_batch=1000)
        chk(111, tight.capacity(64) == 64 and tight.capacity(1000) == 704)
        big = np.random.default_rng(50 + rank).integers(0, 400, size=1000).astype(np.int64)  # 1000 ids -> segment of 704 (500 + 8 sigma, rounded up to 64)
        got = tight.embed(T(big), T(planes), T(buckets)).numpy()
        want = oracle.lsh_embed(big, feat, planes, buckets)
        kept = ~np.isnan(got).all(1) | np.isnan(want).all(1)
        chk(116, int(tight.overflow) == 1000 - 704 and 250 < int((~kept).sum()) <= 1000 - 704 and _same(got[kept], want[kept]))
        try:
            tight.check_overflow()
            fails.append("check_overflow did not raise")
        except RuntimeError:
            pass

        # uniform_batches: the capacity follows the batch of the call (every rank passes the same size)
        uni = sharded.ShardedLSHTable(T(feat[lo:hi]), N, prims=OraclePrims, cap_factor=1.0, uniform_batches=True)
        for nb in (64, 1000):
            idsu = np.random.default_rng(70 + rank).integers(0, N, size=nb).astype(np.int64)
            p = uni.begin(T(idsu))
            chk(130, p.cap == uni.capacity(nb))
            uni.owner(p, T(planes))
            chk(131, _same(uni.finish(p, T(buckets))[1].numpy(), oracle.lsh_embed(idsu, feat, planes, buckets)))
        chk(132, int(uni.overflow) == 0)

        # software-pipelined steps (three in flight) == the steps one by one
        n_steps, M = 5, 200
        r2 = np.random.default_rng(60 + rank)
        ids_l = [r2.integers(0, N, size=M).astype(np.int64) for _ in range(n_steps)]
        oth_l = [r2.standard_normal((M, D), dtype=np.float32) for _ in range(n_steps)]
        sc_l = [torch.empty(M) for _ in range(n_steps)]
        for n_run in (1, 2, n_steps):
            sharded.LshPipeline(table, T(planes), T(buckets)).run([T(i) for i in ids_l[:n_run]], [T(o) for o in oth_l[:n_run]],
                                                                  sc_l[:n_run])
            for t in range(n_run):
                chk(133, _same(sc_l[t].numpy(), oracle.lsh_embed_score(ids_l[t], feat, planes, buckets, oth_l[t])[0]))

        # slsh over a row-sharded feature table; the bucket table row-sharded too, its reachable window replicated
        for NB, n_pl in ((777, 10), (12, 10), (21, 10)):  # window inside the table / wrapping ids: whole table replicated
            D2 = 20
            planes_s = rng.standard_normal((n_pl, F), dtype=np.float32)
            bigt = rng.standard_normal((NB, D2), dtype=np.float32)
            bigt[n_pl] = -0.0  # a row of negative zeros must survive the window gather bit for bit
            blo, bhi, _ = sharded.shard_bounds(NB, world, rank)
            window, win_lo = sharded.ShardedSLSHTable.gather_window(T(bigt[blo:bhi]), NB, n_pl)
            wlo, whi = sharded.ShardedSLSHTable.window_bounds(n_pl, NB)
            chk(144, win_lo == wlo and np.array_equal(window.numpy().view(np.uint32), bigt[wlo:whi].view(np.uint32)))
            st = sharded.ShardedSLSHTable(T(feat[lo:hi]), N, window, win_lo, NB, prims=OraclePrims)
            for name, ids in cases.items():
                ids = ids.astype(np.int64)
                got, gidx = st.embed(T(ids), T(planes_s))
                want, widx = oracle.slsh_embed(ids, feat, planes_s, bigt)
                chk(150, np.array_equal(gidx.numpy(), widx) and _same(got.numpy(), want))
        # top-k merge over an item-sharded catalogue
        U = rng.standard_normal((9, D), dtype=np.float32)
        E = rng.standard_normal((N, D), dtype=np.float32)
        E[700] = E[3]  # a tie across shards -> the lower global index wins
        k = 5
        lv, li = oracle.score_topk(U, E[lo:hi], k)
        mv, mi = sharded.merge_topk(T(lv), T(li + lo), k)
        wv, wi = oracle.score_topk(U, E, k)
        chk(159, np.array_equal(mi.numpy(), wi) and np.array_equal(mv.numpy(), wv))
        _check_embedding_table(sharded, oracle, rank, world, chk)
        ret[rank] = fails
    finally:
        dist.destroy_process_group()


def _worker_odd(rank, world, port, ret):
    """Three ranks: shards of unequal size (334 / 334 / 333 rows), a rank that is asked for nothing, ids on every
    boundary; lsh, the pipelined steps and slsh against the unsharded oracle."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mi_oov  # noqa: F401
        from mi_oov import sharded
        from oracle import oov_oracle as oracle
        rng = np.random.default_rng(1)
        N, F, H, D = 1001, 16, 8, 8
        feat = rng.standard_normal((N, F), dtype=np.float32)
        planes = rng.standard_normal((H, F), dtype=np.float32)
        buckets = rng.standard_normal((H, D), dtype=np.float32)
        lo, hi, per = sharded.shard_bounds(N, world, rank)
        T = torch.from_numpy
        fails = []
        if (per, hi - lo) != (334, 334 if rank < 2 else 333):
            fails.append(f"shard_bounds {per} {lo} {hi}")
        table = sharded.ShardedLSHTable(T(feat[lo:hi]), N, prims=OraclePrims)
        cases = {
            "ragged": np.random.default_rng(10 + rank).integers(0, N, size=100 + 211 * rank),
            "nothing_for_rank1": np.concatenate([np.arange(0, 334, 7), np.arange(668, N, 5)]),
            "boundaries": np.array([0, 333, 334, 667, 668, 1000, 1001, -1, 333, 668]),
            "empty_on_rank2": np.zeros((0,), np.int64) if rank == 2 else np.arange(300, 700),
        }
        for name, ids in cases.items():
            ids = ids.astype(np.int64)
            other = np.random.default_rng(40 + rank).standard_normal((len(ids), D), dtype=np.float32)
            if not _same(table.embed(T(ids), T(planes), T(buckets)).numpy(), oracle.lsh_embed(ids, feat, planes, buckets)):
                fails.append(f"embed {name}")
            if not _same(table.embed_score(T(ids), T(planes), T(buckets), T(other)).numpy(),
                         oracle.lsh_embed_score(ids, feat, planes, buckets, other)[0]):
                fails.append(f"embed_score {name}")
        # capacities agreed through max_batch; five pipelined steps
        fixed = sharded.ShardedLSHTable(T(feat[lo:hi]), N, prims=OraclePrims, cap_factor=1.5, max_batch=400)
        r2 = np.random.default_rng(60 + rank)
        ids_l = [r2.integers(0, N, size=150 + 50 * rank).astype(np.int64) for _ in range(5)]
        oth_l = [r2.standard_normal((len(i), D), dtype=np.float32) for i in ids_l]
        sc_l = [torch.empty(len(i)) for i in ids_l]
        sharded.LshPipeline(fixed, T(planes), T(buckets)).run([T(i) for i in ids_l], [T(o) for o in oth_l], sc_l)
        for t in range(5):
            if not _same(sc_l[t].numpy(), oracle.lsh_embed_score(ids_l[t], feat, planes, buckets, oth_l[t])[0]):
                fails.append(f"pipeline step {t}")
        if int(fixed.overflow) != 0:
            fails.append("overflow")
        # slsh, bucket table of 100 rows sharded 34 / 34 / 32
        NB, n_pl = 100, 7
        planes_s = rng.standard_normal((n_pl, F), dtype=np.float32)
        bigt = rng.standard_normal((NB, D), dtype=np.float32)
        blo, bhi, _ = sharded.shard_bounds(NB, world, rank)
        window, win_lo = sharded.ShardedSLSHTable.gather_window(T(bigt[blo:bhi]), NB, n_pl)
        st = sharded.ShardedSLSHTable(T(feat[lo:hi]), N, window, win_lo, NB, prims=OraclePrims)
        for name, ids in cases.items():
            ids = ids.astype(np.int64)
            got, gidx = st.embed(T(ids), T(planes_s))
            want, widx = oracle.slsh_embed(ids, feat, planes_s, bigt)
            if not (np.array_equal(gidx.numpy(), widx) and _same(got.numpy(), want)):
                fails.append(f"slsh {name}")
        _check_embedding_table(sharded, oracle, rank, world, lambda line, cond: cond or fails.append(line), seed=1)
        ret[rank] = fails
    finally:
        dist.destroy_process_group()


def test_exchange_lookup_world3_uneven_shards():
    from oracle import oov_oracle
    oov_oracle.build()
    world = 3
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_odd, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: [], 1: [], 2: []}


def test_exchange_lookup_world2():
    from oracle import oov_oracle
    oov_oracle.build()
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: [], 1: []}


def test_shard_bounds():
    from mi_oov import sharded
    assert sharded.shard_bounds(10, 4, 0) == (0, 3, 3)
    assert sharded.shard_bounds(10, 4, 3) == (9, 10, 3)
    assert sharded.shard_bounds(2, 4, 3) == (2, 2, 1)  # more ranks than rows: empty tail shards
    covered = []
    for r in range(8):
        lo, hi, _ = sharded.shard_bounds(100_000_001, 8, r)
        covered.append((lo, hi))
    assert covered[0][0] == 0 and covered[-1][1] == 100_000_001
    assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    assert sharded.ShardedSLSHTable.window_bounds(27, 100_000_000) == (27, 55)
    assert sharded.ShardedSLSHTable.window_bounds(3, 8) == (3, 7) and sharded.ShardedSLSHTable.window_bounds(4, 8) == (0, 8)


def test_oracle_bucket_and_codes_embed():
    """The oracle's two exchange ends against plain numpy: stable partition, and codes -> rows == lsh_embed."""
    from oracle import oov_oracle as oracle
    rng = np.random.default_rng(3)
    N, world, B = 1000, 4, 700
    per = -(-N // world)
    ids = rng.integers(-5, N + 5, size=B).astype(np.int64)
    send, slot, counts = oracle.bucket_by_owner(ids, N, per, world, B)
    valid = (ids >= 0) & (ids < N)
    owner = np.minimum(ids // per, world - 1)
    assert np.array_equal(counts, np.bincount(owner[valid], minlength=world))
    assert (slot[~valid] == -2).all() and np.array_equal(slot[valid] // B, owner[valid])
    assert np.array_equal(send.reshape(-1)[slot[valid]], ids[valid] - owner[valid] * per)
    assert (send.reshape(-1) == -1).sum() == world * B - valid.sum()
    for w in range(world):  # stable: order of appearance
        assert np.array_equal(send[w, :counts[w]], (ids - w * per)[valid & (owner == w)])
    F, H, D = 20, 7, 36
    feat = rng.standard_normal((N, F), dtype=np.float32)
    planes = rng.standard_normal((H, F), dtype=np.float32)
    buckets = rng.standard_normal((H, D), dtype=np.float32)
    other = rng.standard_normal((B, D), dtype=np.float32)
    emb, bits = oracle.lsh_embed(ids, feat, planes, buckets, want_bits=True)
    score, emb2 = oracle.lsh_codes_embed(bits, np.arange(B, dtype=np.int32), buckets, other)
    want_score, _ = oracle.lsh_embed_score(ids, feat, planes, buckets, other)
    assert _same(emb2, emb) and _same(score, want_score)
    score3, emb3 = oracle.lsh_codes_embed(bits, np.full(B, -1, np.int32), buckets, other)
    assert np.isnan(emb3).all() and np.isnan(score3).all()
