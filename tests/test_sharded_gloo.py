"""CPU, world_size 2 (gloo): the N > 1 exchange logic of mi_oov.sharded with the per-rank compute
injected (the oracle), checked against the unsharded oracle.  Covers ragged splits, ids that all
live on one rank, empty local batches and the top-k merge."""
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

        def local_embed(local_ids, feat_local, planes_t, buckets_t):
            out = oracle.lsh_embed(local_ids.numpy(), feat_local.numpy(), planes_t.numpy(), buckets_t.numpy())
            return torch.from_numpy(out)

        table = sharded.ShardedLSHTable(torch.from_numpy(feat[lo:hi]), N, local_embed=local_embed)
        cases = {
            "ragged": np.random.default_rng(10 + rank).integers(0, N, size=300 + 77 * rank),
            "all_on_rank0": np.random.default_rng(20 + rank).integers(0, 400, size=64),
            "all_on_rank1": np.random.default_rng(30 + rank).integers(600, N, size=50),
            "empty_here": np.zeros((0,), np.int64) if rank == 0 else np.arange(900, 1001),
            "edges": np.array([0, 500, 501, 1000, 500, 0]),
        }
        ok = True
        for name, ids in cases.items():
            ids = ids.astype(np.int64)
            got = table.embed(torch.from_numpy(ids), torch.from_numpy(planes), torch.from_numpy(buckets)).numpy()
            want = oracle.lsh_embed(ids, feat, planes, buckets)
            same = got.shape == want.shape and np.array_equal(np.nan_to_num(got, nan=7.0), np.nan_to_num(want, nan=7.0))
            ok = ok and same
        # slsh over a row-sharded feature table AND a row-sharded bucket table (BASELINE config 4), two exchanges
        NB, D2 = 777, 20
        planes_s = rng.standard_normal((10, F), dtype=np.float32)  # bits_req of 777 buckets
        big = rng.standard_normal((NB, D2), dtype=np.float32)
        blo, bhi, _ = sharded.shard_bounds(NB, world, rank)

        def local_index(local_ids, feat_local, planes_t, nb):
            _, idx = oracle.slsh_embed(local_ids.numpy(), feat_local.numpy(), planes_t.numpy(),
                                       np.zeros((nb, 1), np.float32))
            return torch.from_numpy(idx)

        def local_gather(local_idx, rows):
            li = local_idx.numpy()
            out = np.full((len(li), rows.shape[1]), np.nan, np.float32)
            okm = (li >= 0) & (li < rows.shape[0])
            out[okm] = rows.numpy()[li[okm]]
            return torch.from_numpy(out)

        st = sharded.ShardedSLSHTable(torch.from_numpy(feat[lo:hi]), N, torch.from_numpy(big[blo:bhi]), NB,
                                      local_index=local_index, local_gather=local_gather)
        for name, ids in cases.items():
            ids = ids.astype(np.int64)
            if name == "edges":
                ids = np.concatenate((ids, [N + 3, -2]))  # outside the table: NaN rows, index -1
            got, gidx = st.embed(torch.from_numpy(ids), torch.from_numpy(planes_s))
            want, widx = oracle.slsh_embed(ids, feat, planes_s, big)
            ok = ok and np.array_equal(gidx.numpy(), widx)
            ok = ok and np.array_equal(np.nan_to_num(got.numpy(), nan=7.0), np.nan_to_num(want, nan=7.0))
        # top-k merge over an item-sharded catalogue
        U = rng.standard_normal((9, D), dtype=np.float32)
        E = rng.standard_normal((N, D), dtype=np.float32)
        E[700] = E[3]  # a tie across shards -> the lower global index wins
        k = 5
        lv, li = oracle.score_topk(U, E[lo:hi], k)
        mv, mi = sharded.merge_topk(torch.from_numpy(lv), torch.from_numpy(li + lo), k)
        wv, wi = oracle.score_topk(U, E, k)
        ok = ok and np.array_equal(mi.numpy(), wi) and np.array_equal(mv.numpy(), wv)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_exchange_lookup_world2():
    from oracle import oov_oracle
    oov_oracle.build()
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


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
