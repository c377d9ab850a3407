"""Round 4: the evaluation loop's rows either side of model.predict -- mi_oov_eval_rows_build (the layout of
NegSampleEvalDataLoader's batches, R/data/dataloader/general_dataloader.py:157-190 + abstract_dataloader.py:227-235),
mi_oov_segment_dedup (the one-entry-per-pair effect of the reference's dense scatter, R/inductive/evaluator.py:118-134),
mi_oov_topk_hits_range, and SampledRankingEvaluator.eval_group (three rankings for nine collectors) against the
nine eval_batch_collect calls it replaces.  The vectorised metric arithmetic against the row loops it replaces."""
import numpy as np
import pytest
import torch


def _collate_like_reference(user_ids, pos_per_user, neg_per_user):
    """Pure-Python restatement of NegSampleEvalDataLoader.collate_fn for one batch: per user `_neg_sampling` returns the
    user's positive rows followed by its sampled rows (`new_data = inter_feat.repeat(times);
    new_data[iid][pos_inter_num:] = neg_item_ids`), `cat_interactions` concatenates the users."""
    row_user, row_item, idx_list, pos_u, pos_i = [], [], [], [], []
    for idx, (uid, pos, neg) in enumerate(zip(user_ids, pos_per_user, neg_per_user)):
        items = list(pos) + list(neg)
        row_user += [uid] * len(items)
        row_item += items
        idx_list += [idx] * len(items)
        pos_u += [idx] * len(pos)
        pos_i += list(pos)
    return (np.array(row_user, np.int64), np.array(row_item, np.int64), np.array(idx_list, np.int64),
            np.array(pos_u, np.int64), np.array(pos_i, np.int64))


def _case(rng, counts, n_neg, n_items=5000):
    counts = np.asarray(counts, np.int64)
    uids = rng.permutation(100000)[:len(counts)].astype(np.int64)
    pos = [rng.integers(1, n_items, c) for c in counts]
    neg = [rng.integers(1, n_items, c * n_neg) for c in counts]
    pos_ptr = np.concatenate(([0], np.cumsum(counts))).astype(np.int64)
    cat = lambda parts: np.concatenate(parts + [np.zeros(0, np.int64)]).astype(np.int64)  # noqa: E731
    return uids, pos, neg, pos_ptr, cat(pos), cat(neg)


@pytest.mark.parametrize("counts,n_neg", [([1, 3, 2, 1], 5), ([4], 250), ([1, 0, 2, 0], 3), ([2, 2, 2], 0), ([], 7), ([30, 1, 1], 50)])
def test_oracle_eval_rows_is_the_dataloader_layout(counts, n_neg, oracle):
    rng = np.random.default_rng(len(counts) + n_neg)
    uids, pos, neg, pos_ptr, pos_items, neg_items = _case(rng, counts, n_neg)
    ru, ri, idx_list, pu, pi = _collate_like_reference(uids, pos, neg)
    row_user, row_item, seg_ptr, pos_user = oracle.eval_rows_build(pos_ptr, uids, pos_items, neg_items, n_neg)
    assert np.array_equal(row_user, ru) and np.array_equal(row_item, ri)
    assert np.array_equal(pos_user, pu) and np.array_equal(pos_items, pi)
    # seg_ptr is the CSR of the reference's idx_list (the batch-local user index of every row)
    want_ptr = np.concatenate(([0], np.cumsum(np.bincount(idx_list, minlength=len(counts))))).astype(np.int64)
    assert np.array_equal(seg_ptr, want_ptr)


def _dedup_np(cols, seg_ptr):
    out = cols.copy()
    for s in range(len(seg_ptr) - 1):
        seg = cols[seg_ptr[s]:seg_ptr[s + 1]]
        _, first = np.unique(seg, return_index=True)
        dead = np.ones(len(seg), bool)
        dead[first] = False
        out[seg_ptr[s]:seg_ptr[s + 1]][dead] = -1
    return out


def test_oracle_segment_dedup_and_hits_range(oracle):
    rng = np.random.default_rng(3)
    lens = [0, 7, 300, 1, 64]
    seg_ptr = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    cols = rng.integers(0, 40, seg_ptr[-1]).astype(np.int64)  # few distinct columns: many repeats
    assert np.array_equal(oracle.segment_dedup(cols, seg_ptr), _dedup_np(cols, seg_ptr))
    # hits with a range on the positives == compacting the positives first (what the generic collector does)
    S, k = 6, 5
    idx = rng.integers(-1, 30, (S, k)).astype(np.int64)
    plen = rng.integers(0, 6, S)
    pos_ptr = np.concatenate(([0], np.cumsum(plen))).astype(np.int64)
    pos = rng.integers(0, 30, pos_ptr[-1]).astype(np.int64)
    for lo, hi in ((0, 1 << 62), (10, 1 << 62), (0, 10), (5, 20)):
        keep = (pos >= lo) & (pos < hi)
        grp = np.repeat(np.arange(S), plen)[keep]
        cptr = np.concatenate(([0], np.cumsum(np.bincount(grp, minlength=S)))).astype(np.int64)
        idx_in = np.where((idx >= lo) & (idx < hi), idx, -1)  # the ranked columns come from the same slice
        want = oracle.topk_hits(idx_in, cptr, np.concatenate((pos[keep], np.zeros(1, np.int64))))
        got = oracle.topk_hits_range(idx_in, pos_ptr, np.concatenate((pos, np.zeros(1, np.int64))), lo, hi)
        assert np.array_equal(got, want), (lo, hi)


def _metrics_with_row_loops(rec, topk, metrics):
    """The row loops topk_metrics had before round 4 (themselves pinned on the reference's Evaluator by eval_uni.npz)."""
    pos_index = rec[:, :-1].astype(bool)
    pos_len = rec[:, -1].astype(np.int64)
    U, K = pos_index.shape
    ranks = np.arange(1, K + 1)
    out = {}
    for name in metrics:
        if name == "mrr":
            first = pos_index.argmax(axis=1)
            val = np.zeros((U, K))
            for row, j in enumerate(first):
                if pos_index[row, j]:
                    val[row, j:] = 1.0 / (j + 1)
        elif name == "map":
            pre = pos_index.cumsum(axis=1) / ranks
            sum_pre = np.cumsum(pre * pos_index.astype(float), axis=1)
            actual = np.minimum(pos_len, K)
            val = np.zeros((U, K))
            for row, n in enumerate(actual):
                denom = ranks.copy()
                denom[n:] = denom[n - 1]
                val[row] = sum_pre[row] / denom
        else:
            idcg_len = np.minimum(pos_len, K)
            idcg = np.tile(np.cumsum(1.0 / np.log2(ranks + 1.0)), (U, 1))
            for row, n in enumerate(idcg_len):
                idcg[row, n:] = idcg[row, n - 1]
            dcg = np.cumsum(np.where(pos_index, 1.0 / np.log2(ranks + 1.0), 0), axis=1)
            with np.errstate(divide="ignore", invalid="ignore"):
                val = dcg / idcg
        nan_rows = np.isnan(val).any(axis=1)
        avg = val[~nan_rows].mean(axis=0) if (~nan_rows).any() else np.full(K, np.nan)
        for k in topk:
            out[f"{name}@{k}"] = float(avg[k - 1])
    return out


def test_vectorised_metrics_equal_the_row_loops():
    import mi_oov
    rng = np.random.default_rng(0)
    U, K = 500, 20
    rec = np.zeros((U, K + 1), np.int32)
    rec[:, :K] = rng.random((U, K)) < 0.15
    rec[:, K] = rng.integers(0, 30, U)         # 0 positives (the reference's wrap-around row), fewer / more than K
    rec[:5, :K] = 0
    topk = [1, 5, 10, 20]
    want = _metrics_with_row_loops(rec, topk, ("mrr", "map", "ndcg"))
    with np.errstate(divide="ignore", invalid="ignore"):
        got = mi_oov.evaluator.topk_metrics(rec, topk, ("mrr", "map", "ndcg"), decimal_place=15)
    for name, w in want.items():
        assert got[name] == round(w, 15), name


def _numpy_means(rec, topk_all=True):
    """NumPy's own unrounded means of the six curves (the code of evaluator.topk_metrics without the rounding)."""
    pos_index = rec[:, :-1].astype(bool)
    pos_len = rec[:, -1].astype(np.int64)
    U, K = pos_index.shape
    ranks = np.arange(1, K + 1)
    vals = {}
    with np.errstate(divide="ignore", invalid="ignore"):
        vals[0] = np.cumsum(pos_index, axis=1) / pos_len.reshape(-1, 1)
    vals[1] = (np.cumsum(pos_index, axis=1) > 0).astype(int)
    vals[2] = pos_index.cumsum(axis=1) / ranks
    base = np.cumsum(1.0 / np.log2(ranks + 1.0))
    idcg = base[np.minimum(np.arange(K)[None, :], np.minimum(pos_len, K)[:, None] - 1)]
    vals[3] = np.cumsum(np.where(pos_index, 1.0 / np.log2(ranks + 1.0), 0), axis=1) / idcg
    first = pos_index.argmax(axis=1)
    has = pos_index[np.arange(U), first]
    vals[4] = np.where((np.arange(K)[None, :] >= first[:, None]) & has[:, None], (1.0 / (first + 1))[:, None], 0.0)
    pre = pos_index.cumsum(axis=1) / ranks
    sum_pre = np.cumsum(pre * pos_index.astype(float), axis=1)
    vals[5] = sum_pre / ranks[np.minimum(np.arange(K)[None, :], np.minimum(pos_len, K)[:, None] - 1)]
    return vals


def _random_rec(rng, U, K):
    rec = np.zeros((U, K + 1), np.int32)
    rec[:, :K] = rng.random((U, K)) < 0.12
    rec[:, K] = rng.integers(0, 2 * K, U)   # users without a positive, with fewer / more positives than K
    rec[rec[:, K] == 0, :K] = 0
    return rec


@pytest.mark.parametrize("U,K", [(3000, 10), (517, 20), (40, 1), (1, 7)])
def test_oracle_metric_sums_are_numpys_means_bit_for_bit(U, K, oracle):
    """oov_topk_metric_sums: curves with NumPy's operation order, sums over users in user order == what
    `val[~nan_rows].mean(axis=0)` (base_metric.py:60-84) computes, for every side (all / old / new users)."""
    rng = np.random.default_rng(U + K)
    rec = _random_rec(rng, U, K)
    uids = rng.integers(0, 100, U)
    ranks = np.arange(1, K + 1)
    disc = 1.0 / np.log2(ranks + 1.0)
    sums, counts = oracle.topk_metric_sums(rec, disc, np.cumsum(disc), uids, 50)
    vals = _numpy_means(rec)
    for side, sel in enumerate((np.ones(U, bool), uids < 50, uids >= 50)):
        for m in range(6):
            v = vals[m][sel]
            keep = ~np.isnan(v).any(axis=1)
            assert counts[side, m] == keep.sum()
            if keep.any():
                want = v[keep].mean(axis=0)
                got = sums[side, m] / counts[side, m]
                assert np.array_equal(got.view(np.uint64), np.asarray(want, np.float64).view(np.uint64)), (side, m)


# ---------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("U,K", [(36000, 10), (517, 20), (1, 7), (300, 100), (0, 5)])
def test_gpu_metric_sums_vs_oracle(U, K, oracle, dev):
    from mi_oov import ops
    rng = np.random.default_rng(U + K)
    rec = _random_rec(rng, U, K)
    uids = rng.integers(0, 100, U)
    ranks = np.arange(1, K + 1)
    disc = 1.0 / np.log2(ranks + 1.0)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    for with_uids in (True, False):
        s, c = ops.topk_metric_sums(T(rec), T(disc), T(np.cumsum(disc)), T(uids) if with_uids else None, 50)
        ws, wc = oracle.topk_metric_sums(rec, disc, np.cumsum(disc), uids if with_uids else None, 50)
        assert np.array_equal(c.cpu().numpy(), wc)
        assert np.array_equal(s.cpu().numpy().view(np.uint64), ws.view(np.uint64))

@pytest.mark.gpu
@pytest.mark.parametrize("counts,n_neg", [([1, 3, 2, 1], 5), ([4], 250), ([1, 0, 2, 0], 3), ([2, 2, 2], 0), ([], 7),
                                          ("many", 250), ("heavy", 100)])
def test_gpu_eval_rows_build_vs_oracle(counts, n_neg, oracle, dev):
    from mi_oov import ops
    rng = np.random.default_rng(11)
    if counts == "many":       # more users than the launch has waves: the grid-stride loop
        counts = rng.integers(1, 4, 20000)
    elif counts == "heavy":    # one user with thousands of rows between light ones
        counts = np.array([1, 700, 2, 1])
    uids, pos, neg, pos_ptr, pos_items, neg_items = _case(rng, counts, n_neg)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    got = ops.eval_rows_build(T(pos_ptr), T(uids), T(pos_items), T(neg_items), n_neg, want_pos_user=True)
    want = oracle.eval_rows_build(pos_ptr, uids, pos_items, neg_items, n_neg)
    for g, w, name in zip(got, want, ("row_user", "row_item", "seg_ptr", "pos_user")):
        assert np.array_equal(g.cpu().numpy(), w), name
    ru, ri, sp = ops.eval_rows_build(T(pos_ptr), T(uids), T(pos_items), T(neg_items), n_neg)
    assert np.array_equal(ru.cpu().numpy(), want[0]) and np.array_equal(sp.cpu().numpy(), want[2])


def _murmur_inverse(h):
    """x with dedup_mix(x) == h (csrc/evalrows.hip: murmur3's 64-bit finalizer is a bijection)."""
    M = (1 << 64) - 1
    inv1, inv2 = pow(0xff51afd7ed558ccd, -1, 1 << 64), pow(0xc4ceb9fe1a85ec53, -1, 1 << 64)
    x = h
    x ^= x >> 33
    x = (x * inv2) & M
    x ^= x >> 33
    x = (x * inv1) & M
    x ^= x >> 33
    return x


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["small", "long", "all_equal", "colliding"])
def test_gpu_segment_dedup_vs_oracle(case, oracle, dev):
    """small: many short segments with repeats (one table pass each); long: segments of 5 000 .. 60 000 candidates (several
    hash partitions per segment, tables near their planned load); all_equal: one column 30 000 times; colliding: 9 000
    DISTINCT columns built to agree in every bit the table and the partition function look at except the slot bits -- more
    than the table holds, in one partition however often it is split: the quadratic last resort, which must terminate."""
    from mi_oov import ops, _cabi
    rng = np.random.default_rng(len(case))
    if case == "small":
        lens = np.concatenate((rng.integers(0, 600, 3000), [0, 0, 1, 2047, 2048, 2049]))
        cols = rng.integers(0, 400, int(lens.sum())).astype(np.int64)
    elif case == "long":
        lens = np.array([5000, 60000, 3, 20481, 0, 8192])
        cols = rng.integers(0, 50000, int(lens.sum())).astype(np.int64)
    elif case == "all_equal":
        lens = np.array([30000, 5])
        cols = np.full(int(lens.sum()), 77, np.int64)
    else:
        top = int(rng.integers(1, 1 << 43)) << 20          # bits 20..63 of the mix: the same for every column
        hashes = [top | int(v) for v in rng.permutation(1 << 20)[:9000]]
        distinct = np.array([_murmur_inverse(h) for h in hashes], np.uint64).view(np.int64)
        cols = np.concatenate((distinct, distinct[rng.integers(0, 9000, 3000)]))   # + 3000 repeats behind them
        cols = cols[rng.permutation(len(cols))]
        lens = np.array([len(cols)])
    seg_ptr = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    got = ops.segment_dedup(T(cols), T(seg_ptr)).cpu().numpy()
    want = _dedup_np(cols, seg_ptr) if case == "long" else oracle.segment_dedup(cols, seg_ptr)
    assert np.array_equal(got, want)
    if case == "small":
        assert np.array_equal(want, _dedup_np(cols, seg_ptr))
        c = T(cols)  # out must not be the input
        rc = _cabi.lib().mi_oov_segment_dedup(c.data_ptr(), T(seg_ptr).data_ptr(), len(lens), c.data_ptr(), None)
        assert rc == -7 and b"output" in _cabi.lib().mi_oov_strerror(rc)


@pytest.mark.gpu
def test_gpu_topk_hits_range_vs_oracle(oracle, dev):
    from mi_oov import ops
    rng = np.random.default_rng(5)
    S, k = 700, 20
    idx = rng.integers(-1, 300, (S, k)).astype(np.int64)
    plen = rng.integers(0, 9, S)
    pos_ptr = np.concatenate(([0], np.cumsum(plen))).astype(np.int64)
    pos = rng.integers(0, 300, pos_ptr[-1]).astype(np.int64)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    for lo, hi in ((0, None), (100, None), (0, 100), (50, 200)):
        got = ops.topk_hits(T(idx), T(pos_ptr), T(pos), lo, hi).cpu().numpy()
        want = oracle.topk_hits_range(idx, pos_ptr, pos, lo, (1 << 62) if hi is None else hi)
        assert np.array_equal(got, want), (lo, hi)


@pytest.mark.gpu
@pytest.mark.parametrize("n_users,n_neg,n_items", [(300, 20, 150), (40, 250, 3000), (5, 3, 8)])
def test_gpu_eval_group_equals_the_nine_collectors(n_users, n_neg, n_items, dev):
    """eval_group (dedup kernel + three rankings + range hits, user slices on the host) == eval_batch (nine collector calls,
    each sorting, compacting duplicates and positives): same metric dictionaries AND the same rec.topk blocks per
    collector.  Few items -> every user has repeated candidates, some users fewer than k distinct ones."""
    import mi_oov
    from mi_oov import ops
    rng = np.random.default_rng(n_users)
    counts = rng.integers(1, 5, n_users)
    uids, pos, neg, pos_ptr, pos_items, neg_items = _case(rng, counts, n_neg, n_items)
    uids = rng.permutation(2 * n_users)[:n_users].astype(np.int64)  # about half old (< n_users), half new
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    row_user, row_item, seg_ptr, pos_user = ops.eval_rows_build(T(pos_ptr), T(uids), T(pos_items), T(neg_items), n_neg, want_pos_user=True)
    M = row_item.numel()
    # a score that is a function of (user, item) -- as model.predict's is: duplicates of a pair score the same
    scores = torch.sin(row_user.double() * 12.9898 + row_item.double() * 78.233).float()
    scores[rng.integers(0, M, 5)] = float("nan")
    row_idx = torch.repeat_interleave(torch.arange(n_users, device=dev), (seg_ptr[1:] - seg_ptr[:-1]))
    topk, metrics = [1, 5, 10], ("recall", "mrr", "ndcg", "hit", "precision", "map")
    kw = dict(n_old_users=n_users, n_old_items=n_items // 2)
    a = mi_oov.evaluator.SampledRankingEvaluator(topk, metrics, **kw)
    b = mi_oov.evaluator.SampledRankingEvaluator(topk, metrics, **kw)
    a.eval_batch(scores, T(uids), row_idx, row_item, pos_user, T(pos_items))
    b.eval_group(scores, T(uids), row_item, seg_ptr, T(pos_ptr), T(pos_items))
    blocks_a = {name: torch.cat(c.blocks).cpu().numpy() for name, c in a.collectors.items() if c.blocks}
    blocks_b = b._group_blocks()
    for name, blk in blocks_a.items():
        assert np.array_equal(blk, blocks_b[name]), name
    for name in set(blocks_b) - set(blocks_a):
        assert blocks_b[name].shape[0] == 0, name
    b.eval_group(scores, T(uids), row_item, seg_ptr, T(pos_ptr), T(pos_items))  # (consumed by _group_blocks above)
    with np.errstate(divide="ignore", invalid="ignore"):
        ra, rb = a.evaluate(), b.evaluate()   # a: NumPy on the host; b: mi_oov_topk_metric_sums on the device
    assert set(ra) == set(rb) and "overall" in ra
    for name in ra:
        for key, want in ra[name].items():
            assert rb[name][key] == want or (np.isnan(want) and np.isnan(rb[name][key])), (name, key)


@pytest.mark.gpu
def test_gpu_eval_group_never_syncs(dev):
    import mi_oov
    from mi_oov import ops
    rng = np.random.default_rng(1)
    counts = rng.integers(1, 4, 200)
    uids, pos, neg, pos_ptr, pos_items, neg_items = _case(rng, counts, 30)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    args = [T(pos_ptr), T(uids), T(pos_items), T(neg_items)]
    ev = mi_oov.evaluator.SampledRankingEvaluator([10], ("recall",), n_old_users=50000, n_old_items=2500)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        row_user, row_item, seg_ptr = ops.eval_rows_build(*args, 30)
        scores = (row_user * 31 + row_item).float()
        ev.eval_group(scores, args[1], row_item, seg_ptr, args[0], args[2])
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert "overall" in ev.evaluate()
