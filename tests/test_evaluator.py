"""Sampled-ranking evaluation against what the REAL reference's Collector + Evaluator
produced for the same sparse batches (tests/golden/eval_uni.npz, made by make_golden_eval.py)."""
import os

import numpy as np
import pytest
import torch


def _batches(z):
    for b in range(int(z["n_batches"])):
        yield (z[f"b{b}_scores"], z[f"b{b}_row_idx"], z[f"b{b}_col_idx"], z[f"b{b}_pos_u"], z[f"b{b}_pos_i"])


def _ptr(group, n):
    return np.concatenate(([0], np.cumsum(np.bincount(group, minlength=n)))).astype(np.int64)


def test_oracle_rec_topk_and_metrics_match_reference(golden, oracle):
    """CPU: oracle segment_topk + topk_hits == the reference collector's rec.topk block; the host-side metric
    arithmetic == the reference Evaluator's values."""
    import mi_oov
    z = golden("eval_uni.npz")
    kmax = int(z["topk"].max())
    blocks = []
    for scores, row, col, pu, pi in _batches(z):
        n = int(pu[-1]) + 1
        _, idx = oracle.segment_topk(scores, col, _ptr(row, n), kmax)
        blocks.append(oracle.topk_hits(idx, _ptr(pu, n), pi))
    rec = np.concatenate(blocks)
    assert np.array_equal(rec, z["rec_topk"])
    got = mi_oov.evaluator.topk_metrics(rec, [int(k) for k in z["topk"]])
    for name, want in zip(z["metric_names"], z["metric_values"]):
        assert got[str(name)] == pytest.approx(float(want), abs=1e-12), name


def test_oracle_segment_topk_edges(oracle):
    scores = np.array([1, 5, np.nan, 2, 9, 7, 7], np.float32)
    cols = np.array([10, 11, 12, 13, 14, 15, 16])
    seg = np.array([0, 4, 4, 7])  # an empty segment in the middle
    v, i = oracle.segment_topk(scores, cols, seg, 3)
    assert i.tolist() == [[12, 11, 13], [-1, -1, -1], [14, 15, 16]]  # NaN first, ties -> earlier candidate
    v, i = oracle.segment_topk(scores, cols, seg, 3, 11, 16)        # column filter [11, 16)
    assert i.tolist() == [[12, 11, 13], [-1, -1, -1], [14, 15, -1]] and np.isneginf(v[2, 2])


@pytest.mark.gpu
def test_gpu_rec_topk_matches_reference(golden, oracle, dev):
    import mi_oov
    z = golden("eval_uni.npz")
    topk = [int(k) for k in z["topk"]]
    col = mi_oov.evaluator.RankingCollector(topk)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    for scores, row, c, pu, pi in _batches(z):
        col.eval_batch_collect(T(scores), T(row), T(c), T(pu), T(pi))
    rec = col.get_data_struct().cpu().numpy()
    assert np.array_equal(rec, z["rec_topk"])
    got = mi_oov.evaluator.topk_metrics(rec, topk)
    for name, want in zip(z["metric_names"], z["metric_values"]):
        assert got[str(name)] == pytest.approx(float(want), abs=1e-12), name


@pytest.mark.gpu
@pytest.mark.parametrize("S,lens,k", [(5, (0, 40), 10), (64, (900, 1200), 20), (3, (5000, 30000), 256), (300, (1, 8), 17),
                                      (9, (2040, 2056), 50), (7, (1500, 2048), 256)])
def test_gpu_segment_topk_vs_oracle(S, lens, k, oracle, dev):
    from mi_oov import ops
    rng = np.random.default_rng(S + k)
    n = rng.integers(lens[0], lens[1] + 1, S)
    seg = np.concatenate(([0], np.cumsum(n))).astype(np.int64)
    M = int(seg[-1])
    scores = rng.standard_normal(M).astype(np.float32)
    scores[rng.random(M) < 0.02] = np.nan
    scores[rng.random(M) < 0.05] = 0.25          # ties
    scores[rng.random(M) < 0.01] = -np.inf
    cols = rng.integers(0, 5000, M)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    for lo, hi in ((0, None), (1000, 3500)):
        v, i = ops.segment_topk(T(scores), T(cols), T(seg), k, lo, hi)
        ov, oi = oracle.segment_topk(scores, cols, seg, k, lo, hi if hi is not None else 2 ** 62)
        assert np.array_equal(i.cpu().numpy(), oi)
        assert np.array_equal(np.nan_to_num(v.cpu().numpy(), nan=7e9), np.nan_to_num(ov, nan=7e9))
    pos_u = np.sort(rng.integers(0, S, 4 * S))
    pos_i = rng.integers(0, 5000, 4 * S)
    pptr = np.concatenate(([0], np.cumsum(np.bincount(pos_u, minlength=S)))).astype(np.int64)
    assert np.array_equal(ops.topk_hits(i, T(pptr), T(pos_i)).cpu().numpy(), oracle.topk_hits(oi, pptr, pos_i))


@pytest.mark.gpu
def test_gpu_collector_dedups_and_slices(dev):
    """Duplicate (user, item) candidates count once (dense scatter semantics); item / user slices restrict
    candidates, positives and rows."""
    import mi_oov
    T = lambda a, dt=torch.int64: torch.tensor(a, dtype=dt, device=dev)  # noqa: E731
    # user 0: positive item 5 (also sampled as a negative with a higher score), user 1: positives 7 (old) and 12 (new)
    row = T([0, 0, 0, 0, 1, 1, 1, 1, 1])
    col = T([5, 9, 5, 11, 7, 12, 3, 10, 4])
    sc = T([1.0, 0.5, 9.0, 2.0, 3.0, 2.5, 9.0, 8.0, 1.0], torch.float32)
    pu, pi = T([0, 1, 1]), T([5, 7, 12])
    c = mi_oov.evaluator.RankingCollector([2])
    rec = c.eval_batch_collect(sc, row, col, pu, pi).cpu().numpy()
    assert rec.tolist() == [[0, 1, 1], [0, 0, 2]]            # user 0: 11 (2.0), 5 (1.0: first occurrence kept)
    rec = c.eval_batch_collect(sc, row, col, pu, pi, col_hi=10).cpu().numpy()    # old items only (< 10)
    assert rec.tolist() == [[1, 0, 1], [0, 1, 1]]            # user 1: 3 (9.0), 7 (3.0); positive 12 not counted
    rec = c.eval_batch_collect(sc, row, col, pu, pi, col_lo=10, user_mask=T([False, True], torch.bool)).cpu().numpy()
    assert rec.tolist() == [[0, 1, 1]]                       # user 1, new items: 10 (8.0), 12 (2.5)
    ev = mi_oov.evaluator.SampledRankingEvaluator([1, 2], ("recall", "hit"), n_old_users=1, n_old_items=10)
    ev.eval_batch(sc, T([0, 4]), row, col, pu, pi)           # user ids 0 (old) and 4 (new)
    res = ev.evaluate()
    assert set(res) == {"overall", "old_users", "new_users", "old_old", "old_new", "new_old", "new_new", "old_items", "new_items"}
    assert res["new_new"]["recall@2"] == 1.0 and res["old_old"]["hit@1"] == 1.0 and res["old_users"]["recall@2"] == 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["masked", "masked_chunked", "k_plus_hmax", "dense"])
@pytest.mark.parametrize("B,N,k,hmax", [(70, 3000, 10, 40), (9, 1300, 5, 0), (130, 4159, 20, 200), (40, 2500, 10, 600), (300, 2700, 10, 90)])
def test_gpu_full_sort_topk_with_history_vs_oracle(B, N, k, hmax, route, oracle, dev, monkeypatch):
    """Full-sort evaluation: column 0 and every user's history masked (evaluator.py:92-95) inside the top-k, on every route of
    ops.score_topk_excl -- masked: the exclusion bitmap inside the fused kernel (any history length); masked_chunked: the
    same with a workspace bound so small that the user batch is cut into 128-row chunks (the CSR addressed through a
    shifted excl_ptr); k_plus_hmax: top-(k + longest history) then a filter pass (the caller states h_max; longer
    histories than 256 - k go on to the dense route); dense: materialised chunk + bitmap + exact select, all inside the
    library (round 4: this replaced a torch.topk fallback with a Python loop over rows)."""
    from mi_oov import ops
    monkeypatch.setattr(ops, "_USE_MASKED_TOPK", route.startswith("masked"))
    if route == "masked_chunked":
        monkeypatch.setattr(ops, "_MASKED_TOPK_MAX_BYTES", 1 << 16)
    rng = np.random.default_rng(B + N)
    U = rng.standard_normal((B, 64), dtype=np.float32)
    E = rng.standard_normal((N, 64), dtype=np.float32)
    lens = rng.integers(0, hmax + 1, B)
    lens[0] = hmax
    ptr = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    cols = np.concatenate([np.sort(rng.choice(np.arange(1, N), size=n, replace=False)) for n in lens] + [np.zeros(0, np.int64)])
    # make sure masking matters: put each user's best items into its history
    _, best = oracle.score_topk(U, E, 3, 1)
    for b in range(B):
        if lens[b] >= 3:
            seg = cols[ptr[b]:ptr[b + 1]]
            seg[:3] = best[b]
            cols[ptr[b]:ptr[b + 1]] = np.sort(np.unique(np.concatenate((seg, best[b])))[:lens[b]]) if len(np.unique(seg)) == lens[b] else np.sort(seg)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    v, i = ops.score_topk_excl(T(U), T(E), k, T(ptr), T(cols.astype(np.int64)), n_skip_low=1,
                               h_max=hmax if route == "k_plus_hmax" else None)
    ov, oi = oracle.score_topk_excl(U, E, k, ptr, cols, 1)
    i = i.cpu().numpy()
    assert np.array_equal(i, oi)
    assert np.array_equal(v.cpu().numpy(), ov)
    for b in range(B):  # nothing excluded is ever recommended
        assert not np.intersect1d(i[b], cols[ptr[b]:ptr[b + 1]]).size and 0 not in i[b]


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["heavy", "nearly_all", "unsorted_dups", "wide"])
def test_gpu_masked_topk_edge_cases(case, oracle, dev):
    """mi_oov_score_topk_masked: histories far beyond 256 entries that hold every top item; rows with fewer than k
    allowed columns (missing entries are (-inf, -1)); unsorted lists with duplicates and out-of-range columns; a batch
    that spans several row blocks and a catalogue that is not a multiple of 64."""
    from mi_oov import ops
    rng = np.random.default_rng(len(case))
    B, N, k = 48, 6000, 20
    if case == "wide":
        B, N, k = 300, 20011, 50
    U = rng.standard_normal((B, 64), dtype=np.float32)
    E = rng.standard_normal((N, 64), dtype=np.float32)
    _, best = oracle.score_topk(U, E, 700, 1)
    lists = []
    for b in range(B):
        if case == "heavy":          # the 700 best columns of the user are its history
            c = best[b]
        elif case == "nearly_all":   # only k - 3 + (b % 7) columns stay allowed
            keep = rng.choice(np.arange(1, N), size=k - 3 + (b % 7), replace=False)
            c = np.setdiff1d(np.arange(0, N), keep)
        elif case == "unsorted_dups":
            c = np.concatenate((best[b][:40], best[b][:17], [-5, N, N + 99, 2 ** 40]))
            rng.shuffle(c)
        else:
            c = best[b][: int(rng.integers(0, 500))]
        lists.append(np.asarray(c, np.int64))
    ptr = np.concatenate(([0], np.cumsum([len(c) for c in lists]))).astype(np.int64)
    cols = np.concatenate(lists + [np.zeros(0, np.int64)])
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    from mi_oov import _cabi
    assert _cabi.lib().mi_oov_score_topk_masked_workspace(B, N, 64, k) > 0   # the shape is one the masked kernel takes
    v, i = ops.score_topk_excl(T(U), T(E), k, T(ptr), T(cols), n_skip_low=1)
    ov, oi = oracle.score_topk_excl(U, E, k, ptr, cols, 1)
    assert np.array_equal(i.cpu().numpy(), oi)
    assert np.array_equal(v.cpu().numpy(), ov)


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,D,k", [(33, 900, 200, 10),      # rows wider than the fused path's 128 floats
                                     (20, 5000, 64, 300),     # k beyond the fused path's 256 (the k-pass kernel)
                                     (50, 700, 64, 20),       # a catalogue of fewer than 128 k rows
                                     (7, 257, 19, 257)])      # k = N: every allowed column, then (-inf, -1)
def test_gpu_score_topk_excl_dense_route_vs_oracle(B, N, D, k, oracle, dev):
    """Shapes the fused masked kernel refuses (its workspace query returns 0) go through mi_oov_score_topk_excl_dense --
    no torch.topk, no Python loop over rows, no host synchronisation -- and equal the oracle: unsorted lists with
    duplicates and out-of-range columns, histories that hold the best columns, users left with fewer than k columns."""
    from mi_oov import ops, _cabi
    rng = np.random.default_rng(B + N + D)
    U = rng.standard_normal((B, D), dtype=np.float32)
    E = rng.standard_normal((N, D), dtype=np.float32)
    _, best = oracle.score_topk(U, E, min(N - 1, 40), 1)
    lists = []
    for b in range(B):
        c = np.concatenate((best[b][: int(rng.integers(0, best.shape[1] + 1))], rng.integers(1, N, int(rng.integers(0, 300))), [-3, N, N + 5]))
        if b % 9 == 0:  # nearly everything excluded
            c = np.setdiff1d(np.arange(N), rng.choice(np.arange(1, N), size=3 + b % 5, replace=False))
        rng.shuffle(c)
        lists.append(np.asarray(c, np.int64))
    ptr = np.concatenate(([0], np.cumsum([len(c) for c in lists]))).astype(np.int64)
    cols = np.concatenate(lists)
    assert _cabi.lib().mi_oov_score_topk_masked_workspace(B, N, D, k) == 0
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    args = (T(U), T(E), k, T(ptr), T(cols))
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        v, i = ops.score_topk_excl(*args, n_skip_low=1)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    ov, oi = oracle.score_topk_excl(U, E, k, ptr, cols, 1)
    assert np.array_equal(i.cpu().numpy(), oi)
    assert np.array_equal(v.cpu().numpy(), ov)


def test_oracle_full_sort_rec_topk_matches_reference(golden, oracle):
    import mi_oov
    z = golden("eval_full.npz")
    n_users, n_items = z["U"].shape[0], z["E"].shape[0]
    order = np.lexsort((z["hist_i"], z["hist_u"]))
    hu, hi = z["hist_u"][order], z["hist_i"][order]
    _, idx = oracle.score_topk_excl(z["U"], z["E"], int(z["topk"].max()), _ptr(hu, n_users), hi, 1)
    rec = oracle.topk_hits(idx, _ptr(z["pos_u"], n_users), z["pos_i"])
    assert np.array_equal(rec, z["rec_topk"])
    got = mi_oov.evaluator.topk_metrics(rec, [int(k) for k in z["topk"]])
    for name, want in zip(z["metric_names"], z["metric_values"]):
        assert got[str(name)] == pytest.approx(float(want), abs=1e-12), name


@pytest.mark.gpu
def test_gpu_full_sort_rec_topk_matches_reference(golden, dev):
    import mi_oov
    z = golden("eval_full.npz")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    col = mi_oov.evaluator.FullSortCollector([int(k) for k in z["topk"]])
    rec = col.eval_batch_collect_full(T(z["U"]), T(z["E"]), (T(z["hist_u"]), T(z["hist_i"])), T(z["pos_u"]), T(z["pos_i"]))
    assert np.array_equal(rec.cpu().numpy(), z["rec_topk"])


def test_topk_metrics_edge_cases():
    """Users without positives give NaN recall rows, which the reference drops from the mean of that metric only
    (base_metric.py:74-79); an all-miss block gives zeros; MAP / NDCG cap their ideal at min(positives, k)."""
    import mi_oov
    rec = np.array([[1, 0, 1, 2],      # 2 positives, hits at ranks 1 and 3
                    [0, 0, 0, 0],      # no positives at all -> recall NaN (0/0), dropped for recall only
                    [0, 1, 0, 5]])     # 5 positives, one hit at rank 2
    m = mi_oov.evaluator.topk_metrics(rec, [1, 3], ("recall", "hit", "precision", "mrr", "ndcg"), decimal_place=6)
    assert m["recall@1"] == pytest.approx((0.5 + 0.0) / 2) and m["recall@3"] == pytest.approx((1.0 + 0.2) / 2)
    assert m["hit@1"] == pytest.approx(1 / 3, abs=1e-6) and m["hit@3"] == pytest.approx(2 / 3, abs=1e-6)
    assert m["precision@3"] == pytest.approx((2 / 3 + 0 + 1 / 3) / 3, abs=1e-6)
    assert m["mrr@3"] == pytest.approx((1.0 + 0.0 + 0.5) / 3, abs=1e-6)
    idcg2 = 1 + 1 / np.log2(3)
    dcg_a = 1 + 1 / np.log2(4)
    idcg3 = idcg2 + 1 / np.log2(4)
    # the user without positives is NOT dropped from ndcg: the reference's `idcg[row, idx:] = idcg[row, idx - 1]` wraps to the
    # last column for idx = 0 (metrics.py:206-207), so its ideal is non-zero and its ndcg is 0
    want = (dcg_a / idcg2 + 0.0 + (1 / np.log2(3)) / idcg3) / 3
    assert m["ndcg@3"] == pytest.approx(want, abs=1e-6)
    zero = mi_oov.evaluator.topk_metrics(np.array([[0, 0, 3]]), [2], ("recall", "hit"))
    assert zero == {"recall@2": 0.0, "hit@2": 0.0}


@pytest.mark.gpu
def test_reference_filtered_collectors_block_for_block(dev):
    """InductiveEvaluator's seven collectors on sampled batches, bug for bug (row-indexed user ids, in-place column
    masks shared by later collectors, shifted new-item positives): every rec.topk block and every metric of the
    fixture produced by the REAL FilteredCollector + FastUserItemCollectorFilter (make_golden_eval_filtered.py)."""
    import mi_oov  # noqa: F401
    from mi_oov import evaluator
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eval_filtered.npz"))
    topk = [int(k) for k in z["topk"]]
    ev = evaluator.ReferenceFilteredEvaluator(topk, int(z["n_old_users"]), int(z["n_old_items"]),
                                              metrics=("recall", "hit", "ndcg", "mrr"))
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    for b in range(int(z["n_batches"])):
        ev.eval_batch(T(z[f"b{b}_scores"]), T(z[f"b{b}_row_uid"]), T(z[f"b{b}_row_idx"]), T(z[f"b{b}_col_idx"]),
                      T(z[f"b{b}_pos_u"]), T(z[f"b{b}_pos_i"]))
    res = ev.evaluate()
    names = [str(n) for n in z["metric_names"]]
    for name in ("overall", "old_users", "new_users", "old_old", "old_new", "new_old", "new_new"):
        want = z[name + "_rec_topk"]
        assert (z[name + "_finite_k"] >= max(topk)).all()  # the fixture's rows are fully determined
        got = ev.rec[name].cpu().numpy()
        assert got.shape == want.shape and np.array_equal(got, want), name
        assert ev.undetermined[name] == 0
        for mname, mval in zip(names, z[name + "_metric_values"]):
            assert abs(res[name][mname] - float(mval)) < 1e-9, (name, mname)
    # and the two forms differ where the reference's quirks bite: the intended new_new slice finds new-item positives
    # among new-item candidates, the reference compares shifted positives with unshifted columns
    intended = evaluator.SampledRankingEvaluator(topk, n_old_users=int(z["n_old_users"]), n_old_items=int(z["n_old_items"]))
    for b in range(int(z["n_batches"])):
        intended.eval_batch(T(z[f"b{b}_scores"]), T(z[f"b{b}_uid_of"]), T(z[f"b{b}_row_idx"]), T(z[f"b{b}_col_idx"]),
                            T(z[f"b{b}_pos_u"]), T(z[f"b{b}_pos_i"]))
    ires = intended.evaluate()
    assert ires["overall"] == res["overall"]  # the unfiltered collector is the same in both
    assert ires["new_new"]["recall@10"] > res["new_new"]["recall@10"]
