"""Ranking evaluation of sampled (uni-N) batches: SURVEY.md section 8f rank 1.

Mirrors, for the batches the paper's driver actually produces (eval mode uni250, S/run_recbole.py:214-221):

    InductiveEvaluator.neg_sample_batch_eval   R/inductive/evaluator.py:118-134   scores -> dense -inf matrix
    Collector.eval_batch_collect ("rec.topk")  R/evaluator/collector.py:158-167   topk + positive matrix + gather
    Evaluator / TopkMetric                     R/evaluator/metrics.py:36-235, base_metric.py:60-84

without ever building the [users, items] score matrix or the [users, items] positive matrix: the batch stays
sparse (score, user row, item column), `mi_oov_segment_topk` ranks each user's candidates and
`mi_oov_topk_hits` emits the rec.topk block.  The metric arithmetic itself is host-side NumPy in the reference
(float64, rounded to `metric_decimal_place`) and is the same here.

The reference's *filtered* collectors (old/new user x item slices, R/inductive/filtered_collector.py) are not
mirrored bug for bug: on sampled batches `FastUserItemCollectorFilter.map_user_items` indexes the per-row user
column with per-user batch indices (collector_filter.py:211), `apply_score_filter` masks item columns IN PLACE on
the score matrix every later collector shares (:169-172), and new-item positives are shifted by n_old_items
against unshifted recommendation columns (:252).  `slice_eval` gives the evidently intended quantities (users
restricted to old/new, candidate columns restricted to old/new items) with the same kernels.
"""
import numpy as np
import torch

from . import ops


def _csr_ptr(group, n_groups):
    """int64[n_groups+1] row pointer of a non-decreasing group index."""
    counts = torch.bincount(group, minlength=n_groups)
    return torch.cat((torch.zeros(1, dtype=torch.int64, device=group.device), torch.cumsum(counts, 0)))


class RankingCollector:
    """The 'rec.topk' resource of recbole's Collector for sampled batches."""

    def __init__(self, topk):
        self.topk = sorted(int(k) for k in topk)
        self.blocks = []

    def eval_batch_collect(self, origin_scores, row_idx, col_idx, positive_u, positive_i, n_users=None,
                           col_lo=0, col_hi=None, user_mask=None):
        """origin_scores f32[M] = model.predict of the batch rows; row_idx i64[M] the batch-local user index of
        every row and col_idx i64[M] its item id (what the reference scatters with `scores[row_idx, col_idx] =
        origin_scores`); positive_u / positive_i the batch's positives.  Optional slice: candidate and positive
        columns restricted to [col_lo, col_hi), users restricted by the bool mask `user_mask[n_users]`."""
        dev = origin_scores.device
        row_idx, col_idx = row_idx.to(dev), col_idx.to(dev)
        positive_u, positive_i = positive_u.to(dev), positive_i.to(dev)
        if n_users is None:
            n_users = int(positive_u[-1]) + 1 if positive_u.numel() else 0  # batch_user_num (evaluator.py:129)
        if row_idx.numel() > 1 and bool((row_idx[1:] < row_idx[:-1]).any()):
            order = torch.sort(row_idx, stable=True).indices
            row_idx, col_idx, origin_scores = row_idx[order], col_idx[order], origin_scores[order]
        if positive_u.numel() > 1 and bool((positive_u[1:] < positive_u[:-1]).any()):
            order = torch.sort(positive_u, stable=True).indices
            positive_u, positive_i = positive_u[order], positive_i[order]
        # the reference's scatter `scores[row_idx, col_idx] = origin_scores` keeps ONE score per (user, item); which
        # of several duplicates survives is undefined there -- here the first (a positive, if the item is one)
        key = row_idx * (int(col_idx.max()) + 1 if col_idx.numel() else 1) + col_idx
        ks, order = torch.sort(key, stable=True)
        dup = ks[1:] == ks[:-1]
        if bool(dup.any()):
            first = torch.ones_like(ks, dtype=torch.bool)
            first[1:] = ~dup
            keep = torch.sort(order[first]).values
            row_idx, col_idx, origin_scores = row_idx[keep], col_idx[keep], origin_scores[keep]
        hi = col_hi if col_hi is not None else (1 << 62)
        if col_lo > 0 or col_hi is not None:  # positives outside the item slice do not count
            keep = (positive_i >= col_lo) & (positive_i < hi)
            positive_u, positive_i = positive_u[keep], positive_i[keep]
        seg_ptr, pos_ptr = _csr_ptr(row_idx, n_users), _csr_ptr(positive_u, n_users)
        _, idx = ops.segment_topk(origin_scores, col_idx, seg_ptr, self.topk[-1], col_lo, col_hi)
        rec = ops.topk_hits(idx, pos_ptr, positive_i)
        if user_mask is not None:
            rec = rec[user_mask.to(dev)]
        self.blocks.append(rec)
        return rec

    def get_data_struct(self):
        rec = torch.cat(self.blocks) if self.blocks else torch.zeros((0, self.topk[-1] + 1), dtype=torch.int32)
        self.blocks = []
        return rec


class FullSortCollector(RankingCollector):
    """'rec.topk' for full-sort batches (InductiveEvaluator.eval_batch, R/inductive/evaluator.py:70-96): the scores
    U @ E.T are never materialised; column 0 (padding) and each user's history are excluded inside the fused top-k
    (`mi_oov_score_topk_excl`), the hit flags come from `mi_oov_topk_hits`."""

    def eval_batch_collect_full(self, U, E, history_index, positive_u, positive_i, n_skip_low=1):
        """U f32[n_users, D] user rows of the batch, E f32[n_items, D] the catalogue (model.get_*_embedding output);
        history_index = (rows, cols) as produced by FullSortEvalDataLoader, or None."""
        dev = U.device
        n_users = U.shape[0]
        if history_index is None:
            hu = hi = torch.zeros((0,), dtype=torch.int64, device=dev)
        else:
            hu, hi = (t.to(dev) for t in history_index)
        key = hu * (E.shape[0] + 1) + hi
        order = torch.sort(key).indices  # by user, then column: the exclusion lists must ascend
        hu, hi = hu[order], hi[order]
        positive_u, positive_i = positive_u.to(dev), positive_i.to(dev)
        if positive_u.numel() > 1 and bool((positive_u[1:] < positive_u[:-1]).any()):
            o = torch.sort(positive_u, stable=True).indices
            positive_u, positive_i = positive_u[o], positive_i[o]
        # the item table is the same for every user batch of an evaluation run: prepared once per table version
        cat = getattr(self, "_catalogue", None)
        if cat is None or not cat.fresh(E):
            cat = self._catalogue = ops.TopkCatalogue.of(E)
        _, idx = ops.score_topk_excl(U, cat if cat is not None else E, self.topk[-1], _csr_ptr(hu, n_users), hi,
                                     n_skip_low=n_skip_low)
        rec = ops.topk_hits(idx, _csr_ptr(positive_u, n_users), positive_i)
        self.blocks.append(rec)
        return rec


def topk_metrics(rec_topk, topk, metrics=("recall", "hit", "precision", "ndcg", "mrr", "map"), decimal_place=4):
    """recbole's TopkMetric family on a rec.topk block int[U, kmax+1] (metrics.py:36-235): per-user curves for
    k = 1..kmax, users whose curve contains NaN (no positives) dropped, mean, rounded."""
    rec = rec_topk.detach().cpu().numpy() if torch.is_tensor(rec_topk) else np.asarray(rec_topk)
    pos_index = rec[:, :-1].astype(bool)
    pos_len = rec[:, -1].astype(np.int64)
    U, K = pos_index.shape
    ranks = np.arange(1, K + 1)
    out = {}
    for name in metrics:
        name = name.lower()
        if name == "recall":
            with np.errstate(divide="ignore", invalid="ignore"):
                val = np.cumsum(pos_index, axis=1) / pos_len.reshape(-1, 1)
        elif name == "hit":
            val = (np.cumsum(pos_index, axis=1) > 0).astype(int)
        elif name == "precision":
            val = pos_index.cumsum(axis=1) / ranks
        elif name == "mrr":
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
                denom[n:] = denom[n - 1]  # n = 0 wraps to the last rank, as in the reference
                val[row] = sum_pre[row] / denom
        elif name == "ndcg":
            idcg_len = np.minimum(pos_len, K)
            idcg = np.tile(np.cumsum(1.0 / np.log2(ranks + 1.0)), (U, 1))
            for row, n in enumerate(idcg_len):
                idcg[row, n:] = idcg[row, n - 1]
            dcg = np.cumsum(np.where(pos_index, 1.0 / np.log2(ranks + 1.0), 0), axis=1)
            val = dcg / idcg
        else:
            raise NotImplementedError(f"metric {name}: only the rec.topk family is built")
        nan_rows = np.isnan(val).any(axis=1)
        avg = val[~nan_rows].mean(axis=0) if (~nan_rows).any() else np.full(K, np.nan)
        for k in topk:
            out[f"{name}@{k}"] = round(float(avg[k - 1]), decimal_place)
    return out


class SampledRankingEvaluator:
    """evaluate_model of InductiveEvaluator for sampled batches: overall metrics plus the intended old/new slices."""

    def __init__(self, topk, metrics=("recall", "hit", "ndcg", "mrr"), n_old_users=None, n_old_items=None,
                 decimal_place=4):
        self.topk, self.metrics, self.decimal_place = list(topk), list(metrics), decimal_place
        self.n_old_users, self.n_old_items = n_old_users, n_old_items
        names = ["overall"]  # the reference's seven collectors (evaluator.py:41-49) + the two item-only filters (:29-30)
        if n_old_users is not None:
            names += ["old_users", "new_users"]
            if n_old_items is not None:
                names += ["old_old", "old_new", "new_old", "new_new"]
        if n_old_items is not None:
            names += ["old_items", "new_items"]
        self.collectors = {n: RankingCollector(self.topk) for n in names}

    def eval_batch(self, origin_scores, user_ids, row_idx, col_idx, positive_u, positive_i):
        """user_ids i64[n_users]: the real id of every batch-local user (decides old / new)."""
        n_users = user_ids.numel()
        for name, col in self.collectors.items():
            kw = {}
            if name in ("old_items", "new_items"):
                kw["col_hi" if name == "old_items" else "col_lo"] = self.n_old_items
            elif name != "overall":
                u_old = name.startswith("old")
                mask = user_ids < self.n_old_users if u_old else user_ids >= self.n_old_users
                kw["user_mask"] = mask
                if name in ("old_old", "new_old"):
                    kw["col_hi"] = self.n_old_items
                elif name in ("old_new", "new_new"):
                    kw["col_lo"] = self.n_old_items
                if not bool(mask.any()):
                    continue
            col.eval_batch_collect(origin_scores, row_idx, col_idx, positive_u, positive_i, n_users=n_users, **kw)

    def evaluate(self):
        res = {}
        for name, col in self.collectors.items():
            rec = col.get_data_struct()
            if rec.shape[0]:
                res[name] = topk_metrics(rec, self.topk, self.metrics, self.decimal_place)
        return res
