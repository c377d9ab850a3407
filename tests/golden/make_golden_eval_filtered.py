#!/usr/bin/env python3
"""Filtered-collector fixtures: the REAL reference's FilteredCollector + FastUserItemCollectorFilter.

    python tests/golden/make_golden_eval_filtered.py        (build container only; needs /root/reference)

Reproduces InductiveEvaluator.evaluate_model's inner loop (R/inductive/evaluator.py:163-178) on uni-N sampled batches:
the batch's scores are scattered into a dense [users, items] matrix of -inf (neg_sample_batch_eval, :118-134) and the
SAME tensor is handed to the seven collectors in the order of the evaluator's dict (:41-49): overall, old_users,
new_users, old_old, old_new, new_old, new_new.  Recorded per collector: the rec.topk blocks and the metric values.

What the fixture shows (R/inductive/collector_filter.py, R/inductive/filtered_collector.py):
  * map_user_items looks the user id of a positive up with `user_ids[users]` (:211): `user_ids` is the per-ROW user
    column of the batch, `users` the per-USER batch index, so user u is classified old / new by the id in ROW u;
  * apply_score_filter masks item columns IN PLACE on the shared matrix, and chooses the side by return_old_USERS
    (:169-172): old_old and old_new both mask the NEW columns, new_old and new_new then mask the OLD ones -- after that
    every column is -inf and the top-k of new_old / new_new is whatever the random column permutation
    (filtered_collector.py:39-40) puts first;
  * new-item positives are shifted by n_old_items (:252) against unshifted recommendation columns.
Rows whose top-k contains -inf entries are therefore random draws; `finite_k` records how many leading entries of each
row are determined.  Output: tests/golden/eval_filtered.npz.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
for alias, typ in (("float", float), ("int", int), ("bool", bool)):  # NumPy-2-removed aliases used by metrics.py
    if not hasattr(np, alias):
        setattr(np, alias, typ)

import torch  # noqa: E402
from recbole.evaluator.collector import Collector  # noqa: E402
from recbole.evaluator.evaluator import Evaluator  # noqa: E402
from recbole.inductive.collector_filter import FastUserItemCollectorFilter  # noqa: E402
from recbole.inductive.filtered_collector import FilteredCollector  # noqa: E402

from make_golden_eval import FakeConfig  # noqa: E402

NAMES = ["overall", "old_users", "new_users", "old_old", "old_new", "new_old", "new_new"]
FILTERS = {"old_users": (True, None), "new_users": (False, None), "old_old": (True, True), "old_new": (True, False),
           "new_old": (False, True), "new_new": (False, False)}


def main():
    topk = [1, 5, 10]
    metrics = ["Recall", "Hit", "NDCG", "MRR"]
    cfg = FakeConfig(metrics=metrics, topk=topk, device="cpu", eval_args={"mode": "uni50"}, metric_decimal_place=4,
                     USER_ID_FIELD="user_id", ITEM_ID_FIELD="item_id", eval_type=None, model_eval_type="retrieval")
    n_old_users, n_old_items, tot_items, n_neg = 1000, 300, 500, 50
    collectors = {"overall": Collector(cfg)}
    for name, (ru, ri) in FILTERS.items():
        collectors[name] = FilteredCollector(cfg, FastUserItemCollectorFilter(n_old_users, n_old_items, ru, ri), name)
    rng = np.random.default_rng(17)
    torch.manual_seed(17)  # the collectors' randperm
    out = {"n_old_users": np.array(n_old_users), "n_old_items": np.array(n_old_items), "tot_items": np.array(tot_items),
           "topk": np.array(topk)}
    finite = {n: [] for n in NAMES}
    for b, n_users in enumerate((9, 14, 3)):
        uid_of = rng.choice(np.arange(1, 2000), size=n_users, replace=False)  # old (< 1000) and new users mixed
        rows_u, row_uid, cols, pos_u, pos_i = [], [], [], [], []
        for u in range(n_users):
            n_pos = int(rng.integers(1, 5))
            cand = rng.choice(np.arange(1, tot_items), size=n_pos * (1 + n_neg), replace=False)
            rows_u += [u] * len(cand)
            row_uid += [int(uid_of[u])] * len(cand)
            cols += cand.tolist()
            pos_u += [u] * n_pos
            pos_i += cand[:n_pos].tolist()
        M = len(cols)
        scores = rng.permutation(M).astype(np.float32) / 7.0 - 3.0
        is_pos = np.zeros(M, bool)
        off = 0
        for u in range(n_users):
            is_pos[off:off + pos_u.count(u)] = True
            off += rows_u.count(u)
        lift = is_pos & (rng.random(M) < 0.6)
        scores[lift] += np.float32(M / 7.0 + 1.0) + rng.permutation(M).astype(np.float32)[lift] / 16.0
        assert len(np.unique(scores)) == M
        row_idx, col_idx = torch.tensor(rows_u), torch.tensor(cols)
        dense = torch.full((n_users, tot_items), -np.inf)
        dense[row_idx, col_idx] = torch.from_numpy(scores)
        inter = {"user_id": torch.tensor(row_uid), "item_id": col_idx}
        for name in NAMES:  # the SAME tensor to every collector, as evaluate_model does
            before = {n: (None if c.data_struct._data_dict.get("rec.topk") is None else c.data_struct.get("rec.topk").shape[0])
                      for n, c in collectors.items()}
            collectors[name].eval_batch_collect(dense, inter, torch.tensor(pos_u), torch.tensor(pos_i))
            after = collectors[name].data_struct._data_dict.get("rec.topk")
            n_new = 0 if after is None else after.shape[0] - (before[name] or 0)
            # how many leading top-k entries of each appended row were finite when this collector looked at the matrix
            if n_new:
                f = collectors[name].filter if name != "overall" else None
                rows = torch.arange(n_users) if f is None else f.last_users.unique(sorted=True)
                fin = torch.isfinite(dense[rows]).sum(1).clamp(max=max(topk))
                finite[name].append(fin.numpy())
        for k_, v in (("row_idx", rows_u), ("row_uid", row_uid), ("col_idx", cols), ("scores", scores), ("pos_u", pos_u),
                      ("pos_i", pos_i), ("uid_of", uid_of)):
            out[f"b{b}_{k_}"] = np.asarray(v)
    out["n_batches"] = np.array(3)
    for name, c in collectors.items():
        struct = c.get_data_struct()
        rec = struct.get("rec.topk").numpy()
        res = Evaluator(cfg).evaluate(struct)
        out[name + "_rec_topk"] = rec
        out[name + "_finite_k"] = np.concatenate(finite[name]) if finite[name] else np.zeros((0,), np.int64)
        out[name + "_metric_values"] = np.array([float(v) for v in res.values()])
        out["metric_names"] = np.array(list(res.keys()))
        print(name, rec.shape, "rows fully determined:", int((out[name + "_finite_k"] >= max(topk)).sum()), dict(list(res.items())[:3]))
    np.savez_compressed(os.path.join(HERE, "eval_filtered.npz"), **out)
    print("wrote eval_filtered.npz")


if __name__ == "__main__":
    main()
