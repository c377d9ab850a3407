#!/usr/bin/env python3
"""Evaluation fixtures: the REAL reference's Collector + Evaluator on uni-N sampled batches.

    python tests/golden/make_golden_eval.py        (build container only; needs /root/reference)

Drives recbole.evaluator.collector.Collector.eval_batch_collect exactly as
InductiveEvaluator.neg_sample_batch_eval does (R/inductive/evaluator.py:118-134): the batch's sampled scores are
scattered into a dense [users, items] matrix of -inf, the collector takes torch.topk and the positive matrix, and
recbole.evaluator.Evaluator turns the collected "rec.topk" block into Recall/Hit/Precision/NDCG/MRR/MAP@k.
Writes eval_uni.npz: the sparse inputs, the rec.topk block and the metric values.  Scores are distinct within a
user and candidate columns are distinct within a user (the reference's scatter keeps an arbitrary duplicate).
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


class FakeConfig(dict):
    def __getitem__(self, k):
        return self.get(k, None)


def main():
    topk = [1, 5, 10, 20]
    metrics = ["Recall", "Hit", "Precision", "NDCG", "MRR", "MAP"]
    cfg = FakeConfig(metrics=metrics, topk=topk, device="cpu", eval_args={"mode": "uni50"}, metric_decimal_place=4,
                     USER_ID_FIELD="user_id", ITEM_ID_FIELD="item_id", eval_type=None)
    collector, evaluator = Collector(cfg), Evaluator(cfg)
    rng = np.random.default_rng(11)
    tot_items, n_neg = 500, 50
    out = {}
    blocks = []
    for b, n_users in enumerate((7, 12, 1)):
        rows_u, cols, pos_u, pos_i = [], [], [], []
        for u in range(n_users):
            n_pos = int(rng.integers(1, 6)) if not (b == 1 and u == 3) else 30   # one user with 30 positives
            cand = rng.choice(np.arange(1, tot_items), size=min(tot_items - 1, n_pos * (1 + n_neg)), replace=False)
            rows_u += [u] * len(cand)
            cols += cand.tolist()                      # positives first, then the sampled negatives
            pos_u += [u] * n_pos
            pos_i += cand[:n_pos].tolist()
        M = len(cols)
        scores = rng.permutation(M).astype(np.float32) / 7.0 - 3.0        # distinct
        is_pos = np.zeros(M, bool)
        off = 0
        for u in range(n_users):
            n_u = rows_u.count(u)
            is_pos[off:off + pos_u.count(u)] = True
            off += n_u
        lift = is_pos & (rng.random(M) < 0.6)                               # most positives rank high
        scores[lift] += np.float32(M / 7.0 + 1.0) + rng.permutation(M).astype(np.float32)[lift] / 16.0
        assert len(np.unique(scores)) == M
        if b == 0:
            scores[5] = np.nan                                              # NaN ranks first in torch.topk
        row_idx, col_idx = torch.tensor(rows_u), torch.tensor(cols)
        dense = torch.full((n_users, tot_items), -np.inf)
        dense[row_idx, col_idx] = torch.from_numpy(scores)
        collector.eval_batch_collect(dense, None, torch.tensor(pos_u), torch.tensor(pos_i))
        for k_, v in (("row_idx", rows_u), ("col_idx", cols), ("scores", scores), ("pos_u", pos_u), ("pos_i", pos_i)):
            out[f"b{b}_{k_}"] = np.asarray(v)
        blocks.append(n_users)
    struct = collector.get_data_struct()
    rec = struct.get("rec.topk").numpy()
    result = evaluator.evaluate(struct)
    out["rec_topk"] = rec
    out["topk"] = np.array(topk)
    out["metric_names"] = np.array(list(result.keys()))
    out["metric_values"] = np.array([float(v) for v in result.values()])
    out["tot_items"] = np.array(tot_items)
    out["n_batches"] = np.array(len(blocks))
    np.savez_compressed(os.path.join(HERE, "eval_uni.npz"), **out)
    print("eval_uni.npz rec.topk", rec.shape, dict(result))
    full_sort_fixture(cfg)


def full_sort_fixture(cfg):
    """InductiveEvaluator.eval_batch (R/inductive/evaluator.py:70-96) on a full-sort batch: dense scores = U @ E.T
    (what BPR.full_sort_predict returns), column 0 and the users' history set to -inf, then the collector."""
    cfg = FakeConfig({**cfg, "eval_args": {"mode": "full"}})
    collector, evaluator = Collector(cfg), Evaluator(cfg)
    g = torch.Generator().manual_seed(21)
    n_users, n_items, D = 37, 800, 64
    U, E = torch.randn((n_users, D), generator=g), torch.randn((n_items, D), generator=g)
    rng = np.random.default_rng(22)
    hist_u, hist_i, pos_u, pos_i = [], [], [], []
    for u in range(n_users):
        items = rng.choice(np.arange(1, n_items), size=int(rng.integers(3, 60)), replace=False)
        n_pos = int(rng.integers(1, 5))
        pos_u += [u] * n_pos
        pos_i += items[:n_pos].tolist()
        hist_u += [u] * (len(items) - n_pos)
        hist_i += items[n_pos:].tolist()
    U[torch.tensor(pos_u)] += 0.35 * E[torch.tensor(pos_i)]   # positives tend to rank high
    scores = (U @ E.T).view(-1, n_items)
    scores[:, 0] = -np.inf
    scores[torch.tensor(hist_u), torch.tensor(hist_i)] = -np.inf
    top2 = torch.topk(scores, 21, dim=1).values
    gap = float(((top2[:, :-1] - top2[:, 1:]) / top2[:, :-1].abs()).min())
    collector.eval_batch_collect(scores, None, torch.tensor(pos_u), torch.tensor(pos_i))
    struct = collector.get_data_struct()
    result = evaluator.evaluate(struct)
    np.savez_compressed(os.path.join(HERE, "eval_full.npz"), U=U.numpy(), E=E.numpy(), hist_u=np.array(hist_u),
                        hist_i=np.array(hist_i), pos_u=np.array(pos_u), pos_i=np.array(pos_i),
                        rec_topk=struct.get("rec.topk").numpy(), topk=np.array(cfg["topk"]),
                        metric_names=np.array(list(result.keys())),
                        metric_values=np.array([float(v) for v in result.values()]), min_rel_gap=np.array(gap))
    print("eval_full.npz min relative gap inside the top 21:", gap, {k: float(v) for k, v in list(result.items())[:4]})


if __name__ == "__main__":
    main()
