#!/usr/bin/env python3
"""End-to-end golden for BASELINE configs 1 and 2 on the bundled ml-100k, produced by the REAL
reference (run by hand in the build container, like make_golden.py):

    python tests/golden/make_golden_ml100k.py

* loads ml-100k through the reference's own Config / create_dataset (user + item feature columns);
* declares the first 80 % of the remapped user/item ids "in vocabulary" and the rest OOV, so that
  OOV users/items are embedded by the plugin from their features;
* trains the reference's BPR (its own calculate_loss / forward, OOV rows through the reference's
  LSHInductiveEmbedder) for a few epochs of plain Adam on CPU;
* evaluates with the reference's ind_full_sort_predict over ALL items (old + new), masks the padding
  item and the training positives like the reference trainer does (trainer.py:541-544), takes
  torch.topk(10) and computes Recall@10 = hits / #positives per user (R/evaluator/metrics.py:159-160);
* repeats the evaluation with the reference's MeanEmbedder on the same tables (config 1).

Output: tests/golden/ml100k_bpr.npz (weights, planes, feature matrices, eval users, CSR of train
and test positives, reference top-10 and Recall@10).  ml-1m (config 2) is not in the container, so
ml-100k stands in for it.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()

import torch  # noqa: E402
from recbole.config import Config  # noqa: E402
from recbole.data import create_dataset  # noqa: E402
from recbole.data.interaction import Interaction  # noqa: E402
from recbole.inductive.feature_cache import InductiveFeatureCache  # noqa: E402
from recbole.inductive.lsh_embedder import LSHInductiveEmbedder  # noqa: E402
from recbole.inductive.mean_embedder import MeanEmbedder  # noqa: E402
from recbole.model.general_recommender.bpr import BPR  # noqa: E402

from make_golden import FakeConfig, FakeDataset, np_  # noqa: E402

PRIME_PAD = 112062759511
D, H, K = 64, 8, 10


def csr(lists):
    ptr = np.zeros(len(lists) + 1, np.int64)
    for i, l in enumerate(lists):
        ptr[i + 1] = ptr[i] + len(l)
    return ptr, np.array([x for l in lists for x in l], np.int64)


def evaluate(model, users, train_pos, test_pos, tot_items):
    with torch.no_grad():
        scores = model.ind_full_sort_predict(Interaction({"user_id": users}), torch.arange(tot_items))
    scores = scores.view(len(users), tot_items).clone()
    scores[:, 0] = -np.inf  # padding item (trainer.py:541-544)
    for r, u in enumerate(users.tolist()):
        scores[r, train_pos[u]] = -np.inf
    _, top = torch.topk(scores, K, dim=-1)
    rec = np.zeros(len(users))
    for r, u in enumerate(users.tolist()):
        pos = set(test_pos[u])
        rec[r] = len(pos & set(top[r].tolist())) / len(pos)
    return scores, top, rec


def main():
    torch.manual_seed(2020)
    np.random.seed(2020)
    cfg = Config(model="BPR", dataset="ml-100k", config_dict={
        "data_path": "/root/reference/RecBole/dataset/", "seed": 2020, "use_gpu": False,
        "load_col": {"inter": ["user_id", "item_id", "rating", "timestamp"],
                     "user": ["user_id", "age", "gender", "occupation", "zip_code"],
                     "item": ["item_id", "movie_title", "release_year", "class"]},
        "inductive_embedder": "lsh", "add_oov_buckets": True, "user_oov_buckets": H, "item_oov_buckets": H,
        "embedding_size": D})
    ds = create_dataset(cfg)
    ds._change_feat_format()  # DataFrame -> Interaction, what Dataset.build() does before handing features out
    uf, itf = ds.get_user_feature(), ds.get_item_feature()
    tot_users, tot_items = ds.user_num, ds.item_num
    n_users, n_items = int(tot_users * 0.8), int(tot_items * 0.8)
    inter_u, inter_i = np_(ds.inter_feat["user_id"]), np_(ds.inter_feat["item_id"])
    rng = np.random.default_rng(2020)
    is_test = rng.random(len(inter_u)) < 0.1
    train_pos = [[] for _ in range(tot_users)]
    test_pos = [[] for _ in range(tot_users)]
    for u, i, t in zip(inter_u.tolist(), inter_i.tolist(), is_test.tolist()):
        (test_pos if t else train_pos)[u].append(i)

    torch.manual_seed(7)
    lsh = LSHInductiveEmbedder(uf, itf, n_users, n_items, H, H, D, "cpu", PRIME_PAD, "per-feature",
                               InductiveFeatureCache())
    mcfg = FakeConfig(USER_ID_FIELD="user_id", ITEM_ID_FIELD="item_id", NEG_PREFIX="neg_", device="cpu",
                      embedding_size=D, add_oov_buckets=True, user_oov_buckets=H, item_oov_buckets=H,
                      oov_freeze_embedding=False)
    model = BPR(mcfg, FakeDataset(n_users, n_items), None, lsh)
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    tu, ti = torch.from_numpy(inter_u[~is_test]), torch.from_numpy(inter_i[~is_test])
    g = torch.Generator().manual_seed(11)
    for epoch in range(4):
        perm = torch.randperm(len(tu), generator=g)
        tot = 0.0
        for lo in range(0, len(perm), 2048):
            idx = perm[lo:lo + 2048]
            neg = torch.randint(1, tot_items, (len(idx),), generator=g)
            loss = model.calculate_loss(Interaction({"user_id": tu[idx], "item_id": ti[idx], "neg_item_id": neg}))
            if not torch.isfinite(loss):  # all-zero-code NaN rows poison the batch mean: drop them like NaN-guard
                continue
            opt.zero_grad()
            loss.backward()
            for p in model.parameters():
                if p.grad is not None:
                    torch.nan_to_num_(p.grad, nan=0.0)
            opt.step()
            tot += float(loss)
        print(f"epoch {epoch} loss {tot:.3f}")
    model.eval()
    users = torch.tensor([u for u in range(1, tot_users) if test_pos[u]], dtype=torch.int64)
    scores, top, rec = evaluate(model, users, train_pos, test_pos, tot_items)
    nan_items = int(torch.isnan(scores[0]).sum())
    print(f"lsh: users {len(users)} (OOV {(users >= n_users).sum()}), NaN items {nan_items}, recall@10 {rec.mean():.6f}")

    mean = MeanEmbedder(uf, itf, n_users, n_items, H, H, D, "cpu")
    model_mean = BPR(mcfg, FakeDataset(n_users, n_items), None, mean)
    model_mean.load_state_dict({k: v for k, v in model.state_dict().items() if not k.startswith("inductive_embedder")},
                               strict=False)
    model_mean.eval()
    _, top_m, rec_m = evaluate(model_mean, users, train_pos, test_pos, tot_items)
    print(f"mean: recall@10 {rec_m.mean():.6f}")

    tr_ptr, tr_idx = csr(train_pos)
    te_ptr, te_idx = csr(test_pos)
    sd = model.state_dict()
    np.savez_compressed(
        os.path.join(HERE, "ml100k_bpr.npz"),
        n_users=np.array(n_users), n_items=np.array(n_items), tot_users=np.array(tot_users),
        tot_items=np.array(tot_items),
        user_feat=np_(lsh.user_feature_mat), item_feat=np_(lsh.item_feature_mat),
        user_planes=np_(sd["inductive_embedder.user_lsh.uniform_planes.0"]),
        item_planes=np_(sd["inductive_embedder.item_lsh.uniform_planes.0"]),
        user_buckets=np_(sd["user_oov_buckets.weight"]), item_buckets=np_(sd["item_oov_buckets.weight"]),
        user_table=np_(sd["user_embedding.weight"]), item_table=np_(sd["item_embedding.weight"]),
        users=np_(users), train_ptr=tr_ptr, train_idx=tr_idx, test_ptr=te_ptr, test_idx=te_idx,
        lsh_top10=np_(top), lsh_recall=rec, lsh_recall_mean=np.array(rec.mean()),
        lsh_scores_sample=np_(scores[:8]),
        mean_top10=np_(top_m), mean_recall=rec_m, mean_recall_mean=np.array(rec_m.mean()))
    print("wrote ml100k_bpr.npz")


if __name__ == "__main__":
    main()
