"""BASELINE configs 1 and 2 on the bundled ml-100k (ml-1m is not in the container): Recall@10 of the
build against the reference's own evaluation of the same trained weights
(tests/golden/make_golden_ml100k.py ran the REAL reference: its loader, its BPR, its lsh/mean
embedders, torch.topk and the Recall definition of R/evaluator/metrics.py:159-160).
north_star bar: Recall@10 within 1e-4 of the reference."""
import numpy as np
import pytest
import torch

K = 10
PRIME_PAD = 112062759511


def recall_at_k(top, users, te_ptr, te_idx):
    rec = np.zeros(len(users))
    for r, u in enumerate(users.tolist()):
        pos = set(te_idx[te_ptr[u]:te_ptr[u + 1]].tolist())
        rec[r] = len(pos & set(top[r].tolist())) / len(pos)
    return rec


def mask_scores(scores, users, tr_ptr, tr_idx):
    scores[:, 0] = -np.inf  # padding item, as R/trainer/trainer.py:541-544
    for r, u in enumerate(users.tolist()):
        scores[r, tr_idx[tr_ptr[u]:tr_ptr[u + 1]]] = -np.inf
    return scores


def test_oracle_recall_matches_reference(golden, oracle):
    z = golden("ml100k_bpr.npz")
    users, tot_items = z["users"], int(z["tot_items"])
    ue = oracle.lsh_lookup(users, z["user_table"], z["user_feat"], z["user_planes"], z["user_buckets"])
    ie = oracle.lsh_lookup(np.arange(tot_items), z["item_table"], z["item_feat"], z["item_planes"], z["item_buckets"])
    scores = mask_scores(oracle.full_sort_scores(ue, ie), users, z["train_ptr"], z["train_idx"])
    assert np.allclose(scores[:8], z["lsh_scores_sample"], rtol=1e-5, atol=1e-6)
    top = np.argsort(-scores, axis=1, kind="stable")[:, :K]
    rec = recall_at_k(top, users, z["test_ptr"], z["test_idx"])
    assert abs(rec.mean() - float(z["lsh_recall_mean"])) <= 1e-4
    assert (np.sort(top, 1) == np.sort(z["lsh_top10"], 1)).all(1).mean() > 0.99  # same top-10 sets (fp ties aside)
    # config 1: mean embedder -- OOV rows are the column mean of the in-vocabulary table
    n_users, n_items = int(z["n_users"]), int(z["n_items"])
    um, im = oracle.col_mean(z["user_table"]), oracle.col_mean(z["item_table"])
    ue = np.where((users < n_users)[:, None], z["user_table"][np.minimum(users, n_users - 1)], um[None, :])
    ie = np.concatenate([z["item_table"], np.tile(im, (tot_items - n_items, 1))]).astype(np.float32)
    scores = mask_scores(oracle.full_sort_scores(ue.astype(np.float32), ie), users, z["train_ptr"], z["train_idx"])
    top = np.argsort(-scores, axis=1, kind="stable")[:, :K]
    rec = recall_at_k(top, users, z["test_ptr"], z["test_idx"])
    assert abs(rec.mean() - float(z["mean_recall_mean"])) <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("embedder", ["lsh", "mean"])
def test_gpu_recall_matches_reference(embedder, golden, dev):
    import mi_oov

    class Cfg(dict):
        def __getitem__(self, k):
            return self.get(k, None)

    class DS:
        def __init__(self, a, b):
            self.n = {"user_id": a, "item_id": b}

        def num(self, f):
            return self.n[f]

    z = golden("ml100k_bpr.npz")
    T = lambda k: torch.from_numpy(z[k]).to(dev)  # noqa: E731
    n_users, n_items, tot_items = int(z["n_users"]), int(z["n_items"]), int(z["tot_items"])
    cfg = Cfg(USER_ID_FIELD="user_id", ITEM_ID_FIELD="item_id", NEG_PREFIX="neg_", device=dev, embedding_size=64,
              add_oov_buckets=True, user_oov_buckets=8, item_oov_buckets=8, oov_freeze_embedding=False)
    ft_u = mi_oov.FeatureTable({"id": torch.arange(z["user_feat"].shape[0]), "f": torch.from_numpy(z["user_feat"])})
    ft_i = mi_oov.FeatureTable({"id": torch.arange(z["item_feat"].shape[0]), "f": torch.from_numpy(z["item_feat"])})
    if embedder == "lsh":
        emb = mi_oov.LSHInductiveEmbedder(ft_u, ft_i, n_users, n_items, 8, 8, 64, dev, PRIME_PAD, "none",
                                          mi_oov.InductiveFeatureCache())
    else:
        emb = mi_oov.MeanEmbedder(ft_u, ft_i, n_users, n_items, 8, 8, 64, dev)
    bpr = mi_oov.BPR(cfg, DS(n_users, n_items), None, emb).to(dev)
    sd = {"user_oov_buckets.weight": T("user_buckets"), "item_oov_buckets.weight": T("item_buckets"),
          "user_embedding.weight": T("user_table"), "item_embedding.weight": T("item_table")}
    if embedder == "lsh":
        sd["inductive_embedder.user_lsh.uniform_planes.0"] = T("user_planes")
        sd["inductive_embedder.item_lsh.uniform_planes.0"] = T("item_planes")
    bpr.load_state_dict(sd)
    users = T("users")
    with torch.no_grad():
        scores = bpr.ind_full_sort_predict({"user_id": users}, torch.arange(tot_items, device=dev))
    scores = scores.view(len(users), tot_items).cpu().numpy()
    scores = mask_scores(scores, z["users"], z["train_ptr"], z["train_idx"])
    _, top = torch.topk(torch.from_numpy(scores), K, dim=-1)
    rec = recall_at_k(top.numpy(), z["users"], z["test_ptr"], z["test_idx"])
    assert abs(rec.mean() - float(z[embedder + "_recall_mean"])) <= 1e-4
    assert (np.sort(top.numpy(), 1) == np.sort(z[embedder + "_top10"], 1)).all(1).mean() > 0.99
