#!/usr/bin/env python3
"""Golden for the context-model OOV splice, produced by the REAL reference:

    python tests/golden/make_golden_context.py

Builds the reference's DCNV2 (an InductiveContextRecommender) on the bundled ml-100k, shrinks its
vocabulary to the first 80 % of the user/item ids so that the rest are OOV, and records
  * InductiveContextRecommender.embed_token_fields   (R/model/abstract_recommender.py:794-842)
  * InductiveFMFirstOrderLinear.embed_token_fields   (R/model/layers.py:1634-1693)
for (a) an lsh embedder (main, D=16, and the first-order one, D=1, that shares its feature matrices)
and (b) the random mapper with OOV bucket tables.  Output: tests/golden/context_splice.npz.

BASELINE config 5 (DCNV2 + knn): `knn` and `mean` embedders reading their rows out of the fused token table
(knn_embedder.py:117-123,135-144, mean_embedder.py:53-60,75-86), on the >= 3-field ml-100k model (item slice
offsets[1]:offsets[2]) and, for `mean`, on a 2-field model (user_id, item_id only: item slice offsets[1]:).
ScaNN is absent: the neighbour search is the exact stand-in of ref_shims (indices unpinned, recorded here);
what the reference does with them is pinned.  Output: tests/golden/context_knn_mean.npz.
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
from recbole.inductive.get_inductive import get_inductive_embedder, get_inductive_mapper  # noqa: E402
from recbole.model.context_aware_recommender.dcnv2 import DCNV2  # noqa: E402

from make_golden import np_  # noqa: E402

D, H = 16, 8


def build(kind, two_field=False):
    load_col = {"inter": ["user_id", "item_id", "rating", "timestamp"]}
    if not two_field:
        load_col.update({"user": ["user_id", "age", "gender", "occupation", "zip_code"],
                         "item": ["item_id", "movie_title", "release_year", "class"]})
    cfg = Config(model="DCNV2", dataset="ml-100k", config_dict={
        "data_path": "/root/reference/RecBole/dataset/", "seed": 2020, "use_gpu": False,
        "load_col": load_col, "n_neighbors": 2,
        "inductive_embedder": kind if kind in ("lsh", "knn", "mean") else None,
        "inductive_mapper": "random" if kind == "mapper" else None,
        "add_oov_buckets": True, "user_oov_buckets": H, "item_oov_buckets": H, "embedding_size": D,
        "oov_normalization_type": "per-feature", "threshold": {"rating": 3}})
    ds = create_dataset(cfg)
    ds._change_feat_format()
    torch.manual_seed(5)
    emb = get_inductive_embedder(cfg, ds)
    mp = get_inductive_mapper(cfg, ds)
    torch.manual_seed(6)
    m = DCNV2(cfg, ds, mp, emb)
    return cfg, ds, m


def main():
    out = {}
    for kind in ("lsh", "mapper"):
        cfg, ds, m = build(kind)
        tot_users, tot_items = m.n_users, m.n_items
        n_users, n_items = int(tot_users * 0.8), int(tot_items * 0.8)
        m.n_users, m.n_items = n_users, n_items  # ids beyond are OOV for the splice
        fo = m.first_order_linear
        fo.n_users, fo.n_items = n_users, n_items
        if kind == "mapper":  # the mappers were built with the full vocabulary: re-point them too
            for mp in (m.inductive_mapper, fo.inductive_mapper):
                mp.n_original_users, mp.n_original_items = n_users, n_items
        dims = list(m.token_field_dims)
        g = torch.Generator().manual_seed(9)
        B = 777
        cols = [torch.randint(1, tot_users, (B,), generator=g), torch.randint(1, tot_items, (B,), generator=g)]
        cols += [torch.randint(0, d, (B,), generator=g) for d in dims[2:]]
        tokens = torch.stack(cols, dim=1)
        with torch.no_grad():
            second = m.embed_token_fields(tokens.clone())
            first = fo.embed_token_fields(tokens.clone(), 0, 1)
        p = kind + "_"
        out.update({
            p + "tokens": np_(tokens), p + "offsets": np.asarray(m.token_embedding_table.offsets, dtype=np.int64),
            p + "table": np_(m.token_embedding_table.embedding.weight),
            p + "fo_table": np_(fo.token_embedding_table.embedding.weight),
            p + "second": np_(second), p + "first": np_(first),
            p + "user_buckets": np_(m.user_oov_buckets.weight), p + "item_buckets": np_(m.item_oov_buckets.weight),
            p + "fo_user_buckets": np_(fo.user_oov_buckets.weight), p + "fo_item_buckets": np_(fo.item_oov_buckets.weight),
            p + "n_users": np.array(n_users), p + "n_items": np.array(n_items)})
        if kind == "lsh":
            e, fe = m.inductive_embedder, fo.inductive_embedder
            assert fe.user_feature_mat is e.user_feature_mat  # shared (abstract_recommender.py:751-753)
            out.update({"lsh_user_feat": np_(e.user_feature_mat), "lsh_item_feat": np_(e.item_feature_mat),
                        "lsh_user_planes": np_(e.user_lsh.uniform_planes[0].data),
                        "lsh_item_planes": np_(e.item_lsh.uniform_planes[0].data),
                        "lsh_fo_user_planes": np_(fe.user_lsh.uniform_planes[0].data),
                        "lsh_fo_item_planes": np_(fe.item_lsh.uniform_planes[0].data)})
        n_oov = int((tokens[:, 0] >= n_users).sum()), int((tokens[:, 1] >= n_items).sum())
        print(kind, "fields", m.token_field_names, "B", B, "oov users/items", n_oov, "second", tuple(second.shape),
              "first", tuple(first.shape), "nan rows", int(torch.isnan(second).any(2).any(1).sum()))
    np.savez_compressed(os.path.join(HERE, "context_splice.npz"), **out)
    print("wrote context_splice.npz")
    knn_mean()


def knn_mean():
    out = {}
    for kind, two_field in (("knn", False), ("mean", False), ("mean", True)):
        cfg, ds, m = build(kind, two_field)
        tot_users, tot_items = m.n_users, m.n_items
        n_users, n_items = int(tot_users * 0.8), int(tot_items * 0.8)
        if kind == "knn":  # the searchers index feature rows [:n_original]: rebuild the embedders for the shrunk vocabulary
            from recbole.inductive.knn_embedder import KNNInductiveEmbedder
            for holder, width in ((m, D), (m.first_order_linear, 1)):
                e = holder.inductive_embedder
                holder.inductive_embedder = KNNInductiveEmbedder(e.user_features, e.item_features, n_users, n_items, H, H,
                                                                 width, "cpu", e.prime_pad, n_neighbors=2)
        m.n_users, m.n_items = n_users, n_items
        fo = m.first_order_linear
        fo.n_users, fo.n_items = n_users, n_items
        dims = list(m.token_field_dims)
        assert (len(dims) == 2) == two_field
        g = torch.Generator().manual_seed(11)
        B = 613
        cols = [torch.randint(1, tot_users, (B,), generator=g), torch.randint(1, tot_items, (B,), generator=g)]
        cols += [torch.randint(0, d, (B,), generator=g) for d in dims[2:]]
        tokens = torch.stack(cols, dim=1)
        with torch.no_grad():
            second = m.embed_token_fields(tokens.clone())
            first = fo.embed_token_fields(tokens.clone(), 0, 1)
        p = kind + ("2_" if two_field else "_")
        out.update({
            p + "tokens": np_(tokens), p + "offsets": np.asarray(m.token_embedding_table.offsets, dtype=np.int64),
            p + "table": np_(m.token_embedding_table.embedding.weight),
            p + "fo_table": np_(fo.token_embedding_table.embedding.weight),
            p + "second": np_(second), p + "first": np_(first),
            p + "n_users": np.array(n_users), p + "n_items": np.array(n_items)})
        if kind == "knn":
            e = m.inductive_embedder
            u_oov = tokens[:, 0][tokens[:, 0] >= n_users]
            i_oov = tokens[:, 1][tokens[:, 1] >= n_items]
            out.update({"knn_user_feat": np.asarray(e.user_feature_mat), "knn_item_feat": np.asarray(e.item_feature_mat),
                        "knn_user_idx": np_(e._hash_users(u_oov)), "knn_item_idx": np_(e._hash_items(i_oov))})
        print(kind, "two_field" if two_field else "", "fields", m.token_field_names, "offsets", list(m.token_embedding_table.offsets),
              "second", tuple(second.shape), "first", tuple(first.shape))
    np.savez_compressed(os.path.join(HERE, "context_knn_mean.npz"), **out)
    print("wrote context_knn_mean.npz")


if __name__ == "__main__":
    main()
