#!/usr/bin/env python3
"""Gradient golden for the context-model OOV splice, produced by the REAL reference's autograd:

    python tests/golden/make_golden_context_grad.py

The reference's DCNV2 (make_golden_context.build) embeds one batch of token fields through
InductiveContextRecommender.embed_token_fields (R/model/abstract_recommender.py:794-842) and
InductiveFMFirstOrderLinear.embed_token_fields (R/model/layers.py:1634-1693) with gradients enabled; the loss is a
fixed random weighting of both outputs.  Recorded: the inputs, the outputs and d loss / d (fused token table, OOV bucket
tables) for the main and the first-order model, for an lsh embedder and for the random mapper.
Output: tests/golden/context_grad.npz.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()

import torch  # noqa: E402

from make_golden import np_  # noqa: E402
from make_golden_context import build  # noqa: E402


def main():
    out = {}
    for kind in ("lsh", "mapper"):
        cfg, ds, m = build(kind)
        tot_users, tot_items = m.n_users, m.n_items
        n_users, n_items = int(tot_users * 0.8), int(tot_items * 0.8)
        m.n_users, m.n_items = n_users, n_items
        fo = m.first_order_linear
        fo.n_users, fo.n_items = n_users, n_items
        if kind == "mapper":
            for mp in (m.inductive_mapper, fo.inductive_mapper):
                mp.n_original_users, mp.n_original_items = n_users, n_items
        dims = list(m.token_field_dims)
        g = torch.Generator().manual_seed(21)
        B = 401
        cols = [torch.randint(1, tot_users, (B,), generator=g), torch.randint(1, tot_items, (B,), generator=g)]
        cols += [torch.randint(0, d, (B,), generator=g) for d in dims[2:]]
        tokens = torch.stack(cols, dim=1)
        m.zero_grad()
        second = m.embed_token_fields(tokens.clone())
        first = fo.embed_token_fields(tokens.clone(), 0, 1)
        w2 = torch.randn(second.shape, generator=g)
        w1 = torch.randn(first.shape, generator=g)
        loss = (second * w2).sum() + (first * w1).sum()
        loss.backward()
        p = kind + "_"
        out.update({
            p + "tokens": np_(tokens), p + "offsets": np.asarray(m.token_embedding_table.offsets, dtype=np.int64),
            p + "table": np_(m.token_embedding_table.embedding.weight), p + "fo_table": np_(fo.token_embedding_table.embedding.weight),
            p + "second": np_(second), p + "first": np_(first), p + "w2": np_(w2), p + "w1": np_(w1), p + "loss": np_(loss),
            p + "user_buckets": np_(m.user_oov_buckets.weight), p + "item_buckets": np_(m.item_oov_buckets.weight),
            p + "fo_user_buckets": np_(fo.user_oov_buckets.weight), p + "fo_item_buckets": np_(fo.item_oov_buckets.weight),
            p + "g_table": np_(m.token_embedding_table.embedding.weight.grad),
            p + "g_fo_table": np_(fo.token_embedding_table.embedding.weight.grad),
            p + "g_user_buckets": np_(m.user_oov_buckets.weight.grad), p + "g_item_buckets": np_(m.item_oov_buckets.weight.grad),
            p + "g_fo_user_buckets": np_(fo.user_oov_buckets.weight.grad),
            p + "g_fo_item_buckets": np_(fo.item_oov_buckets.weight.grad),
            p + "n_users": np.array(n_users), p + "n_items": np.array(n_items)})
        if kind == "lsh":
            e, fe = m.inductive_embedder, fo.inductive_embedder
            out.update({"lsh_user_feat": np_(e.user_feature_mat), "lsh_item_feat": np_(e.item_feature_mat),
                        "lsh_user_planes": np_(e.user_lsh.uniform_planes[0].data),
                        "lsh_item_planes": np_(e.item_lsh.uniform_planes[0].data),
                        "lsh_fo_user_planes": np_(fe.user_lsh.uniform_planes[0].data),
                        "lsh_fo_item_planes": np_(fe.item_lsh.uniform_planes[0].data)})
        print(kind, "loss", float(loss), "nan in grads", any(bool(torch.isnan(t.grad).any()) for t in
              (m.token_embedding_table.embedding.weight, m.user_oov_buckets.weight, m.item_oov_buckets.weight)))
    np.savez_compressed(os.path.join(HERE, "context_grad.npz"), **out)
    print("wrote context_grad.npz")


if __name__ == "__main__":
    main()
