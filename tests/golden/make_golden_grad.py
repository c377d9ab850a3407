#!/usr/bin/env python3
"""Gradient fixtures: the REAL reference's autograd through the OOV training step.

    python tests/golden/make_golden_grad.py        (build container only; needs /root/reference)

Writes bpr_grad.npz: for each plugin variant (lsh / slsh with 8 buckets / slsh with 200 buckets / knn /
mapper) a reference BPR in train mode, one `calculate_loss` over a batch that mixes in-vocabulary and
prime-padded OOV ids (bpr.py:127-143, lsh_embedder.py:153-155), and the gradient of every table.
Ids whose lsh code is all-zero are excluded from the batch: the reference's loss is NaN for them and
its trainer aborts (trainer.py `_check_nan`), so no gradient exists to compare.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (installs the shims, imports the reference)

import torch  # noqa: E402
from recbole.data.interaction import Interaction  # noqa: E402

PRIME_PAD = mg.PRIME_PAD


def one_case(tag, build, n_users, n_items, n_new_u, n_new_i, D, n_ub, n_ib, seed, out, post=None):
    uf = mg.features(n_new_u, [("a", 1, "float"), ("v", 9, "float")], seed, "user_id")
    itf = mg.features(n_new_i, [("y", 1, "float"), ("w", 20, "float")], seed + 1, "item_id")
    torch.manual_seed(seed + 2)
    mapper, emb = build(uf, itf)
    model = mg.make_bpr(n_users, n_items, D, mapper, emb, n_ub, n_ib, seed + 3)
    if post is not None:
        post(model)
    model.train()
    if emb is not None and hasattr(emb, "set_train"):
        emb.set_train()
    g = torch.Generator().manual_seed(seed + 4)
    B = 300
    users = torch.randint(1, n_new_u, (B,), generator=g)
    pos = torch.randint(1, n_new_i, (B,), generator=g)
    neg = torch.randint(1, n_new_i, (B,), generator=g)
    if tag == "lsh":  # drop ids with an all-zero code (NaN rows)
        with torch.no_grad():
            ok_u = emb._hash_users(torch.arange(n_new_u)).sum(1) > 0
            ok_i = emb._hash_items(torch.arange(n_new_i)).sum(1) > 0
        keep = ok_u[users] & ok_i[pos] & ok_i[neg]
        users, pos, neg = users[keep], pos[keep], neg[keep]
    # the trainer adds prime_pad to the OOV ids it samples (trainer / dataloader `_transform_interaction_oov`)
    pad = lambda ids, n: torch.where(ids >= n, ids + PRIME_PAD, ids)  # noqa: E731
    if emb is not None:
        u_in, p_in, n_in = pad(users, n_users), pad(pos, n_items), pad(neg, n_items)
    else:
        u_in, p_in, n_in = users, pos, neg
    inter = Interaction({"user_id": u_in.clone(), "item_id": p_in.clone(), "neg_item_id": n_in.clone()})
    loss = model.calculate_loss(inter)
    loss.backward()
    assert torch.isfinite(loss), tag
    out[f"{tag}__users"], out[f"{tag}__pos"], out[f"{tag}__neg"] = mg.np_(u_in), mg.np_(p_in), mg.np_(n_in)
    out[f"{tag}__loss"] = mg.np_(loss)
    for name, p in model.named_parameters():
        key = name.replace(".", "_")
        out[f"{tag}__w__{key}"] = mg.np_(p)
        out[f"{tag}__g__{key}"] = mg.np_(p.grad) if p.grad is not None else np.zeros((0,), np.float32)
    if emb is not None and hasattr(emb, "user_feature_mat"):
        out[f"{tag}__user_feat"] = np.asarray(mg.np_(emb.user_feature_mat) if torch.is_tensor(emb.user_feature_mat) else emb.user_feature_mat)
        out[f"{tag}__item_feat"] = np.asarray(mg.np_(emb.item_feature_mat) if torch.is_tensor(emb.item_feature_mat) else emb.item_feature_mat)
    if emb is not None and hasattr(emb, "user_lsh"):
        out[f"{tag}__user_planes"] = mg.np_(emb.user_lsh.uniform_planes[0].data)
        out[f"{tag}__item_planes"] = mg.np_(emb.item_lsh.uniform_planes[0].data)
    out[f"{tag}__dims"] = np.array([n_users, n_items, n_new_u, n_new_i, D, n_ub, n_ib])
    grads = {n: float(p.grad.abs().sum()) for n, p in model.named_parameters() if p.grad is not None}
    print(f"{tag}: B={len(users)} loss={float(loss):.6f} |grad| = {grads}")


def directau_case(out, n_users, n_items, n_new_u, n_new_i, D):
    """The second caller of the plugin: the reference's DirectAU (+lsh) on a mixed in-vocabulary / OOV batch:
    normalised rows, predict, alignment + uniformity loss and every table gradient."""
    from recbole.model.general_recommender.directau import DirectAU
    tag, seed = "directau", 750
    uf = mg.features(n_new_u, [("a", 1, "float"), ("v", 9, "float")], seed, "user_id")
    itf = mg.features(n_new_i, [("y", 1, "float"), ("w", 20, "float")], seed + 1, "item_id")
    torch.manual_seed(seed + 2)
    emb = mg.LSHInductiveEmbedder(uf, itf, n_users, n_items, 8, 8, D, "cpu", PRIME_PAD, "per-feature", mg.InductiveFeatureCache())
    cfg = mg.FakeConfig(USER_ID_FIELD="user_id", ITEM_ID_FIELD="item_id", NEG_PREFIX="neg_", device="cpu", embedding_size=D,
                        add_oov_buckets=True, user_oov_buckets=8, item_oov_buckets=8, oov_freeze_embedding=False, gamma=0.7)
    torch.manual_seed(seed + 3)
    model = DirectAU(cfg, mg.FakeDataset(n_users, n_items), None, emb)
    model.train()
    emb.set_train()
    g = torch.Generator().manual_seed(seed + 4)
    users = torch.randint(1, n_new_u, (200,), generator=g)
    items = torch.randint(1, n_new_i, (200,), generator=g)
    with torch.no_grad():
        ok = (emb._hash_users(torch.arange(n_new_u)).sum(1) > 0)[users] & (emb._hash_items(torch.arange(n_new_i)).sum(1) > 0)[items]
    users, items = users[ok], items[ok]
    u_in = torch.where(users >= n_users, users + PRIME_PAD, users)
    i_in = torch.where(items >= n_items, items + PRIME_PAD, items)
    inter = Interaction({"user_id": u_in.clone(), "item_id": i_in.clone()})
    loss = model.calculate_loss(inter)
    loss.backward()
    with torch.no_grad():
        ue, ie = model.forward(u_in.clone(), i_in.clone())
        pred = model.predict(Interaction({"user_id": u_in.clone(), "item_id": i_in.clone()}))
    out[f"{tag}__users"], out[f"{tag}__items"] = mg.np_(u_in), mg.np_(i_in)
    out[f"{tag}__loss"], out[f"{tag}__pred"] = mg.np_(loss), mg.np_(pred)
    out[f"{tag}__user_e"], out[f"{tag}__item_e"] = mg.np_(ue), mg.np_(ie)
    for name, p in model.named_parameters():
        key = name.replace(".", "_")
        out[f"{tag}__w__{key}"] = mg.np_(p)
        out[f"{tag}__g__{key}"] = mg.np_(p.grad) if p.grad is not None else np.zeros((0,), np.float32)
    out[f"{tag}__user_feat"], out[f"{tag}__item_feat"] = mg.np_(emb.user_feature_mat), mg.np_(emb.item_feature_mat)
    out[f"{tag}__user_planes"] = mg.np_(emb.user_lsh.uniform_planes[0].data)
    out[f"{tag}__item_planes"] = mg.np_(emb.item_lsh.uniform_planes[0].data)
    out[f"{tag}__dims"] = np.array([n_users, n_items, n_new_u, n_new_i, D, 8, 8])
    print(f"directau: B={len(users)} loss={float(loss.detach()):.6f}")


def main():
    torch.set_num_threads(4)
    out = {}
    n_users, n_items, n_new_u, n_new_i, D = 200, 260, 280, 360, 64
    args = (n_users, n_items, n_new_u, n_new_i, D)
    one_case("lsh", lambda uf, itf: (None, mg.LSHInductiveEmbedder(uf, itf, n_users, n_items, 8, 8, D, "cpu", PRIME_PAD,
                                                                  "per-feature", mg.InductiveFeatureCache())),
             *args, 8, 8, 700, out)
    one_case("slsh8", lambda uf, itf: (None, mg.SingleLSHInductiveEmbedder(uf, itf, n_users, n_items, 8, 8, D, "cpu",
                                                                         PRIME_PAD, "per-feature")),
             *args, 8, 8, 710, out)
    one_case("slsh200", lambda uf, itf: (None, mg.SingleLSHInductiveEmbedder(uf, itf, n_users, n_items, 200, 200, D, "cpu",
                                                                           PRIME_PAD, "none")),
             *args, 200, 200, 720, out)
    one_case("knn", lambda uf, itf: (None, mg.KNNInductiveEmbedder(uf, itf, n_users, n_items, 8, 8, D, "cpu", PRIME_PAD,
                                                                  n_neighbors=2)),
             *args, 8, 8, 730, out)
    one_case("mapper", lambda uf, itf: (mg.RandomOOVInductiveMapper(uf, itf, n_users, n_items, 8, 8, D, "cpu", PRIME_PAD,
                                                                   "3round"), None),
             *args, 8, 8, 740, out)
    directau_case(out, n_users, n_items, n_new_u, n_new_i, D)
    np.savez_compressed(os.path.join(HERE, "bpr_grad.npz"), **out)
    print("bpr_grad.npz", os.path.getsize(os.path.join(HERE, "bpr_grad.npz")), "bytes")
    hash_net_cases(*args)


def hash_net_cases(n_users, n_items, n_new_u, n_new_i, D):
    """The MLP plugins under the reference's autograd: `dnn` (features -> MLP) and `dhe` (1024... here 32 SipHash values
    -> MLP) in one BPR.calculate_loss; gradients of every Linear weight / bias and of both embedding tables.
    The dhe net is fed raw hashes up to 1.6e7, which saturate its sigmoid with default weights (every gradient would
    be exactly 0): its first layer is scaled by 1e-7 here so that the step is not vacuous.  -> hash_net_grad.npz"""
    from recbole.inductive.dnn_embedder import DNNEmbedder
    from recbole.inductive.dh_embedder import DeepHashEmbedder
    out = {}
    K = 32
    for tag, seed in (("dnn", 760), ("dhe", 770)):
        os.makedirs("hash_keys", exist_ok=True)
        if tag == "dhe":
            import json
            keys = [bytes((j * 7 + i) % 256 for i in range(16)) for j in range(K)]
            with open(os.path.join("hash_keys", f"{K}.hashes"), "w") as f:
                json.dump([k.hex() for k in keys], f)
            out["dhe__keys"] = np.frombuffer(b"".join(keys), dtype=np.uint8).reshape(K, 16)

        def build(uf, itf, tag=tag):
            if tag == "dnn":
                return None, DNNEmbedder(uf, itf, n_users, n_items, 8, 8, D, "cpu", PRIME_PAD, dhe_layer_size=48)
            e = DeepHashEmbedder(uf, itf, n_users, n_items, 8, 8, D, "cpu", PRIME_PAD, num_hashes=K)
            return None, e

        def post(model, tag=tag):  # after BPR's xavier re-initialisation of the Linear layers (bpr.py:46)
            if tag == "dhe":
                with torch.no_grad():
                    for net in (model.inductive_embedder.user_hash_net, model.inductive_embedder.item_hash_net):
                        net[0].weight.mul_(1e-7)

        one_case(tag, build, n_users, n_items, n_new_u, n_new_i, D, 8, 8, seed, out, post=post)
    np.savez_compressed(os.path.join(HERE, "hash_net_grad.npz"), **out)
    print("hash_net_grad.npz", os.path.getsize(os.path.join(HERE, "hash_net_grad.npz")), "bytes")


if __name__ == "__main__":
    main()
