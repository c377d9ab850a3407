#!/usr/bin/env python3
"""Generate the golden fixtures of tests/golden/ by RUNNING THE REAL REFERENCE.

Run by hand in the build container (the only place /root/reference exists):

    python tests/golden/make_golden.py

It imports recbole.inductive.* / BPR from /root/reference/RecBole through the leaf-dependency
shims of ref_shims.py, drives them on small seeded inputs and writes inputs + expected outputs
as .npz / .json next to this file.  The fixtures are data only (no reference source); they are
committed, the reference never travels.  Everything is CPU, torch default dtype float32.

Fixture index (see tests/test_oracle_golden.py for how each is used):
  mapper.json        RandomOOVInductiveMapper raw hashes + map_*_ids for mod/fast/3round/64bit
  siphash.json       SipHash-2-4 paper vector + DeepHashEmbedder._hash_ids outputs
  lsh_<case>.npz     LSHInductiveEmbedder: feature matrices as built by the ctor, planes, bucket
                     tables, ids, bits, embeddings, |projection| margins
  slsh_<case>.npz    SingleLSHInductiveEmbedder: bucket ids + embeddings
  dhe.npz            DeepHashEmbedder: keys, hash matrix, MLP state, pre-sigmoid and output
  knn.npz            KNNInductiveEmbedder aggregate (neighbour search stubbed exact: unpinned)
  mean.npz           MeanEmbedder / ZeroEmbedder
  bpr_lsh.npz        BPR(+lsh / +mapper) get_*_embedding, predict, full_sort, ind_full_sort
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()

import torch  # noqa: E402
from recbole.data.interaction import Interaction  # noqa: E402
from recbole.inductive.dh_embedder import DeepHashEmbedder  # noqa: E402
from recbole.inductive.feature_cache import InductiveFeatureCache  # noqa: E402
from recbole.inductive.knn_embedder import KNNInductiveEmbedder  # noqa: E402
from recbole.inductive.lsh_embedder import LSHInductiveEmbedder  # noqa: E402
from recbole.inductive.mean_embedder import MeanEmbedder  # noqa: E402
from recbole.inductive.random_mapper import RandomOOVInductiveMapper  # noqa: E402
from recbole.inductive.single_lsh_embedder import SingleLSHInductiveEmbedder  # noqa: E402
from recbole.inductive.zero_embedder import ZeroEmbedder  # noqa: E402
from recbole.model.general_recommender.bpr import BPR  # noqa: E402

PRIME_PAD = 112062759511  # overall.yaml: oov_prime_pad


def np_(t):
    return t.detach().cpu().numpy()


class FakeConfig(dict):
    """config[key] -> None for missing keys, like recbole Config (configurator.py:583-584)."""

    def __getitem__(self, k):
        return self.get(k, None)


class FakeDataset:
    def __init__(self, n_users, n_items):
        self.n = {"user_id": n_users, "item_id": n_items}

    def num(self, field):
        return self.n[field]


class FakeModel(torch.nn.Module):
    """What lsh/slsh read from the model: the two OOV bucket tables."""

    def __init__(self, n_ub, n_ib, D, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.user_oov_buckets = torch.nn.Embedding(n_ub, D)
        self.item_oov_buckets = torch.nn.Embedding(n_ib, D)
        with torch.no_grad():
            self.user_oov_buckets.weight.copy_(torch.randn(n_ub, D, generator=g))
            self.item_oov_buckets.weight.copy_(torch.randn(n_ib, D, generator=g))


def features(n, cols, seed, id_name):
    """cols: list of (name, width, kind); kind 'float' -> randn, 'token' -> small ints."""
    g = torch.Generator().manual_seed(seed)
    d = {id_name: torch.arange(n)}
    for name, width, kind in cols:
        shape = (n,) if width == 1 else (n, width)
        if kind == "token":
            t = torch.randint(0, 7, shape, generator=g)
        else:
            t = torch.randn(shape, generator=g)
        if n > 0:
            t[0] = 0  # padding row 0 is all zeros in RecBole feature tables
        d[name] = t
    return Interaction(d)


def margins(feat, ids, planes):
    x = feat[ids].double()
    return (x @ planes.double().T).abs().min(dim=1).values.float()


# ----------------------------------------------------------------------------------------------
def gen_mapper():
    ids = [0, 1, 5, 14, 15, 16, 17, 100, 1000, 99999, 12345678901, PRIME_PAD + 3, PRIME_PAD + 20,
           2 ** 40 + 7, 2 ** 62 + 12345]
    raw_in = [0, 1, 2, 3, 17, 255, 65536, 2 ** 31 - 1, 2 ** 31, 2 ** 32 + 5, 2 ** 40 + 7, PRIME_PAD, 2 ** 62 + 1,
              2 ** 63 - 1]
    out = {"ids": ids, "raw_in": raw_in, "cases": []}
    t_ids = torch.tensor(ids, dtype=torch.int64)
    for hf in ("mod", "fast", "3round", "64bit"):
        for n_orig, nb in ((15, 8), (100, 1000), (1, 1), (16, 37)):
            m = RandomOOVInductiveMapper([0] * 20, [0] * 30, n_orig, n_orig, nb, nb, 64, "cpu", PRIME_PAD, hf)
            out["cases"].append({"hash": hf, "n_orig": n_orig, "n_buckets": nb,
                                 "map_user": m.map_user_ids(t_ids.clone()).tolist(),
                                 "map_item": m.map_item_ids(t_ids.clone()).tolist()})
    m = RandomOOVInductiveMapper([0] * 20, [0] * 30, 15, 15, 8, 8, 64, "cpu", PRIME_PAD, "fast")
    t_raw = torch.tensor(raw_in, dtype=torch.int64)
    out["raw_fast"] = m._fast_int_hash(t_raw.clone()).tolist()
    out["raw_3round"] = m._three_round_int_hash(t_raw.clone()).tolist()
    for nb in (8, 1000, 2 ** 31 + 11):
        out[f"raw_64bit_mod_{nb}"] = m._big_64bit_hash(t_raw.clone(), nb).tolist()
    # set_train / set_eval bookkeeping (random_mapper.py:60-68)
    m.set_train()
    out["train_n_new"] = [m.n_new_users, m.n_new_items]
    m.set_eval()
    out["eval_n_new"] = [m.n_new_users, m.n_new_items]
    json.dump(out, open(os.path.join(HERE, "mapper.json"), "w"))
    print("mapper.json", len(out["cases"]), "cases")


def gen_lsh_case(name, n, ucols, icols, norm, n_ub, n_ib, D, B, seed, train):
    uf = features(n, ucols, seed, "user_id")
    itf = features(n + 37, icols, seed + 1, "item_id")
    torch.manual_seed(seed + 2)  # planes are torch.randn in the ctor
    emb = LSHInductiveEmbedder(uf, itf, n // 2, n // 2, n_ub, n_ib, D, "cpu", PRIME_PAD, norm,
                               InductiveFeatureCache())
    model = FakeModel(n_ub, n_ib, D, seed + 3)
    g = torch.Generator().manual_seed(seed + 4)
    uids = torch.randint(0, n, (B,), generator=g)
    iids = torch.randint(0, n + 37, (B,), generator=g)
    uids[0] = 0
    iids[0] = 0  # padding row: all projections 0 -> all bits 1
    uids_in, iids_in = uids.clone(), iids.clone()
    if train:
        emb.set_train()
        uids_in[::3] += PRIME_PAD
        iids_in[1::4] += PRIME_PAD
    u_arg, i_arg = uids_in.clone(), iids_in.clone()
    with torch.no_grad():
        ue = emb.embed_user_ids(u_arg, model)
        ie = emb.embed_item_ids(i_arg, model)
        ub = emb._hash_users(uids)
        ib = emb._hash_items(iids)
    up, ip = emb.user_lsh.uniform_planes[0].data, emb.item_lsh.uniform_planes[0].data
    np.savez_compressed(
        os.path.join(HERE, f"lsh_{name}.npz"),
        user_feat=np_(emb.user_feature_mat), item_feat=np_(emb.item_feature_mat),
        user_planes=np_(up), item_planes=np_(ip),
        user_buckets=np_(model.user_oov_buckets.weight), item_buckets=np_(model.item_oov_buckets.weight),
        user_ids_in=np_(uids_in), item_ids_in=np_(iids_in),
        user_ids_after=np_(u_arg), item_ids_after=np_(i_arg),  # in-place prime-pad strip (lsh_embedder.py:153-155)
        user_ids=np_(uids), item_ids=np_(iids),
        user_bits=np_(ub).astype(np.uint8), item_bits=np_(ib).astype(np.uint8),
        user_emb=np_(ue), item_emb=np_(ie),
        user_margin=np_(margins(emb.user_feature_mat, uids, up)),
        item_margin=np_(margins(emb.item_feature_mat, iids, ip)),
        train=np.array(train), norm=np.array(norm),
        # raw Interaction columns (small cases only): inputs of the ctor-time feature build
        **({"ucol_" + c: np_(uf[c]) for c in uf.columns} if name in ("mixed", "global") else {}),
        **({"icol_" + c: np_(itf[c]) for c in itf.columns} if name in ("mixed", "global") else {}),
        ucols=np.array(list(uf.columns)), icols=np.array(list(itf.columns)))
    nan_rows = int(torch.isnan(ie).any(1).sum())
    print(f"lsh_{name}.npz  F_u={emb.user_feature_mat.shape[1]} F_i={emb.item_feature_mat.shape[1]} "
          f"H={n_ib} D={D} B={B} nan_rows(item)={nan_rows} min_margin={float(margins(emb.item_feature_mat, iids, ip)[1:].min()):.3e}")


def gen_slsh_case(name, n, ucols, icols, norm, n_ub, n_ib, D, B, seed, train):
    uf = features(n, ucols, seed, "user_id")
    itf = features(n + 11, icols, seed + 1, "item_id")
    torch.manual_seed(seed + 2)
    emb = SingleLSHInductiveEmbedder(uf, itf, n // 2, n // 2, n_ub, n_ib, D, "cpu", PRIME_PAD, norm)
    model = FakeModel(n_ub, n_ib, D, seed + 3)
    g = torch.Generator().manual_seed(seed + 4)
    uids = torch.randint(0, n, (B,), generator=g)
    iids = torch.randint(0, n + 11, (B,), generator=g)
    uids[0] = 0
    uids_in, iids_in = uids.clone(), iids.clone()
    if train:
        emb.set_train()
        iids_in[::2] += PRIME_PAD
    with torch.no_grad():
        ue = emb.embed_user_ids(uids_in.clone(), model)
        ie = emb.embed_item_ids(iids_in.clone(), model)
        uidx = emb._hash_users(uids)
        iidx = emb._hash_items(iids)
    up, ip = emb.user_lsh.uniform_planes[0].data, emb.item_lsh.uniform_planes[0].data
    np.savez_compressed(
        os.path.join(HERE, f"slsh_{name}.npz"),
        user_feat=np_(emb.user_feature_mat), item_feat=np_(emb.item_feature_mat),
        user_planes=np_(up), item_planes=np_(ip),
        user_buckets=np_(model.user_oov_buckets.weight), item_buckets=np_(model.item_oov_buckets.weight),
        user_ids_in=np_(uids_in), item_ids_in=np_(iids_in), user_ids=np_(uids), item_ids=np_(iids),
        user_idx=np_(uidx), item_idx=np_(iidx), user_emb=np_(ue), item_emb=np_(ie),
        user_margin=np_(margins(emb.user_feature_mat, uids, up)),
        item_margin=np_(margins(emb.item_feature_mat, iids, ip)),
        bits_req=np.array([emb.user_bits_req, emb.item_bits_req]), train=np.array(train))
    print(f"slsh_{name}.npz bits_req={emb.user_bits_req},{emb.item_bits_req} distinct item idx={sorted(set(iidx.tolist()))[:12]}")


def gen_siphash_dhe():
    import hashlib
    out = {}
    key = bytes(range(16))
    # Appendix A of the SipHash paper: key 00..0f, message 00..0e -> a129ca6149be45e5
    out["paper_vector"] = {"key": key.hex(), "msg": bytes(range(15)).hex(),
                           "hash_hex": ref_shims.siphash24_py(key, bytes(range(15)))[::-1].hex()}
    assert out["paper_vector"]["hash_hex"] == "a129ca6149be45e5"
    out["by_len"] = [{"len": n, "hash_le_hex": ref_shims.siphash24_py(key, bytes(range(n))).hex()} for n in range(0, 24)]

    K = 16
    keys = [hashlib.sha256(b"mi-oov-key-%d" % j).digest()[:16] for j in range(K)]
    os.makedirs("/tmp/mi_oov_golden/hash_keys", exist_ok=True)
    cwd = os.getcwd()
    os.chdir("/tmp/mi_oov_golden")  # DeepHashEmbedder reads ./hash_keys/{K}.hashes relative to CWD
    json.dump([k.hex() for k in keys], open(f"hash_keys/{K}.hashes", "w"))
    n, D = 64, 8
    uf = features(n, [("age", 1, "float")], 10, "user_id")
    itf = features(n, [("year", 1, "float")], 11, "item_id")
    torch.manual_seed(12)
    emb = DeepHashEmbedder(uf, itf, 32, 32, 8, 8, D, "cpu", PRIME_PAD, K)
    os.chdir(cwd)
    assert [k.hex() for k in emb.hash_keys] == [k.hex() for k in keys]
    ids = torch.tensor([0, 1, 2, 3, 31, 32, 63, 1000, 2 ** 31, 2 ** 40 + 7, PRIME_PAD + 5, 112062759516], dtype=torch.int64)
    with torch.no_grad():
        hm = emb._hash_ids(ids)
        pre = emb.item_hash_net[:-1](hm.float())
        oute = emb.embed_item_ids(ids, None)
        outu = emb.embed_user_ids(ids, None)
    out["dhe_keys"] = [k.hex() for k in keys]
    out["dhe_ids"] = ids.tolist()
    out["dhe_hashes"] = hm.long().tolist()
    # the independent values recorded in SURVEY.md section 8 (key_j = bytes(range(j, j+16)), K = 4)
    emb.hash_keys = [bytes(range(j, j + 16)) for j in range(4)]
    emb._get_hashes.cache_clear()
    sv = emb._hash_ids(torch.tensor([0, 1, 2, 112062759516])).long().tolist()
    out["survey_k4"] = sv
    assert sv[0] == [7766439, 3375036, 11690175, 4162672], sv
    json.dump(out, open(os.path.join(HERE, "siphash.json"), "w"))
    sd = {k: np_(v) for k, v in emb.state_dict().items() if k.startswith("item_hash_net")}
    np.savez_compressed(os.path.join(HERE, "dhe.npz"), ids=np_(ids), hashes=np_(hm), item_pre_sigmoid=np_(pre),
                        item_out=np_(oute),
                        keys=np.frombuffer(b"".join(keys), dtype=np.uint8).reshape(K, 16),
                        **{k.replace(".", "__"): v for k, v in sd.items()})
    print("siphash.json, dhe.npz  hashes[0,:4] =", hm[0, :4].tolist())


class FakeBPRForKNN(BPR):
    pass


def make_bpr(n_users, n_items, D, mapper, embedder, n_ub, n_ib, seed):
    cfg = FakeConfig(USER_ID_FIELD="user_id", ITEM_ID_FIELD="item_id", NEG_PREFIX="neg_", device="cpu",
                     embedding_size=D, add_oov_buckets=True, user_oov_buckets=n_ub, item_oov_buckets=n_ib,
                     oov_freeze_embedding=False)
    torch.manual_seed(seed)
    return BPR(cfg, FakeDataset(n_users, n_items), mapper, embedder)


def gen_knn_mean():
    n_users, n_items, D = 300, 400, 16
    n_new_u, n_new_i = 380, 520
    uf = features(n_new_u, [("a", 1, "float"), ("v", 6, "float")], 20, "user_id")
    itf = features(n_new_i, [("y", 1, "float"), ("w", 12, "float"), ("c", 3, "token")], 21, "item_id")
    knn = KNNInductiveEmbedder(uf, itf, n_users, n_items, 8, 8, D, "cpu", PRIME_PAD, n_neighbors=2)
    model = make_bpr(n_users, n_items, D, None, knn, 8, 8, 22)
    g = torch.Generator().manual_seed(23)
    uids = torch.randint(n_users, n_new_u, (64,), generator=g)
    iids = torch.randint(n_items, n_new_i, (96,), generator=g)
    with torch.no_grad():
        uidx = knn._hash_users(uids)
        iidx = knn._hash_items(iids)
        ue = knn.embed_user_ids(uids.clone(), model)
        ie = knn.embed_item_ids(iids.clone(), model)
    np.savez_compressed(os.path.join(HERE, "knn.npz"),
                        user_feat=knn.user_feature_mat, item_feat=knn.item_feature_mat,
                        n_users=np.array(n_users), n_items=np.array(n_items),
                        user_ids=np_(uids), item_ids=np_(iids), user_idx=np_(uidx), item_idx=np_(iidx),
                        user_table=np_(model.user_embedding.weight), item_table=np_(model.item_embedding.weight),
                        user_emb=np_(ue), item_emb=np_(ie))
    mean = MeanEmbedder(uf, itf, n_users, n_items, 8, 8, D, "cpu")
    zero = ZeroEmbedder(uf, itf, n_users, n_items, D, "cpu")
    model2 = make_bpr(n_users, n_items, D, None, mean, 8, 8, 24)
    with torch.no_grad():
        mu = mean.embed_user_ids(uids, model2)
        mi = mean.embed_item_ids(iids, model2)
        zu = zero.embed_user_ids(uids, model2)
    np.savez_compressed(os.path.join(HERE, "mean.npz"), user_table=np_(model2.user_embedding.weight),
                        item_table=np_(model2.item_embedding.weight), user_ids=np_(uids), item_ids=np_(iids),
                        user_emb=np_(mu), item_emb=np_(mi), zero_user_emb=np_(zu))
    print("knn.npz, mean.npz")


def gen_bpr():
    n_users, n_items, D, H = 200, 260, 64, 8
    n_new_u, n_new_i = 260, 330
    uf = features(n_new_u, [("a", 1, "float"), ("v", 9, "float")], 30, "user_id")
    itf = features(n_new_i, [("y", 1, "float"), ("w", 20, "float"), ("z", 1, "token")], 31, "item_id")
    torch.manual_seed(32)
    lsh = LSHInductiveEmbedder(uf, itf, n_users, n_items, H, H, D, "cpu", PRIME_PAD, "per-feature",
                               InductiveFeatureCache())
    model = make_bpr(n_users, n_items, D, None, lsh, H, H, 33)
    model.eval()
    g = torch.Generator().manual_seed(34)
    users = torch.randint(1, n_new_u, (257,), generator=g)
    items = torch.randint(1, n_new_i, (257,), generator=g)
    inter = Interaction({"user_id": users, "item_id": items})
    all_items = torch.arange(n_new_i)
    with torch.no_grad():
        ue = model.get_user_embedding(users.clone())
        ie = model.get_item_embedding(items.clone())
        pred = model.predict(inter)
        fs = model.full_sort_predict(Interaction({"user_id": users[:40]}))
        ifs = model.ind_full_sort_predict(Interaction({"user_id": users[:40]}), all_items)
    # mapper-only BPR (no embedder): OOV ids -> random bucket rows (bpr.py:75,122)
    mapper = RandomOOVInductiveMapper(uf, itf, n_users, n_items, H, H, D, "cpu", PRIME_PAD, "3round")
    model_m = make_bpr(n_users, n_items, D, mapper, None, H, H, 35)
    model_m.eval()
    with torch.no_grad():
        ue_m = model_m.get_user_embedding(users.clone())
        ie_m = model_m.get_item_embedding(items.clone())
        pred_m = model_m.predict(inter)
    np.savez_compressed(
        os.path.join(HERE, "bpr_lsh.npz"),
        n_users=np.array(n_users), n_items=np.array(n_items),
        user_feat=np_(lsh.user_feature_mat), item_feat=np_(lsh.item_feature_mat),
        user_planes=np_(lsh.user_lsh.uniform_planes[0].data), item_planes=np_(lsh.item_lsh.uniform_planes[0].data),
        user_table=np_(model.user_embedding.weight), item_table=np_(model.item_embedding.weight),
        user_buckets=np_(model.user_oov_buckets.weight), item_buckets=np_(model.item_oov_buckets.weight),
        users=np_(users), items=np_(items), user_e=np_(ue), item_e=np_(ie), predict=np_(pred),
        full_sort=np_(fs), ind_full_sort=np_(ifs),
        m_user_table=np_(model_m.user_embedding.weight), m_item_table=np_(model_m.item_embedding.weight),
        m_user_buckets=np_(model_m.user_oov_buckets.weight), m_item_buckets=np_(model_m.item_oov_buckets.weight),
        m_user_e=np_(ue_m), m_item_e=np_(ie_m), m_predict=np_(pred_m))
    print("bpr_lsh.npz  state_dict keys:", [k for k in model.state_dict().keys()])


def main():
    torch.set_num_threads(4)
    gen_mapper()
    vec64 = [("vec", 64, "float")]
    mixed_u = [("age", 1, "float"), ("gender", 1, "token"), ("occ", 1, "token"), ("zip", 1, "float")]
    mixed_i = [("year", 1, "float"), ("title", 12, "token"), ("genre", 9, "float")]
    gen_lsh_case("f64", 2048, vec64, vec64, "per-feature", 8, 8, 64, 2048, 100, train=False)
    gen_lsh_case("mixed", 1500, mixed_u, mixed_i, "per-feature", 8, 8, 64, 1024, 200, train=True)
    gen_lsh_case("global", 900, [("a", 3, "float"), ("b", 7, "float")], [("c", 2, "float"), ("d", 5, "float"), ("e", 3, "float")],
                 "global", 16, 12, 32, 512, 300, train=False)
    gen_lsh_case("wide", 400, [("v", 130, "float")], [("w", 300, "float"), ("s", 1, "float")], "none", 5, 40, 50, 384, 400,
                 train=False)
    gen_slsh_case("b8", 1200, mixed_u, vec64, "per-feature", 8, 8, 64, 1024, 500, train=True)
    gen_slsh_case("b1000", 800, [("v", 20, "float")], [("w", 33, "float")], "none", 1000, 777, 24, 512, 600, train=False)
    gen_siphash_dhe()
    gen_knn_mean()
    gen_bpr()


if __name__ == "__main__":
    main()
