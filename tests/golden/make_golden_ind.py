#!/usr/bin/env python3
"""Golden for the inductive dataset format, produced by the REAL reference:

    python tests/golden/make_golden_ind.py        (build container only; needs /root/reference)

1. Cuts a small transductive / inductive pair out of the bundled ml-100k sample (the reference ships no script that
   makes its `X_ind` datasets; S/perform_hashing.py:101-138 only loads them):
     tests/golden/ind_dataset/ml-100k_tr/ml-100k_tr.{inter,user,item}        what the model was trained on: old users x old items
     tests/golden/ind_dataset/ml-100k_ind/ml-100k_ind.{train,empty,test_filt}.inter   + `is_new:token` (-1 old, 1 new)
     tests/golden/ind_dataset/ml-100k_ind/ml-100k_ind.{user,item}            every entity, new ones interleaved
   The feature files of the pair list their rows in different orders and the inductive one has tokens the transductive
   one never saw, so the two datasets number their feature tokens differently -- what `remap_features` exists to undo.
2. Loads the pair through the reference's own Config / create_dataset(inductive=True) / set_orig_dataset / build()
   (R/data/dataset/inductive_dataset.py:73-190) and writes tests/golden/ind_dataset.npz: the transductive feature
   tensors, the inductive feature tensors BEFORE and AFTER `remap_features`, the remapped interaction ids per
   benchmark part and the vocabulary sizes.
"""
import copy
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

SRC = "/root/reference/RecBole/dataset/ml-100k/ml-100k"
OUT = os.path.join(HERE, "ind_dataset")
LOAD_COL = {"inter": ["user_id", "item_id", "rating", "timestamp"],
            "user": ["user_id", "age", "gender", "occupation", "zip_code"],
            "item": ["item_id", "movie_title", "release_year", "class"]}


def read(path):
    with open(path, encoding="utf-8") as f:
        lines = f.read().split("\n")
    return lines[0], [l for l in lines[1:] if l]


def write(path, header, rows):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w", encoding="utf-8") as f:
        f.write(header + "\n")
        for r in rows:
            f.write(r + "\n")


def cut():
    ih, inter = read(SRC + ".inter")
    uh, users = read(SRC + ".user")
    th, items = read(SRC + ".item")
    keep_u = {u.split("\t")[0] for u in users if int(u.split("\t")[0]) <= 160}
    keep_i = {i.split("\t")[0] for i in items if int(i.split("\t")[0]) <= 320}
    new_u = {u for u in keep_u if int(u) % 5 == 0}
    new_i = {i for i in keep_i if int(i) % 4 == 0}
    inter = [r for r in inter if r.split("\t")[0] in keep_u and r.split("\t")[1] in keep_i]
    train, test = [], []
    for n, r in enumerate(inter):
        u, i = r.split("\t")[:2]
        if u in new_u or i in new_i:
            test.append(r + "\t1")
        elif n % 9 == 0:
            test.append(r + "\t-1")   # held-out old x old rows ride along in the test file
        else:
            train.append(r)
    old_u = {r.split("\t")[0] for r in train}
    old_i = {r.split("\t")[1] for r in train}
    # transductive twin: the train interactions, feature rows of the entities seen in training, file order reversed
    write(f"{OUT}/ml-100k_tr/ml-100k_tr.inter", ih, train)
    write(f"{OUT}/ml-100k_tr/ml-100k_tr.user", uh, [u for u in reversed(users) if u.split("\t")[0] in old_u])
    write(f"{OUT}/ml-100k_tr/ml-100k_tr.item", th, [i for i in reversed(items) if i.split("\t")[0] in old_i])
    # inductive twin
    ihn = ih + "\tis_new:token"
    write(f"{OUT}/ml-100k_ind/ml-100k_ind.train.inter", ihn, [r + "\t-1" for r in train])
    write(f"{OUT}/ml-100k_ind/ml-100k_ind.empty.inter", ihn, [])
    seen_u = old_u | {r.split("\t")[0] for r in test}
    seen_i = old_i | {r.split("\t")[1] for r in test}
    write(f"{OUT}/ml-100k_ind/ml-100k_ind.test_filt.inter", ihn, test)
    write(f"{OUT}/ml-100k_ind/ml-100k_ind.user", uh, [u for u in users if u.split("\t")[0] in seen_u])
    write(f"{OUT}/ml-100k_ind/ml-100k_ind.item", th, [i for i in items if i.split("\t")[0] in seen_i])
    print(f"cut: {len(train)} train rows, {len(test)} test rows; {len(old_u)} old / {len(seen_u) - len(old_u)} new users, "
          f"{len(old_i)} old / {len(seen_i) - len(old_i)} new items")


def feats(interaction):
    return {k: interaction[k].numpy().copy() for k in interaction.columns}


def main():
    cut()
    base = {"data_path": OUT + "/", "seed": 2020, "use_gpu": False, "load_col": LOAD_COL, "embedding_size": 64,
            "checkpoint_dir": "/tmp/mi_oov_golden_ind_ckpt", "save_dataset": False, "save_dataloaders": False}
    cfg_tr = Config(model="BPR", dataset="ml-100k_tr", config_dict=dict(base))
    orig = create_dataset(cfg_tr)
    orig._change_feat_format()
    out = {}
    for side, inter in (("user", orig.get_user_feature()), ("item", orig.get_item_feature())):
        for k, v in feats(inter).items():
            out[f"tr__{side}__{k}"] = v
    out["tr__nums"] = np.array([orig.user_num, orig.item_num])

    cfg_ind = Config(model="BPR", dataset="ml-100k_ind",
                     config_dict=dict(base, benchmark_filename=["train", "empty", "test_filt"]))
    ind = create_dataset(cfg_ind, inductive=True)
    ind.set_orig_dataset(orig)
    raw = copy.copy(ind)   # `_change_feat_format` is not idempotent: the untouched features come from a shallow copy
    raw._change_feat_format()
    for side, inter in (("user", raw.get_user_feature()), ("item", raw.get_item_feature())):
        for k, v in feats(inter).items():
            out[f"ind_raw__{side}__{k}"] = v
    parts = ind.build()   # _change_feat_format + remap_features + the split by file sizes
    for side, inter in (("user", ind.get_user_feature()), ("item", ind.get_item_feature())):
        for k, v in feats(inter).items():
            out[f"ind__{side}__{k}"] = v
    out["ind__nums"] = np.array([ind.user_num, ind.item_num])
    for name, part in zip(("train", "empty", "test_filt"), parts):
        out[f"ind__inter__{name}__user"] = part.inter_feat["user_id"].numpy()
        out[f"ind__inter__{name}__item"] = part.inter_feat["item_id"].numpy()
        if "is_new" in part.inter_feat.columns:
            out[f"ind__inter__{name}__is_new"] = part.inter_feat["is_new"].numpy()
    out["ind__is_new_tokens"] = np.array([str(t) for t in ind.field2id_token["is_new"]])
    # the checks S/perform_hashing.py:112-138 makes: old rows of the remapped inductive features equal the transductive ones
    for side in ("user", "item"):
        for k in [c for c in out if c.startswith(f"tr__{side}__") and not c.endswith("_id")]:
            col = k.split("__")[2]
            a, b = out[k], out[f"ind__{side}__{col}"]
            same = np.array_equal(a[1:], b[1:a.shape[0]])
            print(f"{side}.{col}: train rows equal after remap: {same}   (before: {np.array_equal(a[1:], out[f'ind_raw__{side}__{col}'][1:a.shape[0]][:, :a.shape[1]] if a.ndim > 1 else out[f'ind_raw__{side}__{col}'][1:a.shape[0]])})")
    np.savez_compressed(os.path.join(HERE, "ind_dataset.npz"), **out)
    print("ind_dataset.npz", os.path.getsize(os.path.join(HERE, "ind_dataset.npz")), "bytes;",
          sum(os.path.getsize(os.path.join(r, f)) for r, _, fs in os.walk(OUT) for f in fs), "bytes of dataset files")


if __name__ == "__main__":
    torch.manual_seed(0)
    main()
