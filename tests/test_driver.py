"""run_recbole.py entry: flag parsing and atomic-file loading on the CPU, a short end-to-end training +
uni250 evaluation on the GPU for several plugins (synthetic RecBole-format dataset written to tmp)."""
import os

import numpy as np
import pytest
import torch


def write_dataset(root, name="toy", n_users=300, n_items=400, n_inter=20000, seed=0):
    rng = np.random.default_rng(seed)
    d = os.path.join(root, name)
    os.makedirs(d, exist_ok=True)
    genres = ["Action", "Comedy", "Drama", "Horror", "SciFi", "Romance"]
    # interactions follow a hidden taste: users and items share a genre affinity, so the model can learn
    ug = rng.integers(0, len(genres), n_users)
    ig = rng.integers(0, len(genres), n_items)
    with open(os.path.join(d, f"{name}.inter"), "w") as f:
        f.write("user_id:token\titem_id:token\trating:float\ttimestamp:float\n")
        for _ in range(n_inter):
            u = rng.integers(0, n_users)
            cands = np.flatnonzero(ig == ug[u]) if rng.random() < 0.95 else np.arange(n_items)
            f.write(f"u{u}\ti{rng.choice(cands)}\t{rng.integers(1, 6)}\t{rng.integers(1, 10**6)}\n")
    with open(os.path.join(d, f"{name}.user"), "w") as f:
        f.write("user_id:token\tage:float\tgender:token\tfav:token\n")
        for u in range(n_users):
            f.write(f"u{u}\t{rng.integers(18, 70)}\t{'MF'[rng.integers(0, 2)]}\t{genres[ug[u]]}\n")
    with open(os.path.join(d, f"{name}.item"), "w") as f:
        f.write("item_id:token\trelease_year:float\tclass:token_seq\tvec:float_seq\n")
        for i in range(n_items):
            extra = genres[rng.integers(0, len(genres))]
            vec = " ".join(f"{x:.3f}" for x in rng.standard_normal(4))
            f.write(f"i{i}\t{rng.integers(1950, 2020)}\t{genres[ig[i]]} {extra}\t{vec}\n")
    return root


def write_split_dataset(root, name="toy_ind", n_users=300, n_items=400, n_inter=20000, seed=1):
    """The reference's inductive layout (S/perform_hashing.py:101-138): <name>.train.inter / .empty.inter /
    .test_filt.inter; the last 20 % of the users and items only ever appear in the test part."""
    base = write_dataset(root, "tmp_src", n_users, n_items, n_inter, seed)
    src = os.path.join(base, "tmp_src")
    d = os.path.join(root, name)
    os.makedirs(d, exist_ok=True)
    lines = open(os.path.join(src, "tmp_src.inter")).read().splitlines()
    header, rows = lines[0], lines[1:]
    new_u = {f"u{u}" for u in range(int(0.8 * n_users), n_users)}
    new_i = {f"i{i}" for i in range(int(0.8 * n_items), n_items)}
    rng = np.random.default_rng(seed)
    train, test = [], []
    for r in rows:
        u, i = r.split("\t")[:2]
        (test if (u in new_u or i in new_i or rng.random() < 0.05) else train).append(r)
    for part, content in (("train", train), ("empty", []), ("test_filt", test)):
        with open(os.path.join(d, f"{name}.{part}.inter"), "w") as f:
            f.write("\n".join([header] + content) + "\n")
    for ext in ("user", "item"):
        with open(os.path.join(d, f"{name}.{ext}"), "w") as f:
            f.write(open(os.path.join(src, f"tmp_src.{ext}")).read())
    return root, len({r.split("\t")[0] for r in train}), len({r.split("\t")[1] for r in train})


def test_pre_split_inductive_dataset(tmp_path):
    """benchmark_filename = [train, empty, test_filt]: ids numbered by first appearance, train first, so the
    transductive vocabulary is a prefix and every test-only entity is out-of-vocabulary."""
    from mi_oov import driver
    root, n_tu, n_ti = write_split_dataset(str(tmp_path))
    ds = driver.AtomicDataset("toy_ind", root, benchmark_filename=["train", "empty", "test_filt"])
    assert ds.n_train_users == n_tu + 1 and ds.n_train_items == n_ti + 1
    tr, te = ds.split == 0, ds.split == 2
    assert tr.sum() + te.sum() == len(ds.split) and not (ds.split == 1).any()
    assert ds.inter_user[tr].max() < ds.n_train_users and ds.inter_item[tr].max() < ds.n_train_items
    assert ds.inter_user[te].max() >= ds.n_train_users and ds.inter_item[te].max() >= ds.n_train_items  # OOV in the test part
    assert ds.user_num == 301 and ds.item_num == 401           # feature-file-only entities are appended
    assert len(ds.get_user_feature()) == ds.user_num and ds.get_item_feature()["vec"].shape == (ds.item_num, 4)


def test_parse_args_like_reference():
    from mi_oov import driver
    a = driver.custom_parse_args(["run_recbole.py", "--dataset=ml-100k", "--model=BPR", "--embedding_size=64",
                                  "--learning_rate=0.001", "--train_oov", "--oov_only_epoch=False",
                                  "--topk=[5,10]", "--inductive_embedder=lsh", "positional"])
    assert a == {"dataset": "ml-100k", "model": "BPR", "embedding_size": 64, "learning_rate": 0.001,
                 "train_oov": True, "oov_only_epoch": False, "topk": [5, 10], "inductive_embedder": "lsh"}
    cfg = driver.Config(a)
    assert cfg["not_set"] is None  # recbole Config semantics


def test_atomic_loader(tmp_path):
    from mi_oov import driver
    root = write_dataset(str(tmp_path))
    ds = driver.AtomicDataset("toy", root)
    assert ds.user_num <= 301 and ds.item_num <= 401 and ds.inter_user.min() >= 1  # 0 is the padding id
    uf, itf = ds.get_user_feature(), ds.get_item_feature()
    assert uf.columns == ["user_id", "age", "gender", "fav"] and len(uf) == ds.user_num
    assert itf.columns == ["item_id", "release_year", "class", "vec"]
    assert itf["class"].shape == (ds.item_num, 2) and itf["class"].dtype == torch.int64
    assert itf["vec"].shape == (ds.item_num, 4) and not itf["vec"][0].any()  # padding row stays zero
    from mi_oov import embedders
    mat = embedders.build_feature_matrix(itf, ds.item_num, True, "cpu")
    assert mat.shape == (ds.item_num, 1 + 2 + 4)


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [
    ["--inductive_embedder=lsh", "--add_oov_buckets", "--train_oov"],
    ["--inductive_embedder=lsh", "--add_oov_buckets", "--train_oov", "--oov_only_epoch=False"],
    ["--inductive_embedder=slsh", "--add_oov_buckets", "--train_oov"],
    ["--inductive_embedder=dhe", "--dhe_num_hashes=32", "--train_oov"],
    ["--inductive_embedder=knn"],
    ["--inductive_embedder=mean"],
    ["--inductive_mapper=random", "--add_oov_buckets"],
    ["--model=DirectAU", "--inductive_embedder=lsh", "--add_oov_buckets", "--train_oov", "--gamma=0.5"],
    ["--dataset=toy_ind", "--benchmark_filename=train,empty,test_filt", "--inductive_embedder=lsh", "--add_oov_buckets",
     "--train_oov"],
    # the hot tile end to end: 64-d embeddings, the dataset's narrow feature matrices zero-padded to 64 columns
    ["--inductive_embedder=lsh", "--add_oov_buckets", "--train_oov", "--embedding_size=64"],
    ["--inductive_embedder=slsh", "--add_oov_buckets", "--train_oov", "--embedding_size=64", "--item_oov_buckets=1000",
     "--user_oov_buckets=1000"],
    # as many hyperplanes as OOV buckets (plane chunks) and embedding rows wider than 256 floats (column windows)
    ["--inductive_embedder=lsh", "--add_oov_buckets", "--train_oov", "--embedding_size=300", "--item_oov_buckets=300",
     "--user_oov_buckets=300"],
])
def test_end_to_end(flags, tmp_path, monkeypatch, dev):
    from mi_oov import driver
    monkeypatch.chdir(tmp_path)  # dhe writes ./hash_keys
    root = write_dataset(str(tmp_path))
    if any(f.startswith("--dataset=") for f in flags):
        write_split_dataset(str(tmp_path))
    base = [] if any(f.startswith("--model=") for f in flags) else ["--model=BPR"]
    base += [] if any(f.startswith("--dataset=") for f in flags) else ["--dataset=toy"]
    args = driver.custom_parse_args(["x", f"--data_path={root}", "--embedding_size=32",
                                     "--user_oov_buckets=8", "--item_oov_buckets=8", "--epochs=4",
                                     "--learning_rate=0.01", "--train_batch_size=512"] + base + flags)
    results, model = driver.run(args)
    for slice_name in ("overall", "old_users", "new_users", "old_old", "old_new", "new_old", "new_new", "old_items", "new_items"):
        assert slice_name in results
        for k, v in results[slice_name].items():
            assert np.isfinite(v) and 0.0 <= v <= 1.0, (slice_name, k, v)
    # a positive ranked uniformly at random among 251 would give recall@10 = 10/251 = 0.04: in-vocabulary
    # users/items must do clearly better after four epochs on this planted-structure data
    assert results["old_items"]["recall@10"] > 0.07


@pytest.mark.gpu
def test_checkpoint_round_trip(tmp_path, monkeypatch, dev):
    """train -> save (tensors only, state_dict keys of the reference) -> a fresh process-equivalent run that loads the
    file and only evaluates reproduces the metrics (same seed => same sampled negatives)."""
    from mi_oov import driver
    monkeypatch.chdir(tmp_path)
    root = write_dataset(str(tmp_path))
    ck = str(tmp_path / "bpr_lsh.pth")
    common = ["x", "--dataset=toy", f"--data_path={root}", "--model=BPR", "--embedding_size=32", "--user_oov_buckets=8",
              "--item_oov_buckets=8", "--inductive_embedder=lsh", "--add_oov_buckets", "--train_oov", "--train_batch_size=512"]
    res1, model1 = driver.run(driver.custom_parse_args(common + ["--epochs=2", f"--save_checkpoint={ck}"]))
    blob = torch.load(ck, weights_only=True)
    assert set(blob["state_dict"]) == {"user_embedding.weight", "item_embedding.weight", "user_oov_buckets.weight",
                                       "item_oov_buckets.weight", "inductive_embedder.user_lsh.uniform_planes.0",
                                       "inductive_embedder.item_lsh.uniform_planes.0"}
    res2, model2 = driver.run(driver.custom_parse_args(common + ["--eval_only", f"--load_checkpoint={ck}"]))
    assert torch.equal(model1.item_oov_buckets.weight, model2.item_oov_buckets.weight)
    assert res2["overall"].keys() == res1["overall"].keys() and all(0 <= v <= 1 for v in res2["overall"].values())
    assert abs(res2["old_users"]["recall@10"] - res1["old_users"]["recall@10"]) < 0.05  # different negative samples



@pytest.mark.gpu
def test_foreign_checkpoint_and_nan_policy(tmp_path, monkeypatch, dev):
    """A file laid out like the REFERENCE's checkpoint (a pickled Config object next to 'state_dict',
    trainer.py:304-313) is refused by the safe loader with a message that says what to do; nan_policy='raise' is the
    reference's _check_nan; the default policy counts the skipped batches in the epoch line."""
    import argparse
    from mi_oov import driver
    monkeypatch.chdir(tmp_path)
    root = write_dataset(str(tmp_path))
    common = ["x", "--dataset=toy", f"--data_path={root}", "--model=BPR", "--embedding_size=32", "--user_oov_buckets=8",
              "--item_oov_buckets=8", "--inductive_embedder=lsh", "--add_oov_buckets", "--train_oov", "--train_batch_size=512"]
    ck = str(tmp_path / "foreign.pth")
    torch.save({"config": argparse.Namespace(a=1), "state_dict": {}}, ck)  # a non-tensor object, as the reference pickles
    with pytest.raises(RuntimeError, match="re-export"):
        driver.run(driver.custom_parse_args(common + ["--eval_only", f"--load_checkpoint={ck}"]))
    # a loss that is NaN (what an all-zero lsh code does to its batch): the reference aborts (_check_nan), and so does
    # nan_policy=raise; the default policy skips and counts, and refuses a phase in which every batch was skipped
    from mi_oov import model as model_mod
    real = model_mod.BPR.calculate_loss
    monkeypatch.setattr(model_mod.BPR, "calculate_loss", lambda self, batch: real(self, batch) * float("nan"))
    with pytest.raises(ValueError, match="Training loss is nan"):
        driver.run(driver.custom_parse_args(common + ["--epochs=1", "--nan_policy=raise"]))
    with pytest.raises(ValueError, match="trained nothing"):
        driver.run(driver.custom_parse_args(common + ["--epochs=1"]))


def test_inductive_dataset_pair_matches_reference():
    """The on-disk step either side of the path: a transductive dataset and its `X_ind` twin
    (`benchmark_filename = [train, empty, test_filt]`, `is_new` column) loaded by driver.AtomicDataset, and
    `remap_features` onto the transductive numbering, against what the REAL reference's Dataset / InductiveDataset
    produce for the same files (tests/golden/make_golden_ind.py; R/data/dataset/inductive_dataset.py:73-190)."""
    import numpy as np
    from mi_oov import driver
    here = os.path.dirname(os.path.abspath(__file__))
    z = np.load(os.path.join(here, "golden", "ind_dataset.npz"))
    root = os.path.join(here, "golden", "ind_dataset")
    tr = driver.AtomicDataset("ml-100k_tr", root)
    ind = driver.AtomicDataset("ml-100k_ind", root, benchmark_filename=["train", "empty", "test_filt"])
    assert [tr.user_num, tr.item_num] == z["tr__nums"].tolist()
    assert [ind.user_num, ind.item_num] == z["ind__nums"].tolist()
    assert (ind.n_train_users, ind.n_train_items) == (tr.user_num, tr.item_num)   # train-first numbering
    for side, feat in (("user", tr.user_feat), ("item", tr.item_feat)):
        for col in feat.columns:
            assert np.array_equal(feat[col].numpy(), z[f"tr__{side}__{col}"]), (side, col)
    for side, feat in (("user", ind.user_feat), ("item", ind.item_feat)):       # before the remap: the twin's own numbering
        for col in feat.columns:
            assert np.array_equal(feat[col].numpy(), z[f"ind_raw__{side}__{col}"]), (side, col)
    missing = ind.remap_features(tr)
    assert "homemaker" in missing["occupation"] and not missing["gender"]
    for side, feat, tfeat in (("user", ind.user_feat, tr.user_feat), ("item", ind.item_feat, tr.item_feat)):
        for col in feat.columns:
            got = feat[col].numpy()
            assert np.array_equal(got, z[f"ind__{side}__{col}"]), (side, col)
            k = tfeat[col].shape[0]
            assert np.array_equal(got[1:k], tfeat[col].numpy()[1:])               # S/perform_hashing.py:112-138
    for k, part in enumerate(("train", "empty", "test_filt")):
        sel = ind.split == k
        assert np.array_equal(ind.inter_user[sel], z[f"ind__inter__{part}__user"])
        assert np.array_equal(ind.inter_item[sel], z[f"ind__inter__{part}__item"])
    new_tok = z["ind__is_new_tokens"].tolist().index("1")
    assert np.array_equal(ind.is_new[ind.split == 2], z["ind__inter__test_filt__is_new"] == new_tok)
    # every row flagged new touches an out-of-vocabulary user or item; the flagged-old ones do not
    t = ind.split == 2
    oov = (ind.inter_user[t] >= ind.n_train_users) | (ind.inter_item[t] >= ind.n_train_items)
    assert np.array_equal(oov, ind.is_new[t])


@pytest.mark.gpu
def test_train_transductive_then_evaluate_inductive_twin(tmp_path, monkeypatch, dev):
    """The paper driver's two stages on the golden dataset pair (S/run_recbole.py:202-266 -> S/perform_hashing.py:85-170):
    train BPR + lsh on the transductive dataset, save; load the checkpoint against the `_ind` twin with its features
    remapped onto the transductive numbering (`--orig_dataset`) and evaluate the old / new slices."""
    from mi_oov import driver
    monkeypatch.chdir(tmp_path)
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ind_dataset")
    ck = str(tmp_path / "tr.pth")
    common = ["x", f"--data_path={root}", "--model=BPR", "--embedding_size=64", "--user_oov_buckets=8", "--item_oov_buckets=8",
              "--inductive_embedder=lsh", "--add_oov_buckets", "--train_oov", "--train_batch_size=1024"]
    _, m1 = driver.run(driver.custom_parse_args(common + ["--dataset=ml-100k_tr", "--oov_fraction=0", "--epochs=3",
                                                          f"--save_checkpoint={ck}"]))
    res, m2 = driver.run(driver.custom_parse_args(common + ["--dataset=ml-100k_ind", "--benchmark_filename=train,empty,test_filt",
                                                            "--orig_dataset=ml-100k_tr", "--eval_only", f"--load_checkpoint={ck}"]))
    assert m2.user_embedding.weight.shape == m1.user_embedding.weight.shape    # the twin's train part = the transductive vocabulary
    assert torch.equal(m1.item_embedding.weight, m2.item_embedding.weight)
    # the plugin hashes the old entities of the twin exactly as it did in training: same feature rows
    k = m1.inductive_embedder.item_feature_mat.shape[0]
    assert torch.equal(m1.inductive_embedder.item_feature_mat[1:], m2.inductive_embedder.item_feature_mat[1:k])
    assert {"overall", "old_users", "new_users", "new_items"} <= set(res)
    assert all(0 <= v <= 1 for v in res["overall"].values())
