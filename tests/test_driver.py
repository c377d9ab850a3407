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
])
def test_end_to_end(flags, tmp_path, monkeypatch, dev):
    from mi_oov import driver
    monkeypatch.chdir(tmp_path)  # dhe writes ./hash_keys
    root = write_dataset(str(tmp_path))
    base = [] if any(f.startswith("--model=") for f in flags) else ["--model=BPR"]
    args = driver.custom_parse_args(["x", "--dataset=toy", f"--data_path={root}", "--embedding_size=32",
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
