"""CPU: the host-side mirror of the reference's plugin surface (no kernel is launched).
Construction, registration strings, checkpoint key names, mode toggles, error behaviour and the
constructor-time feature build, checked against what the real reference produced (golden)."""
import copy
import json

import numpy as np
import pytest
import torch

import mi_oov
from mi_oov import embedders as E
from mi_oov import factory, mapper, model

PRIME_PAD = 112062759511


class Cfg(dict):
    def __getitem__(self, k):  # recbole Config returns None for missing keys (configurator.py:583-584)
        return self.get(k, None)


class DS:
    def __init__(self, uf, itf, n_users, n_items):
        self.uf, self.itf, self.user_num, self.item_num = uf, itf, n_users, n_items

    def get_user_feature(self):
        return self.uf

    def get_item_feature(self):
        return self.itf

    def num(self, field):
        return {"user_id": self.user_num, "item_id": self.item_num}[field]


def feats(n, widths, seed, id_name):
    g = torch.Generator().manual_seed(seed)
    cols = {id_name: torch.arange(n)}
    for i, w in enumerate(widths):
        cols[f"c{i}"] = torch.randn((n,) if w == 1 else (n, w), generator=g)
    return mi_oov.FeatureTable(cols)


def base_cfg(**kw):
    c = Cfg(embedding_size=16, device="cpu", user_oov_buckets=8, item_oov_buckets=8, oov_prime_pad=PRIME_PAD,
            oov_hash_function="3round", oov_normalization_type="per-feature", dhe_num_hashes=4, dhe_layer_size=32,
            oov_knn_num_neighbors=2, USER_ID_FIELD="user_id", ITEM_ID_FIELD="item_id", NEG_PREFIX="neg_",
            add_oov_buckets=True, oov_freeze_embedding=False)
    c.update(kw)
    return c


@pytest.fixture()
def ds():
    return DS(feats(30, [1, 3], 1, "user_id"), feats(40, [1, 5, 1], 2, "item_id"), 20, 25)


def test_factory_strings(ds, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)  # dhe/fdhe write ./hash_keys/{K}.hashes relative to the CWD
    expect = {"knn": E.KNNInductiveEmbedder, "lsh": E.LSHInductiveEmbedder, "slsh": E.SingleLSHInductiveEmbedder,
              "dhe": E.DeepHashEmbedder, "fdhe": E.FeatDeepHashEmbedder, "dnn": E.DNNEmbedder,
              "mean": E.MeanEmbedder, "zero": E.ZeroEmbedder}
    assert set(expect) == set(factory.EMBEDDERS)
    for name, cls in expect.items():
        emb = factory.get_inductive_embedder(base_cfg(inductive_embedder=name), ds)
        assert type(emb) is cls
        assert emb.n_new_users == 30 and emb.n_new_items == 40 and emb.training is False
        assert emb.n_original_users == 20 and emb.n_original_items == 25
    assert factory.get_inductive_embedder(base_cfg(inductive_embedder="nope"), ds) is None
    assert factory.get_inductive_embedder(base_cfg(), ds) is None
    m = factory.get_inductive_mapper(base_cfg(inductive_mapper="random"), ds)
    assert type(m) is mapper.RandomOOVInductiveMapper and m.hash_function == "3round"
    assert factory.get_inductive_mapper(base_cfg(inductive_mapper="other"), ds) is None
    # user_num / item_num override the dataset's (perform_inductive_eval passes the ORIGINAL sizes)
    emb = factory.get_inductive_embedder(base_cfg(inductive_embedder="slsh"), ds, user_num=7, item_num=9)
    assert (emb.n_original_users, emb.n_original_items) == (7, 9)


def test_feature_cache_shared_and_reset(ds):
    a = factory.get_inductive_embedder(base_cfg(inductive_embedder="lsh"), ds, mode="transductive")
    b = factory.get_inductive_embedder(base_cfg(inductive_embedder="lsh"), ds, mode="transductive", embedding_size=1)
    assert a.user_feature_mat is b.user_feature_mat  # main + first-order embedder share matrices
    c = factory.get_inductive_embedder(base_cfg(inductive_embedder="lsh"), ds, mode="inductive")
    assert c.user_feature_mat is not a.user_feature_mat  # mode change resets the module-global cache


@pytest.mark.parametrize("case,norm", [("mixed", "per-feature"), ("global", "global")])
def test_feature_build_matches_reference(case, norm, golden):
    z = golden(f"lsh_{case}.npz")
    uf = mi_oov.FeatureTable({c: torch.from_numpy(z["ucol_" + c]) for c in z["ucols"]})
    itf = mi_oov.FeatureTable({c: torch.from_numpy(z["icol_" + c]) for c in z["icols"]})
    emb = E.LSHInductiveEmbedder(uf, itf, 10, 10, z["user_planes"].shape[0], z["item_planes"].shape[0], 8, "cpu",
                                 PRIME_PAD, norm, E.InductiveFeatureCache())
    assert np.array_equal(emb.user_feature_mat.numpy(), z["user_feat"])
    assert np.array_equal(emb.item_feature_mat.numpy(), z["item_feat"])
    assert emb.user_lsh.uniform_planes[0].shape == z["user_planes"].shape  # one plane per bucket
    s = E.SingleLSHInductiveEmbedder(uf, itf, 10, 10, 8, 1000, 8, "cpu", PRIME_PAD, "per-feature")
    assert (s.user_bits_req, s.item_bits_req) == (3, 10)
    assert s.item_lsh.uniform_planes[0].shape[0] == 10
    with pytest.raises(ValueError, match="Invalid normalization type"):
        E.LSHInductiveEmbedder(uf, itf, 10, 10, 8, 8, 8, "cpu", PRIME_PAD, "bogus", E.InductiveFeatureCache())
    with pytest.raises(ValueError, match="Invalid normalization type"):
        E.SingleLSHInductiveEmbedder(uf, itf, 10, 10, 8, 8, 8, "cpu", PRIME_PAD, "bogus")


def test_checkpoint_key_names(ds, golden, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    lsh = factory.get_inductive_embedder(base_cfg(inductive_embedder="lsh"), ds)
    bpr = model.BPR(base_cfg(), ds, None, lsh)
    # exactly the keys the reference's BPR+lsh state_dict has (tests/golden/make_golden.py prints them)
    assert list(bpr.state_dict().keys()) == [
        "inductive_embedder.user_lsh.uniform_planes.0", "inductive_embedder.item_lsh.uniform_planes.0",
        "user_oov_buckets.weight", "item_oov_buckets.weight", "user_embedding.weight", "item_embedding.weight"]
    dhe = factory.get_inductive_embedder(base_cfg(inductive_embedder="dhe"), ds)
    keys = list(dhe.state_dict().keys())
    assert keys == [f"{s}_hash_net.{i}.{p}" for s in ("user", "item") for i in (0, 2, 4, 6) for p in ("weight", "bias")]
    assert dhe.state_dict()["item_hash_net.0.weight"].shape == (512, 4)  # hidden width fixed at 512
    z = golden("dhe.npz")
    ref_keys = sorted(k.replace("__", ".") for k in z.files if k.startswith("item_hash_net"))
    assert ref_keys == sorted(k for k in keys if k.startswith("item_hash_net"))
    fd = factory.get_inductive_embedder(base_cfg(inductive_embedder="fdhe"), ds)
    assert fd.state_dict()["user_hash_net.0.weight"].shape == (32, 4 + 4)  # K + F_user, hidden = dhe_layer_size
    # xavier re-init touches bucket tables and MLPs but never the planes (bpr.py:46)
    planes_before = lsh.item_lsh.uniform_planes[0].detach().clone()
    model.BPR(base_cfg(), ds, None, lsh)
    assert torch.equal(planes_before, lsh.item_lsh.uniform_planes[0])


def test_hash_key_file_protocol(ds, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    a = factory.get_inductive_embedder(base_cfg(inductive_embedder="dhe"), ds)
    path = tmp_path / "hash_keys" / "4.hashes"
    assert path.exists()
    on_disk = json.load(open(path))
    assert [k.hex() for k in a.hash_keys] == on_disk and all(len(k) == 16 for k in a.hash_keys)
    b = factory.get_inductive_embedder(base_cfg(inductive_embedder="fdhe"), ds)  # same K -> same keys reused
    assert b.hash_keys == a.hash_keys


def test_mode_toggles_and_mapper_bookkeeping(ds, golden):
    g = golden("mapper.json")
    m = mapper.RandomOOVInductiveMapper([0] * 20, [0] * 30, 15, 15, 8, 8, 64, "cpu", PRIME_PAD, "fast")
    assert (m.n_new_users, m.n_new_items) == (20, 30)
    m.set_train()
    assert [m.n_new_users, m.n_new_items] == g["train_n_new"] and m.training
    m.set_eval()
    assert [m.n_new_users, m.n_new_items] == g["eval_n_new"] and not m.training
    lsh = factory.get_inductive_embedder(base_cfg(inductive_embedder="lsh"), ds)
    mp = factory.get_inductive_mapper(base_cfg(inductive_mapper="random"), ds)
    bpr = model.BPR(base_cfg(oov_freeze_embedding=True), ds, mp, lsh)
    bpr.set_oov_train()
    assert lsh.training and mp.training and bpr.oov_training and not bpr.user_embedding.weight.requires_grad
    bpr.set_oov_eval()
    assert not lsh.training and not mp.training and bpr.user_embedding.weight.requires_grad
    bpr.set_oov_train(no_freeze=True)
    assert bpr.item_embedding.weight.requires_grad


def test_error_behaviour(ds):
    with pytest.raises(NotImplementedError, match="Must provide either"):
        model.BPR(base_cfg(), ds, None, None)
    bad = mapper.RandomOOVInductiveMapper([0] * 4, [0] * 4, 2, 2, 8, 8, 16, "cpu", PRIME_PAD, "murmur")
    with pytest.raises(ValueError, match="Unknown hash function murmur"):
        bad.map_item_ids(torch.arange(4))
    mean = factory.get_inductive_embedder(base_cfg(inductive_embedder="mean"), ds)
    with pytest.raises(ValueError, match="Invalid model type for mean embedder"):
        mean.embed_user_ids(torch.arange(3), object())
    knn = factory.get_inductive_embedder(base_cfg(inductive_embedder="knn"), ds)
    with pytest.raises(ValueError, match="Unknown model type"):
        E._general_tables(object())
    base = E.AbstractInductiveEmbedder([0] * 3, [0] * 4)
    for fn in (base.embed_user_ids, base.embed_item_ids):
        with pytest.raises(NotImplementedError):
            fn(torch.arange(2), None)
    with pytest.raises(NotImplementedError):
        knn.embed_all_items(None, None)


def test_no_cpu_fallback(ds):
    """The product path must fail loudly off-GPU instead of computing somewhere else."""
    lsh = factory.get_inductive_embedder(base_cfg(inductive_embedder="lsh"), ds)
    bpr = model.BPR(base_cfg(), ds, None, lsh)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lsh.embed_item_ids(torch.tensor([26, 27]), bpr)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        bpr.predict({"user_id": torch.tensor([1, 2]), "item_id": torch.tensor([3, 30])})
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mi_oov.ops.mapper_map(torch.arange(4), "fast", 2, 3)
    # round-2 entry points: queued batches, the two ends of the sharded exchange, the hash-net training pieces
    ids, rows = torch.zeros(4, dtype=torch.int64), torch.zeros(4, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mi_oov.ops.LshBatchQueue([ids], [rows])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mi_oov.ops.lsh_embed_score_multi([ids], torch.zeros(9, 64), torch.zeros(8, 64), torch.zeros(8, 64), [rows])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mi_oov.ops.bucket_by_owner(ids, 100, 50, 2, 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mi_oov.ops.lsh_codes_embed(torch.zeros(4, 8, dtype=torch.uint8), torch.zeros(4, dtype=torch.int32), torch.zeros(8, 64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mi_oov.ops.transpose(rows)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mi_oov.ops.hash_net_train(torch.nn.Sequential(torch.nn.Linear(64, 8), torch.nn.Sigmoid()), rows)
    # a sharded table's default local compute is the HIP kernels: no process group, no silent alternative either
    import torch.distributed as dist
    assert not dist.is_initialized()
    with pytest.raises((RuntimeError, ValueError)):
        mi_oov.sharded.ShardedLSHTable(torch.zeros(4, 64), 4)


def test_deepcopy_survives(ds, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    for name in factory.EMBEDDERS:  # get_flops deep-copies the whole model (R/utils/utils.py:269-271)
        emb = factory.get_inductive_embedder(base_cfg(inductive_embedder=name), ds)
        bpr = model.BPR(base_cfg(), ds, None, emb)
        clone = copy.deepcopy(bpr)
        assert type(clone.inductive_embedder) is type(emb)
        assert list(clone.state_dict().keys()) == list(bpr.state_dict().keys())
    knn = factory.get_inductive_embedder(base_cfg(inductive_embedder="knn", oov_knn_num_neighbors=5), ds)
    assert copy.deepcopy(knn).n_neighbors == 2  # the reference's __deepcopy__ drops n_neighbors (knn_embedder.py:95-98)


def test_torch_library_registration():
    """`torch.ops.mi_oov.*` (mi_oov/torch_ops.py): every op has a schema, the CPU key raises instead of computing, and
    the fake-tensor rule gives shapes / dtypes without a device."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    from mi_oov import torch_ops
    for name in torch_ops.OPS:
        assert hasattr(torch.ops.mi_oov, name), name
    ids = torch.zeros(3, dtype=torch.int64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        torch.ops.mi_oov.lsh_embed(ids, torch.zeros(5, 64), torch.zeros(8, 64), torch.zeros(8, 64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        torch.ops.mi_oov.mapper_map(ids, "3round", 2, 3)
    with FakeTensorMode():
        fids = torch.zeros(7, dtype=torch.int64, device="cuda")
        feat, planes = torch.zeros(5, 64, device="cuda"), torch.zeros(8, 64, device="cuda")
        out = torch.ops.mi_oov.lsh_embed(fids, feat, planes, torch.zeros(8, 32, device="cuda"))
        assert out.shape == (7, 32) and out.dtype == torch.float32
        assert torch.ops.mi_oov.lsh_bits(fids, feat, planes).dtype == torch.uint8
        vals, idx = torch.ops.mi_oov.score_topk(torch.zeros(4, 64, device="cuda"), torch.zeros(100, 64, device="cuda"), 5, 1)
        assert vals.shape == (4, 5) and idx.dtype == torch.int64
        many = torch.ops.mi_oov.lsh_embed_score_multi([fids, fids], feat, planes, torch.zeros(8, 64, device="cuda"),
                                                      [torch.zeros(7, 64, device="cuda")] * 2)
        assert len(many) == 2 and many[0].shape == (7,)


def test_split_layer_host_rules(monkeypatch):
    """Host-side rules of the split-bf16 layers (no GPU): inference takes them unless MI_OOV_LINEAR_X3=0; the share count of
    a training product gives every CU about two workgroups, never a share below 8 stages of 16 k, and leaves products
    with enough tiles alone."""
    import mi_oov
    ops = mi_oov.ops
    monkeypatch.delenv("MI_OOV_LINEAR_X3", raising=False)
    assert ops._x3_wanted()
    monkeypatch.setenv("MI_OOV_LINEAR_X3", "0")
    assert not ops._x3_wanted()
    monkeypatch.setenv("MI_OOV_LINEAR_X3", "1")
    assert ops._x3_wanted()
    assert ops._x3_ksplit(512, 1024, 2048) == 16      # dW of the first dhe layer at a 2048-row step: 32 tiles, 128 stages
    assert ops._x3_ksplit(512, 1024, 65536) == 16     # ... the batch 32 x larger: still 16 shares (two workgroups per CU)
    assert ops._x3_ksplit(1, 512, 2048) == 16         # db: 4 tiles
    assert ops._x3_ksplit(2048, 1024, 512) == 4       # dX: 128 tiles, 32 stages
    assert ops._x3_ksplit(65536, 512, 512) == 1 and ops._x3_ksplit(4, 4, 16) == 1
