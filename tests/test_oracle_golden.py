"""CPU: the oracle (oracle/oov_oracle.c + oracle/ref_torch.py) pinned on vectors produced by the
REAL reference (tests/golden/make_golden.py) and on public SipHash-2-4 vectors.

Bars: integer / index / bit outputs identical; floats within 1e-5 relative (north_star)."""
import numpy as np
import pytest
import torch

from conftest import bits_equal

RTOL = 1e-5
LSH_CASES = ["f64", "mixed", "global", "wide"]


def rel_err(a, b):
    assert np.array_equal(np.isnan(a), np.isnan(b)), "NaN rows differ"
    m = ~np.isnan(b)
    return float(np.abs(a[m] - b[m]).max() / max(1e-30, np.abs(b[m]).max())) if m.any() else 0.0


def test_mapper_known_answers(golden, oracle):
    m = golden("mapper.json")
    ids = np.array(m["ids"], dtype=np.int64)
    for c in m["cases"]:
        out = oracle.mapper_map(ids, c["hash"], c["n_orig"], c["n_buckets"]).tolist()
        assert out == c["map_item"] == c["map_user"], (c["hash"], c["n_orig"])
    raw = np.array(m["raw_in"], dtype=np.int64)
    assert oracle.mapper_hash(raw, "fast").tolist() == m["raw_fast"]
    assert oracle.mapper_hash(raw, "3round").tolist() == m["raw_3round"]
    for nb in (8, 1000, 2 ** 31 + 11):
        assert oracle.mapper_map(raw, "64bit", 0, nb).tolist() == m[f"raw_64bit_mod_{nb}"]
    # values recorded independently in SURVEY.md section 8
    assert oracle.mapper_hash(np.array([1, 2, 3, 2 ** 40 + 7]), "3round").tolist() == [
        8831518008760208934, 4198225440614570042, 2815785134427733286, 4101270232430997791]
    assert oracle.mapper_hash(np.array([1, 2, 3, 2 ** 40 + 7]), "fast").tolist() == [
        2019227867398666867, 4038455734797333735, 6057398897650032378, 7589850910523895961]
    survey_ids = np.array([0, 5, 14, 15, 16, 17, 100, 12345678901, 112062759514])
    assert oracle.mapper_map(survey_ids, "3round", 15, 8).tolist() == [0, 5, 14, 15, 21, 17, 16, 19, 19]
    assert oracle.mapper_map(survey_ids, "mod", 15, 8).tolist() == [0, 5, 14, 15, 16, 17, 20, 21, 18]


def test_siphash_public_vectors(golden, oracle):
    s = golden("siphash.json")
    key = bytes(range(16))
    # SipHash paper, appendix A: key 00..0f, message 00..0e
    assert oracle.siphash24(key, bytes(range(15))) == 0xA129CA6149BE45E5
    assert oracle.siphash24(key, bytes(range(8))).to_bytes(8, "little").hex() == "6224939a79f5f593"
    for e in s["by_len"]:  # lengths 0..23 against the independent pure-Python implementation
        assert oracle.siphash24(key, bytes(range(e["len"]))).to_bytes(8, "little").hex() == e["hash_le_hex"]
    keys = np.frombuffer(b"".join(bytes.fromhex(k) for k in s["dhe_keys"]), dtype=np.uint8).reshape(-1, 16)
    hm = oracle.siphash24_mod(np.array(s["dhe_ids"]), keys)
    assert hm.astype(np.int64).tolist() == s["dhe_hashes"]  # reference DeepHashEmbedder._hash_ids
    k4 = np.stack([np.arange(j, j + 16, dtype=np.uint8) for j in range(4)])
    assert oracle.siphash24_mod(np.array([0, 1, 2, 112062759516]), k4).astype(np.int64).tolist() == s["survey_k4"]


@pytest.mark.parametrize("case", LSH_CASES)
@pytest.mark.parametrize("side", ["user", "item"])
def test_lsh_against_reference(case, side, golden, oracle):
    z = golden(f"lsh_{case}.npz")
    ids, feat, planes, buckets = z[side + "_ids"], z[side + "_feat"], z[side + "_planes"], z[side + "_buckets"]
    emb, bits = oracle.lsh_embed(ids, feat, planes, buckets, want_bits=True)
    marg = z[side + "_margin"]
    safe = (marg > 1e-5) | (marg == 0)  # |projection| clear of fp32 reordering noise, or exactly 0 -> bit 1
    assert safe.mean() > 0.99
    assert np.array_equal(bits[safe], z[side + "_bits"][safe])
    assert (bits != z[side + "_bits"]).sum() == 0  # in fact no bit differs anywhere in the fixtures
    assert rel_err(emb[safe], z[side + "_emb"][safe]) <= RTOL
    # quirks: padding row 0 -> all bits 1; all-zero code -> NaN row (lsh_embedder.py:178)
    assert bits[0].all()
    zero_code = ~bits.any(1)
    assert np.isnan(emb[zero_code]).all() and not np.isnan(emb[~zero_code]).any()
    # train mode: the reference strips prime_pad from the caller's tensor in place
    if bool(z["train"]):
        assert np.array_equal(z[side + "_ids_after"], ids)
        assert (z[side + "_ids_in"] >= 112062759511).any()


@pytest.mark.parametrize("case", ["b8", "b1000"])
@pytest.mark.parametrize("side", ["user", "item"])
def test_slsh_against_reference(case, side, golden, oracle):
    z = golden(f"slsh_{case}.npz")
    emb, idx = oracle.slsh_embed(z[side + "_ids"], z[side + "_feat"], z[side + "_planes"], z[side + "_buckets"])
    assert np.array_equal(idx, z[side + "_idx"])
    assert bits_equal(emb, z[side + "_emb"])
    if case == "b8":  # popcount quirk: only bits_req..2*bits_req are reachable
        assert set(idx.tolist()) <= {3, 4, 5, 6}


def test_knn_mean_against_reference(golden, oracle):
    z = golden("knn.npz")
    for side in ("user", "item"):
        got = oracle.gather_mean(z[side + "_idx"], z[side + "_table"], 2)
        assert rel_err(got, z[side + "_emb"]) <= RTOL
    z = golden("mean.npz")
    for side in ("user", "item"):
        mean = oracle.col_mean(z[side + "_table"])
        got = oracle.broadcast_rows(mean, len(z[side + "_ids"]), mean.shape[0])
        assert rel_err(got, z[side + "_emb"]) <= RTOL
    assert not z["zero_user_emb"].any()


def test_bpr_against_reference(golden, oracle):
    z = golden("bpr_lsh.npz")
    n_users, n_items = int(z["n_users"]), int(z["n_items"])
    for side, ids in (("user", z["users"]), ("item", z["items"])):
        got = oracle.lsh_lookup(ids, z[side + "_table"], z[side + "_feat"], z[side + "_planes"], z[side + "_buckets"])
        assert rel_err(got, z[side + "_e"]) <= RTOL
    score = oracle.rowdot(z["user_e"], z["item_e"])
    m = ~np.isnan(z["predict"])
    assert np.array_equal(np.isnan(score), ~m)
    assert np.allclose(score[m], z["predict"][m], rtol=RTOL, atol=1e-6)
    ue = z["user_e"][:40]
    ok = ~np.isnan(ue).any(1)
    fs = oracle.full_sort_scores(ue[ok], z["item_table"])
    assert np.allclose(fs, z["full_sort"].reshape(40, -1)[ok], rtol=RTOL, atol=1e-6)
    # mapper-only model: OOV ids -> random bucket rows (bpr.py:75,122)
    for side, n_vocab, ids in (("user", n_users, z["users"]), ("item", n_items, z["items"])):
        mapped = oracle.mapper_map(ids, "3round", n_vocab, 8)
        table = np.concatenate([z[f"m_{side}_table"], z[f"m_{side}_buckets"]])
        assert bits_equal(oracle.gather_rows(mapped, table), z[f"m_{side}_e"])
    assert np.allclose(oracle.rowdot(z["m_user_e"], z["m_item_e"]), z["m_predict"], rtol=RTOL, atol=1e-6)


def test_topk_semantics(oracle):
    U = np.array([[1.0, 0.0], [0.0, 1.0]], dtype=np.float32)
    E = np.array([[9, 9], [1, 5], [2, 5], [2, 1], [np.nan, 0]], dtype=np.float32)
    vals, idx = oracle.score_topk(U, E, 3, 1)  # column 0 skipped (padding item)
    # NaN sorts first (torch.topk); 0 * NaN is NaN too, so row 4 leads for both users; ties -> lower index
    assert idx.tolist() == [[4, 2, 3], [4, 1, 2]]
    assert np.isnan(vals[:, 0]).all() and vals[0, 1:].tolist() == [2.0, 2.0] and vals[1, 1:].tolist() == [5.0, 5.0]


@pytest.mark.parametrize("case", LSH_CASES)
def test_ref_torch_restatement(case, golden):
    """oracle/ref_torch.py is the op sequence bench.py times as cpu_baseline: check it too."""
    from oracle import ref_torch
    z = golden(f"lsh_{case}.npz")
    t = lambda k: torch.from_numpy(z[k])  # noqa: E731
    emb = ref_torch.lsh_embed(t("item_ids"), t("item_feat"), t("item_planes"), t("item_buckets")).numpy()
    safe = (z["item_margin"] > 1e-5) | (z["item_margin"] == 0)
    assert rel_err(emb[safe], z["item_emb"][safe]) <= RTOL


def test_lsh_backward_oracle_matches_autograd(oracle):
    """The oracle's bucket-table gradient against torch autograd run on the reference's own op sequence
    (`(bits @ W) / bits.sum(1)`, lsh_embedder.py:158,178)."""
    import torch
    rng = np.random.default_rng(5)
    for B, H, D in ((1, 8, 64), (777, 8, 64), (20000, 12, 32), (100, 3, 22)):
        bits = (rng.random((B, H)) < 0.5).astype(np.uint8)
        bits[bits.sum(1) == 0, 0] = 1
        g = rng.standard_normal((B, D)).astype(np.float32)
        W = torch.zeros((H, D), requires_grad=True)
        bt = torch.from_numpy(bits).float()
        ((bt @ W) / bt.sum(1, keepdim=True) * torch.from_numpy(g)).sum().backward()
        got = oracle.lsh_embed_backward(bits, g)
        assert np.abs(got - W.grad.numpy()).max() <= 1e-5 * np.abs(W.grad.numpy()).max()
    bits[3] = 0  # all-zero code: the reference's gradient is NaN everywhere (0 * inf)
    W = torch.zeros((H, D), requires_grad=True)
    bt = torch.from_numpy(bits).float()
    ((bt @ W) / bt.sum(1, keepdim=True) * torch.from_numpy(g)).sum().backward()
    assert torch.isnan(W.grad).all() and np.isnan(oracle.lsh_embed_backward(bits, g)).all()


def test_fdhe_hashes_of_unstripped_ids(golden, oracle):
    """'fdhe' (feat_dh_embedder.py:180-206): the hashes are SipHash-2-4 of the UN-stripped id (prime pad included) mod
    2^24 -- the oracle's restatement against the reference's own _hash_ids on the fixture's train-mode ids -- and the MLP
    input is those K hashes followed by the feature row of the STRIPPED id."""
    z = golden("fdhe.npz")
    K = int(z["dims"][0])
    for mode in ("eval", "train"):
        ids = z[f"ids_{mode}"]
        want = z[f"{mode}_user_hashes"]
        assert np.array_equal(oracle.siphash24_mod(ids, z["keys"]), want)
        assert np.array_equal(z[f"{mode}_item_hashes"], want)  # both sides hash the id with the same keys
        stripped = np.where(ids >= 112062759511, ids - 112062759511, ids)
        for side in ("user", "item"):
            x = z[f"{mode}_{side}_input"]
            assert np.array_equal(x[:, :K], want)
            assert np.array_equal(x[:, K:], z[f"{side}_feature_mat"][stripped])
    assert (z["ids_train"] >= 112062759511).sum() == 4
