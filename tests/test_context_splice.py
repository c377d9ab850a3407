"""Context-model OOV splice: golden from the reference's DCNV2 on ml-100k
(tests/golden/make_golden_context.py).  CPU: oracle vs reference.  GPU: kernel vs oracle (bit-exact)
and the host helper with an lsh embedder / the random mapper vs the reference's outputs."""
import numpy as np
import pytest
import torch

from conftest import bits_equal

PRIME_PAD = 112062759511


def _oov_rows_oracle(z, p, oracle, side, first_order):
    col = 0 if side == "user" else 1
    n_vocab = int(z[p + ("n_users" if side == "user" else "n_items")])
    ids = z[p + "tokens"][:, col]
    oov_ids = ids[ids >= n_vocab]
    fo = "fo_" if first_order else ""
    buckets = z[p + fo + side + "_buckets"]
    if p == "mapper_":
        mapped = oracle.mapper_map(oov_ids, "3round", n_vocab, buckets.shape[0])
        return oracle.gather_rows(mapped - n_vocab, buckets)
    planes = z["lsh_" + fo + side + "_planes"]
    return oracle.lsh_embed(oov_ids, z["lsh_" + side + "_feat"], planes, buckets)


@pytest.mark.parametrize("kind", ["lsh", "mapper"])
def test_oracle_matches_reference(kind, golden, oracle):
    z = golden("context_splice.npz")
    p = kind + "_"
    n_users, n_items = int(z[p + "n_users"]), int(z[p + "n_items"])
    for first_order in (False, True):
        table = z[p + ("fo_table" if first_order else "table")]
        ru = _oov_rows_oracle(z, p, oracle, "user", first_order)
        ri = _oov_rows_oracle(z, p, oracle, "item", first_order)
        got = oracle.token_fields_embed(z[p + "tokens"], z[p + "offsets"], table, n_users, n_items, ru, ri,
                                        sum_fields=first_order)
        ref = z[p + ("first" if first_order else "second")]
        ref = ref.reshape(got.shape)
        assert np.allclose(got, ref, rtol=1e-5, atol=1e-6, equal_nan=True)  # all-zero lsh codes are NaN on both sides
        if not first_order:
            iv = z[p + "tokens"][:, 0] < n_users  # in-vocabulary user rows are verbatim table rows
            assert bits_equal(got[iv, 0], ref[iv, 0]) and bits_equal(got[:, 2:], ref[:, 2:])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["lsh", "mapper"])
def test_gpu_splice_matches_reference(kind, golden, oracle, dev):
    import mi_oov
    from mi_oov import context
    z = golden("context_splice.npz")
    p = kind + "_"
    T = lambda k: torch.from_numpy(z[k]).to(dev)  # noqa: E731
    n_users, n_items = int(z[p + "n_users"]), int(z[p + "n_items"])
    tokens = T(p + "tokens")

    class Side(torch.nn.Module):  # what the embedder reads from `model`: the OOV bucket tables
        def __init__(self, ub, ib):
            super().__init__()
            self.user_oov_buckets = torch.nn.Embedding.from_pretrained(ub)
            self.item_oov_buckets = torch.nn.Embedding.from_pretrained(ib)

    for first_order in (False, True):
        fo = "fo_" if first_order else ""
        D = 1 if first_order else 16
        ub, ib = T(p + fo + "user_buckets"), T(p + fo + "item_buckets")
        model = Side(ub, ib)
        mapper = embedder = None
        if kind == "mapper":
            ft = mi_oov.FeatureTable({"id": torch.arange(4)})
            mapper = mi_oov.RandomOOVInductiveMapper(ft, ft, n_users, n_items, 8, 8, D, dev, PRIME_PAD, "3round")
        else:
            ft_u = mi_oov.FeatureTable({"id": torch.arange(z["lsh_user_feat"].shape[0]), "f": torch.from_numpy(z["lsh_user_feat"])})
            ft_i = mi_oov.FeatureTable({"id": torch.arange(z["lsh_item_feat"].shape[0]), "f": torch.from_numpy(z["lsh_item_feat"])})
            embedder = mi_oov.LSHInductiveEmbedder(ft_u, ft_i, n_users, n_items, 8, 8, D, dev, PRIME_PAD, "none",
                                                   mi_oov.InductiveFeatureCache())
            embedder.load_state_dict({"user_lsh.uniform_planes.0": T("lsh_" + fo + "user_planes"),
                                      "item_lsh.uniform_planes.0": T("lsh_" + fo + "item_planes")})
        table = T(p + ("fo_table" if first_order else "table"))
        with torch.no_grad():
            got = context.embed_token_fields(tokens, table, z[p + "offsets"], n_users, n_items, model, mapper, embedder,
                                             ub, ib, sum_fields=first_order).cpu().numpy()
        ref = z[p + ("first" if first_order else "second")]
        assert got.shape == ref.shape
        assert np.allclose(got, ref, rtol=1e-5, atol=1e-6, equal_nan=True)  # all-zero lsh codes are NaN on both sides
        # kernel vs oracle on identical OOV rows: bit-exact
        ru = _oov_rows_oracle(z, p, oracle, "user", first_order)
        ri = _oov_rows_oracle(z, p, oracle, "item", first_order)
        want = oracle.token_fields_embed(z[p + "tokens"], z[p + "offsets"], z[p + ("fo_table" if first_order else "table")],
                                         n_users, n_items, ru, ri, sum_fields=first_order)
        assert bits_equal(got.reshape(want.shape), want)


@pytest.mark.gpu
def test_gpu_splice_shapes(oracle, dev):
    from mi_oov import _cabi as C
    rng = np.random.default_rng(0)
    for B, nf, D in ((1, 2, 4), (333, 5, 64), (1000, 7, 10), (257, 3, 1), (64, 2, 130)):
        dims = [50, 70] + [int(x) for x in rng.integers(2, 30, size=nf - 2)]
        offsets = np.concatenate([[0], np.cumsum(dims)[:-1]]).astype(np.int64)
        table = rng.standard_normal((sum(dims), D), dtype=np.float32)
        tokens = np.stack([rng.integers(0, 80, B), rng.integers(0, 100, B)] + [rng.integers(0, d, B) for d in dims[2:]], 1)
        n_users, n_items = 40, 60
        ru = rng.standard_normal((int((tokens[:, 0] >= n_users).sum()), D), dtype=np.float32)
        ri = rng.standard_normal((int((tokens[:, 1] >= n_items).sum()), D), dtype=np.float32)
        for sum_fields in (False, True):
            want = oracle.token_fields_embed(tokens, offsets, table, n_users, n_items, ru, ri, sum_fields)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
            tk = t(tokens.astype(np.int64))
            ou, oi = tk[:, 0] >= n_users, tk[:, 1] >= n_items
            ranku = (torch.cumsum(ou, 0) - ou.long()).contiguous()
            ranki = (torch.cumsum(oi, 0) - oi.long()).contiguous()
            out = torch.empty(want.shape, dtype=torch.float32, device=dev)
            tru, tri, tt, to = t(ru), t(ri), t(table), t(offsets)
            rc = C.lib().mi_oov_token_fields_embed(tk.data_ptr(), B, nf, to.data_ptr(), tt.data_ptr(), table.shape[0], D,
                                                   n_users, n_items, tru.data_ptr() if len(ru) else None,
                                                   ranku.data_ptr(), len(ru), tri.data_ptr() if len(ri) else None,
                                                   ranki.data_ptr(), len(ri), int(sum_fields), out.data_ptr(),
                                                   C.stream_of(tk))
            assert rc == 0
            assert bits_equal(out.cpu().numpy(), want), (B, nf, D, sum_fields)


# ---- BASELINE config 5: knn / mean embedders reading their rows out of the fused token table ------------------------
# (knn_embedder.py:117-123,135-144, mean_embedder.py:53-60,75-86; fixture: tests/golden/make_golden_context.py knn_mean())
def _slices(offsets, table):
    """user / item row windows of the fused table exactly as the reference cuts them."""
    user = table[offsets[0]:offsets[1]]
    item = table[offsets[1]:] if len(offsets) == 2 else table[offsets[1]:offsets[2]]
    return user, item


def _km_oov_rows_oracle(z, p, oracle, first_order):
    table = z[p + ("fo_table" if first_order else "table")]
    uw, iw = _slices(z[p + "offsets"], table)
    n_users, n_items = int(z[p + "n_users"]), int(z[p + "n_items"])
    tok = z[p + "tokens"]
    nu, ni = int((tok[:, 0] >= n_users).sum()), int((tok[:, 1] >= n_items).sum())
    if p.startswith("knn"):
        return oracle.gather_mean(z["knn_user_idx"], uw, 2), oracle.gather_mean(z["knn_item_idx"], iw, 2)
    D = table.shape[1]
    return oracle.broadcast_rows(oracle.col_mean(uw), nu, D), oracle.broadcast_rows(oracle.col_mean(iw), ni, D)


@pytest.mark.parametrize("p", ["knn_", "mean_", "mean2_"])
def test_oracle_matches_reference_knn_mean(p, golden, oracle):
    z = golden("context_knn_mean.npz")
    assert (len(z[p + "offsets"]) == 2) == (p == "mean2_")
    n_users, n_items = int(z[p + "n_users"]), int(z[p + "n_items"])
    for first_order in (False, True):
        ru, ri = _km_oov_rows_oracle(z, p, oracle, first_order)
        table = z[p + ("fo_table" if first_order else "table")]
        got = oracle.token_fields_embed(z[p + "tokens"], z[p + "offsets"], table, n_users, n_items, ru, ri,
                                        sum_fields=first_order)
        ref = z[p + ("first" if first_order else "second")].reshape(got.shape)
        assert np.allclose(got, ref, rtol=1e-5, atol=1e-6)


class _Table(torch.nn.Module):
    def __init__(self, w):
        super().__init__()
        self.embedding = torch.nn.Embedding.from_pretrained(w)


class _CtxModel(torch.nn.Module):
    """What the knn / mean embedders read from a context model: the fused table and the field offsets."""

    def __init__(self, w, offsets):
        super().__init__()
        self.token_embedding_table = _Table(w)
        self.token_field_offsets = [int(o) for o in offsets]


@pytest.mark.gpu
@pytest.mark.parametrize("p", ["knn_", "mean_", "mean2_"])
def test_gpu_knn_mean_through_token_table(p, golden, oracle, dev):
    import mi_oov
    from mi_oov import context
    z = golden("context_knn_mean.npz")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    n_users, n_items = int(z[p + "n_users"]), int(z[p + "n_items"])
    tokens = T(z[p + "tokens"])
    for first_order in (False, True):
        D = 1 if first_order else 16
        table = T(z[p + ("fo_table" if first_order else "table")])
        model = _CtxModel(table, z[p + "offsets"])
        if p == "knn_":
            ft_u = mi_oov.FeatureTable({"id": torch.arange(z["knn_user_feat"].shape[0]), "f": torch.from_numpy(z["knn_user_feat"])})
            ft_i = mi_oov.FeatureTable({"id": torch.arange(z["knn_item_feat"].shape[0]), "f": torch.from_numpy(z["knn_item_feat"])})
            emb = mi_oov.KNNInductiveEmbedder(ft_u, ft_i, n_users, n_items, 8, 8, D, dev, PRIME_PAD, n_neighbors=2)
            emb.user_feature_mat, emb.item_feature_mat = T(z["knn_user_feat"]), T(z["knn_item_feat"])
            tok = z[p + "tokens"]
            uid, iid = tok[:, 0][tok[:, 0] >= n_users], tok[:, 1][tok[:, 1] >= n_items]
            # exact search here == the exact stand-in that produced the fixture (ScaNN itself: parity unpinned)
            # Items with identical feature rows tie exactly, and a tie may be broken either way: where the indices
            # differ the neighbours' inner products must be the same.
            same = {}
            for side, ids_s, hashfn in (("user", uid, emb._hash_users), ("item", iid, emb._hash_items)):
                idx = hashfn(T(ids_s)).cpu().numpy()
                fx, feat = z["knn_" + side + "_idx"], z["knn_" + side + "_feat"]
                same[side] = (idx == fx).all(1)
                q = feat[ids_s]
                s_mine = np.einsum("bf,bkf->bk", q, feat[idx])
                s_fix = np.einsum("bf,bkf->bk", q, feat[fx])
                assert np.allclose(np.sort(s_mine, 1), np.sort(s_fix, 1), rtol=0, atol=1e-6)
                assert same[side].mean() > 0.8
            same_u, same_i = same["user"], same["item"]
        else:
            ft = mi_oov.FeatureTable({"id": torch.arange(4)})
            emb = mi_oov.MeanEmbedder(ft, ft, n_users, n_items, 8, 8, D, dev)
        with torch.no_grad():
            got = context.embed_token_fields(tokens, table, z[p + "offsets"], n_users, n_items, model, None, emb,
                                             sum_fields=first_order).cpu().numpy()
        ref = z[p + ("first" if first_order else "second")]
        assert got.shape == ref.shape
        if p == "knn_":  # rows whose neighbours agree with the fixture's
            tok = z[p + "tokens"]
            ok = np.ones(len(tok), bool)
            ok[np.flatnonzero(tok[:, 0] >= n_users)[~same_u]] = False
            ok[np.flatnonzero(tok[:, 1] >= n_items)[~same_i]] = False
            assert np.allclose(got[ok], ref[ok], rtol=1e-5, atol=1e-6)
            ru, ri = _km_oov_rows_oracle(z, p, oracle, first_order)
            want = oracle.token_fields_embed(z[p + "tokens"], z[p + "offsets"], z[p + ("fo_table" if first_order else "table")],
                                             n_users, n_items, ru, ri, sum_fields=first_order)
            assert bits_equal(got.reshape(want.shape)[ok], want[ok])  # kernel vs oracle on identical neighbours: bit-exact
        else:
            assert np.allclose(got, ref, rtol=1e-5, atol=1e-6)
    # unknown model type -> ValueError, as the reference
    with pytest.raises(ValueError):
        emb.embed_user_ids(tokens[:3, 0].contiguous(), torch.nn.Linear(2, 2))


@pytest.mark.gpu
def test_gpu_config5_bf16_topk_at_catalogue_size(oracle, dev):
    """BASELINE config 5's scoring shape: Amazon-Books-sized catalogue (N = 31 094 items), D = 64, the fused
    bf16-prefilter top-k (exact f32 re-score) against the oracle's exact top-k: identical indices and values."""
    from mi_oov import ops
    rng = np.random.default_rng(31094)
    N, B, D, k = 31094, 512, 64, 10
    items = rng.standard_normal((N, D), dtype=np.float32)
    users = rng.standard_normal((B, D), dtype=np.float32)
    T = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    vals, idx = ops.score_topk(T(users), T(items), k)
    o_vals, o_idx = oracle.score_topk(users, items, k)
    assert np.array_equal(idx.cpu().numpy(), o_idx)
    assert bits_equal(vals.cpu().numpy(), o_vals)


# ---- the splice under autograd: gradients pinned on the reference's autograd (make_golden_context_grad.py) -----------
@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["lsh", "mapper"])
def test_gpu_splice_gradients_match_reference(kind, golden, dev):
    import mi_oov
    from mi_oov import context
    z = golden("context_grad.npz")
    p = kind + "_"
    T = lambda k: torch.from_numpy(z[k]).to(dev)  # noqa: E731
    n_users, n_items = int(z[p + "n_users"]), int(z[p + "n_items"])
    tokens = T(p + "tokens")

    class Side(torch.nn.Module):
        def __init__(self, ub, ib):
            super().__init__()
            self.user_oov_buckets = torch.nn.Embedding.from_pretrained(ub, freeze=False)
            self.item_oov_buckets = torch.nn.Embedding.from_pretrained(ib, freeze=False)

    loss = 0.0
    params = {}
    for first_order in (False, True):
        fo = "fo_" if first_order else ""
        D = 1 if first_order else 16
        model = Side(T(p + fo + "user_buckets"), T(p + fo + "item_buckets"))
        table = torch.nn.Parameter(T(p + fo + "table"))
        mapper = embedder = None
        if kind == "mapper":
            ft = mi_oov.FeatureTable({"id": torch.arange(4)})
            mapper = mi_oov.RandomOOVInductiveMapper(ft, ft, n_users, n_items, 8, 8, D, dev, PRIME_PAD, "3round")
        else:
            ft_u = mi_oov.FeatureTable({"id": torch.arange(z["lsh_user_feat"].shape[0]), "f": torch.from_numpy(z["lsh_user_feat"])})
            ft_i = mi_oov.FeatureTable({"id": torch.arange(z["lsh_item_feat"].shape[0]), "f": torch.from_numpy(z["lsh_item_feat"])})
            embedder = mi_oov.LSHInductiveEmbedder(ft_u, ft_i, n_users, n_items, 8, 8, D, dev, PRIME_PAD, "none",
                                                   mi_oov.InductiveFeatureCache())
            embedder.load_state_dict({"user_lsh.uniform_planes.0": T("lsh_" + fo + "user_planes"),
                                      "item_lsh.uniform_planes.0": T("lsh_" + fo + "item_planes")})
        got = context.embed_token_fields(tokens, table, z[p + "offsets"], n_users, n_items, model, mapper, embedder,
                                         model.user_oov_buckets.weight, model.item_oov_buckets.weight,
                                         sum_fields=first_order)
        assert got.requires_grad
        ref = z[p + ("first" if first_order else "second")]
        assert np.allclose(got.detach().cpu().numpy(), ref, rtol=1e-5, atol=1e-6, equal_nan=True)
        loss = loss + (got * T(p + ("w1" if first_order else "w2"))).sum()
        params[fo] = (table, model)
    assert abs(float(loss) - float(z[p + "loss"])) <= 1e-4 * max(1.0, abs(float(z[p + "loss"])))
    loss.backward()
    for fo, (table, model) in params.items():
        for name, t in (("table", table), ("user_buckets", model.user_oov_buckets.weight),
                        ("item_buckets", model.item_oov_buckets.weight)):
            want = z[p + "g_" + fo + name]
            got = t.grad.cpu().numpy()
            assert got.shape == want.shape, (fo, name)
            scale = max(1e-30, float(np.abs(want).max()))
            assert float(np.abs(got - want).max()) <= 1e-5 * scale, (fo, name)  # north_star: 1e-5 relative


@pytest.mark.gpu
def test_gpu_splice_forward_only_when_nothing_needs_grad(golden, dev):
    """No autograd node when no input requires a gradient (the inference path stays one launch)."""
    import mi_oov
    from mi_oov import context
    z = golden("context_grad.npz")
    T = lambda k: torch.from_numpy(z[k]).to(dev)  # noqa: E731
    ft = mi_oov.FeatureTable({"id": torch.arange(4)})
    n_users, n_items = int(z["mapper_n_users"]), int(z["mapper_n_items"])
    mapper = mi_oov.RandomOOVInductiveMapper(ft, ft, n_users, n_items, 8, 8, 16, dev, PRIME_PAD, "3round")
    out = context.embed_token_fields(T("mapper_tokens"), T("mapper_table"), z["mapper_offsets"], n_users, n_items, None, mapper,
                                     None, T("mapper_user_buckets"), T("mapper_item_buckets"))
    assert not out.requires_grad and np.allclose(out.cpu().numpy(), z["mapper_second"], rtol=1e-5, atol=1e-6)
