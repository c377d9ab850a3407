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
