"""-m gpu: the HIP path (through the C ABI) against the oracle and the reference-generated golden
fixtures.  Integer/index/bit outputs must be identical; float outputs are BIT-EXACT against the
oracle (same canonical summation order) and within 1e-5 relative of the reference's own outputs
(BASELINE.json north_star tolerance)."""
import json

import numpy as np
import pytest
import torch

from conftest import bits_equal

pytestmark = pytest.mark.gpu
RTOL = 1e-5  # north_star: "aggregated embeddings and scores match within 1e-5 rel fp32"


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def rel_err(a, b):
    m = ~(np.isnan(a) | np.isnan(b))
    assert np.array_equal(np.isnan(a), np.isnan(b))
    return float(np.abs(a[m] - b[m]).max() / max(1e-30, np.abs(b[m]).max())) if m.any() else 0.0


@pytest.fixture(scope="module")
def ops():
    import mi_oov
    from mi_oov import ops as _ops
    assert mi_oov.available(), "libmi_oov.so missing on the GPU box"
    return _ops


@pytest.mark.parametrize("case", ["f64", "mixed", "global", "wide"])
@pytest.mark.parametrize("side", ["user", "item"])
def test_lsh_golden(case, side, golden, oracle, ops, dev):
    z = golden(f"lsh_{case}.npz")
    ids, feat, planes, buckets = z[side + "_ids"], z[side + "_feat"], z[side + "_planes"], z[side + "_buckets"]
    emb = ops.lsh_embed(T(ids, dev), T(feat, dev), T(planes, dev), T(buckets, dev)).cpu().numpy()
    bits = ops.lsh_bits(T(ids, dev), T(feat, dev), T(planes, dev)).cpu().numpy()
    o_emb, o_bits = oracle.lsh_embed(ids, feat, planes, buckets, want_bits=True)
    assert np.array_equal(bits, o_bits)
    assert bits_equal(emb, o_emb)
    # reference: bits on margin-safe rows (all rows of these fixtures agree), floats within 1e-5
    marg = z[side + "_margin"]
    safe = (marg > 1e-5) | (marg == 0)
    assert np.array_equal(bits[safe], z[side + "_bits"][safe])
    assert safe.mean() > 0.99
    assert rel_err(emb[safe], z[side + "_emb"][safe]) <= RTOL


@pytest.mark.parametrize("case", ["b8", "b1000"])
@pytest.mark.parametrize("side", ["user", "item"])
def test_slsh_golden(case, side, golden, oracle, ops, dev):
    z = golden(f"slsh_{case}.npz")
    ids, feat, planes, buckets = z[side + "_ids"], z[side + "_feat"], z[side + "_planes"], z[side + "_buckets"]
    emb = ops.slsh_embed(T(ids, dev), T(feat, dev), T(planes, dev), T(buckets, dev)).cpu().numpy()
    idx = ops.slsh_index(T(ids, dev), T(feat, dev), T(planes, dev), buckets.shape[0]).cpu().numpy()
    o_emb, o_idx = oracle.slsh_embed(ids, feat, planes, buckets)
    assert np.array_equal(idx, o_idx) and bits_equal(emb, o_emb)
    assert np.array_equal(idx, z[side + "_idx"])
    assert bits_equal(emb, z[side + "_emb"])


def test_mapper_golden(golden, ops, dev):
    m = golden("mapper.json")
    ids = T(np.array(m["ids"], dtype=np.int64), dev)
    for c in m["cases"]:
        out = ops.mapper_map(ids, c["hash"], c["n_orig"], c["n_buckets"]).cpu().tolist()
        assert out == c["map_item"], c["hash"]
    raw = T(np.array(m["raw_in"], dtype=np.int64), dev)
    assert ops.mapper_hash(raw, "fast").cpu().tolist() == m["raw_fast"]
    assert ops.mapper_hash(raw, "3round").cpu().tolist() == m["raw_3round"]
    for nb in (8, 1000, 2 ** 31 + 11):
        assert ops.mapper_map(raw, "64bit", 0, nb).cpu().tolist() == m[f"raw_64bit_mod_{nb}"]


def test_siphash_golden(golden, oracle, ops, dev):
    s = golden("siphash.json")
    keys = np.frombuffer(b"".join(bytes.fromhex(k) for k in s["dhe_keys"]), dtype=np.uint8).reshape(-1, 16).copy()
    out = ops.siphash24_mod(T(np.array(s["dhe_ids"], dtype=np.int64), dev), T(keys, dev)).cpu().numpy()
    assert out.astype(np.int64).tolist() == s["dhe_hashes"]
    k4 = np.stack([np.arange(j, j + 16, dtype=np.uint8) for j in range(4)])
    out = ops.siphash24_mod(T(np.array([0, 1, 2, 112062759516], dtype=np.int64), dev), T(k4, dev)).cpu().numpy()
    assert out.astype(np.int64).tolist() == s["survey_k4"]
    # bulk: random ids x 1024 keys against the oracle
    rng = np.random.default_rng(5)
    keys = rng.integers(0, 256, size=(1024, 16), dtype=np.uint8)
    ids = rng.integers(-2 ** 62, 2 ** 62, size=300, dtype=np.int64)
    got = ops.siphash24_mod(T(ids, dev), T(keys, dev)).cpu().numpy()
    assert np.array_equal(got, oracle.siphash24_mod(ids, keys))


def test_knn_mean_golden(golden, oracle, ops, dev):
    z = golden("knn.npz")
    for side in ("user", "item"):
        got = ops.gather_mean(T(z[side + "_idx"], dev), T(z[side + "_table"], dev), 2).cpu().numpy()
        assert bits_equal(got, oracle.gather_mean(z[side + "_idx"], z[side + "_table"], 2))
        assert rel_err(got, z[side + "_emb"]) <= RTOL
    z = golden("mean.npz")
    for side in ("user", "item"):
        W = z[side + "_table"]
        mean = ops.col_mean(T(W, dev))
        assert bits_equal(mean.cpu().numpy(), oracle.col_mean(W))
        got = ops.broadcast_rows(mean, len(z[side + "_ids"])).cpu().numpy()
        assert rel_err(got, z[side + "_emb"]) <= RTOL
    zero = ops.broadcast_rows(None, 7, 16, dev).cpu().numpy()
    assert zero.shape == (7, 16) and not zero.any()


def test_bpr_golden(golden, oracle, ops, dev):
    z = golden("bpr_lsh.npz")
    n_users, n_items = int(z["n_users"]), int(z["n_items"])
    for side, n_vocab, ids in (("user", n_users, z["users"]), ("item", n_items, z["items"])):
        args = [z[side + "_table"], z[side + "_feat"], z[side + "_planes"], z[side + "_buckets"]]
        got = ops.lsh_lookup(T(ids, dev), *[T(a, dev) for a in args]).cpu().numpy()
        assert bits_equal(got, oracle.lsh_lookup(ids, *args))
        assert rel_err(got, z[side + "_e"]) <= RTOL
    score = ops.rowdot(T(z["user_e"], dev), T(z["item_e"], dev)).cpu().numpy()
    assert bits_equal(score, oracle.rowdot(z["user_e"], z["item_e"]))
    m = ~np.isnan(z["predict"])
    assert np.allclose(score[m], z["predict"][m], rtol=RTOL, atol=1e-6)
    ue = z["user_e"][:40]
    ok = ~np.isnan(ue).any(1)
    fs = ops.full_sort_scores(T(ue[ok], dev), T(z["item_table"], dev)).cpu().numpy()
    assert bits_equal(fs, oracle.full_sort_scores(ue[ok], z["item_table"]))
    ref = z["full_sort"].reshape(40, -1)[ok]
    assert np.allclose(fs, ref, rtol=RTOL, atol=1e-6)


@pytest.mark.parametrize("B,N,F,H,D,nb", [(100, 50, 3000, 30, 16, 1000), (257, 99, 3000, 27, 64, 5), (64, 50, 5, 0, 7, 1), (100, 50, 64, 0, 64, 1),
                                          (80, 50, 1100, 40, 300, 77)])
def test_slsh_wide_rows_many_planes_and_no_planes_vs_oracle(B, N, F, H, D, nb, oracle, ops, dev):
    """slsh keeps only its planes in LDS: feature rows of thousands of floats times ~30 planes (bits_req of a very large
    bucket table) are staged a chunk of planes at a time (slsh_kernel<..., CHUNK>); n_buckets = 1 means bits_req = 0 planes
    (single_lsh_embedder.py:77-80) and every lookup in bucket 0."""
    rng = np.random.default_rng(B + F)
    feat = rng.standard_normal((N, F), dtype=np.float32)
    planes = rng.standard_normal((H, F), dtype=np.float32)
    big = rng.standard_normal((nb, D), dtype=np.float32)
    ids = rng.integers(0, N, size=B, dtype=np.int64)
    ids[3], ids[4] = N + 5, -1
    want, widx = oracle.slsh_embed(ids, feat, planes, big)
    got = ops.slsh_embed(T(ids, dev), T(feat, dev), T(planes, dev), T(big, dev)).cpu().numpy()
    assert bits_equal(got, want)
    assert np.array_equal(ops.slsh_index(T(ids, dev), T(feat, dev), T(planes, dev), nb).cpu().numpy(), widx)
    if H == 0:
        assert set(widx.tolist()) == {0, -1}


@pytest.mark.parametrize("B,N,F,H,D", [(1000, 700, 64, 100, 64), (777, 500, 64, 333, 64), (300, 200, 64, 1000, 64), (130, 99, 20, 1000, 32),
                                       (65, 50, 130, 700, 200), (4100, 900, 64, 129, 64), (50, 40, 64, 65, 128),
                                       # feature rows from an encoder (hundreds of floats) x many buckets: lsh_wide_kernel<..., CHUNK>
                                       (90, 50, 768, 1000, 64), (65, 50, 770, 300, 300), (70, 40, 2000, 30, 16),
                                       # embedding rows wider than 256 floats: one launch per window of 256 columns
                                       (100, 50, 64, 8, 300), (64, 50, 64, 8, 512), (70, 40, 22, 40, 1030), (33, 20, 64, 8, 257)])
def test_lsh_with_as_many_planes_as_oov_buckets_vs_oracle(B, N, F, H, D, oracle, ops, dev):
    """The lsh plugin has one hyperplane per OOV bucket (lsh_embedder.py:108-114): a model with hundreds or thousands of
    buckets has as many planes, more than the LDS holds at once.  Rows, codes, fused scores and the in-vocabulary splice
    against the oracle with the planes staged a chunk at a time (lsh_fused_kernel<..., CHUNK>): the bucket-row chain runs
    over all H planes in order whatever the chunking, so the bits are the oracle's."""
    rng = np.random.default_rng(B + H)
    feat = rng.standard_normal((N, F), dtype=np.float32)
    feat[0] = 0
    planes = rng.standard_normal((H, F), dtype=np.float32)
    buckets = rng.standard_normal((H, D), dtype=np.float32)
    other = rng.standard_normal((B, D), dtype=np.float32)
    table = rng.standard_normal((N // 2, D), dtype=np.float32)
    ids = rng.integers(0, N, size=B, dtype=np.int64)
    ids[0] = 0
    ids[3], ids[4] = N + 5, -1
    d = lambda a: T(a, dev)  # noqa: E731
    o_emb, o_bits = oracle.lsh_embed(ids, feat, planes, buckets, want_bits=True)
    assert bits_equal(ops.lsh_embed(d(ids), d(feat), d(planes), d(buckets)).cpu().numpy(), o_emb)
    assert np.array_equal(ops.lsh_bits(d(ids), d(feat), d(planes)).cpu().numpy(), o_bits)
    emb_b, bits_b = ops._lsh_forward(d(ids), d(feat), d(planes), d(buckets), want_bits=True)
    assert np.array_equal(bits_b.cpu().numpy(), o_bits) and bits_equal(emb_b.cpu().numpy(), o_emb)
    o_score, _ = oracle.lsh_embed_score(ids, feat, planes, buckets, other)
    assert bits_equal(ops.lsh_embed_score(d(ids), d(feat), d(planes), d(buckets), d(other)).cpu().numpy(), o_score)
    o_look = oracle.lsh_lookup(ids, table, feat, planes, buckets)
    assert bits_equal(ops.lsh_lookup(d(ids), d(table), d(feat), d(planes), d(buckets)).cpu().numpy(), o_look)
    assert bits_equal(ops.lsh_lookup_score(d(ids), d(table), d(feat), d(planes), d(buckets), d(other)).cpu().numpy(),
                      oracle.rowdot(other, o_look))
    # queued batches of a shape the persistent kernel does not take: K single launches, same rows
    multi = ops.lsh_embed_multi([d(ids), d(ids[::-1].copy())], d(feat), d(planes), d(buckets))
    assert bits_equal(multi[0].cpu().numpy(), o_emb) and bits_equal(multi[1].cpu().numpy(), o_emb[::-1])


@pytest.mark.parametrize("B,N,F,H,D", [(1, 7, 64, 8, 64), (63, 100, 64, 8, 64), (4097, 3000, 64, 8, 64), (16, 9, 64, 8, 64),
                                       (500, 400, 22, 8, 64), (333, 200, 4, 3, 1), (257, 150, 128, 16, 128),
                                       (4099, 3000, 64, 16, 64), (1000, 500, 64, 27, 64), (77, 60, 64, 32, 64),
                                       (100, 90, 200, 9, 36), (129, 77, 301, 40, 50), (64, 50, 640, 12, 256)])
def test_lsh_shapes_vs_oracle(B, N, F, H, D, oracle, ops, dev):
    rng = np.random.default_rng(B * 31 + F)
    feat = rng.standard_normal((N, F), dtype=np.float32)
    feat[0] = 0
    planes = rng.standard_normal((H, F), dtype=np.float32)
    buckets = rng.standard_normal((H, D), dtype=np.float32)
    other = rng.standard_normal((B, D), dtype=np.float32)
    ids = rng.integers(0, N, size=B, dtype=np.int64)
    ids[0] = 0
    if B > 5:
        ids[3] = N + 5   # out of range -> NaN row / 0xFF bits, never a fault
        ids[4] = -1
    emb = ops.lsh_embed(T(ids, dev), T(feat, dev), T(planes, dev), T(buckets, dev)).cpu().numpy()
    bits = ops.lsh_bits(T(ids, dev), T(feat, dev), T(planes, dev)).cpu().numpy()
    o_emb, o_bits = oracle.lsh_embed(ids, feat, planes, buckets, want_bits=True)
    assert np.array_equal(bits, o_bits)
    assert bits_equal(emb, o_emb)
    # the training forward asks for rows AND codes in one launch (ops._lsh_forward(want_bits=True))
    emb_b, bits_b = ops._lsh_forward(T(ids, dev), T(feat, dev), T(planes, dev), T(buckets, dev), want_bits=True)
    assert np.array_equal(bits_b.cpu().numpy(), o_bits) and bits_equal(emb_b.cpu().numpy(), o_emb)
    score, emb2 = ops.lsh_embed_score(T(ids, dev), T(feat, dev), T(planes, dev), T(buckets, dev), T(other, dev),
                                      want_emb=True)
    o_score, _ = oracle.lsh_embed_score(ids, feat, planes, buckets, other)
    assert bits_equal(emb2.cpu().numpy(), o_emb)
    assert bits_equal(score.cpu().numpy(), o_score)
    score_only = ops.lsh_embed_score(T(ids, dev), T(feat, dev), T(planes, dev), T(buckets, dev), T(other, dev))
    assert bits_equal(score_only.cpu().numpy(), o_score)
    # slsh on the same inputs with a large bucket table
    nb = 37
    big = rng.standard_normal((nb, D), dtype=np.float32)
    s_emb = ops.slsh_embed(T(ids, dev), T(feat, dev), T(planes, dev), T(big, dev)).cpu().numpy()
    o_s_emb, o_idx = oracle.slsh_embed(ids, feat, planes, big)
    assert bits_equal(s_emb, o_s_emb)
    assert np.array_equal(ops.slsh_index(T(ids, dev), T(feat, dev), T(planes, dev), nb).cpu().numpy(), o_idx)


@pytest.mark.parametrize("F,H,D", [(64, 8, 64), (64, 3, 64), (64, 12, 64), (22, 8, 64), (64, 8, 32), (64, 9, 64),
                                   (64, 16, 64), (64, 24, 64), (64, 31, 64), (64, 32, 64), (64, 33, 64), (64, 48, 64),
                                   (64, 64, 64), (64, 65, 64)])
def test_lookup_and_lookup_score_vs_oracle(F, H, D, oracle, ops, dev):
    """BPR lookups (in-vocab rows spliced with lsh rows) and the lookup fused with BPR.predict;
    (64, <=8, 64) takes the register-resident kernel, the others the generic LDS kernel."""
    rng = np.random.default_rng(F * 100 + H)
    n_vocab, N, B = 500, 900, 3001
    table = rng.standard_normal((n_vocab, D), dtype=np.float32)
    feat = rng.standard_normal((N, F), dtype=np.float32)
    planes = rng.standard_normal((H, F), dtype=np.float32)
    buckets = rng.standard_normal((H, D), dtype=np.float32)
    other = rng.standard_normal((B, D), dtype=np.float32)
    ids = rng.integers(0, N, size=B, dtype=np.int64)
    ids[5], ids[6] = -3, N + 10  # invalid on either side of the vocabulary boundary
    want = oracle.lsh_lookup(ids, table, feat, planes, buckets)
    got = ops.lsh_lookup(T(ids, dev), T(table, dev), T(feat, dev), T(planes, dev), T(buckets, dev)).cpu().numpy()
    assert bits_equal(got, want)
    iv = (ids >= 0) & (ids < n_vocab)
    assert bits_equal(got[iv], table[ids[iv]])  # in-vocabulary rows are verbatim copies
    assert np.isnan(got[5]).all() and np.isnan(got[6]).all()
    score, emb = ops.lsh_lookup_score(T(ids, dev), T(table, dev), T(feat, dev), T(planes, dev), T(buckets, dev),
                                      T(other, dev), want_emb=True)
    assert bits_equal(emb.cpu().numpy(), want)
    assert bits_equal(score.cpu().numpy(), oracle.rowdot(other, want))
    s2 = ops.lsh_lookup_score(T(ids, dev), T(table, dev), T(feat, dev), T(planes, dev), T(buckets, dev), T(other, dev))
    assert bits_equal(s2.cpu().numpy(), oracle.rowdot(other, want))


@pytest.mark.parametrize("H", [3, 6, 7, 8, 13, 24, 32, 47, 64])
def test_lsh_division_extremes(H, oracle, ops, dev):
    """The hot kernel divides by the code's popcount with a shared reciprocal + one fma refinement and
    falls back to IEEE division for tiny / zero / infinite sums: sweep bucket tables whose entries are
    subnormal, near the normal/subnormal border, huge, zero and mixed-sign so that every branch and
    every popcount 0..H is hit, bit for bit against the oracle's plain division."""
    rng = np.random.default_rng(H)
    N, B = 20000, 20000
    feat = rng.standard_normal((N, 64), dtype=np.float32)
    planes = rng.standard_normal((H, 64), dtype=np.float32)
    ids = np.arange(B, dtype=np.int64)
    scales = [1.0, 1e-45, 3e-39, 1.2e-38, 2.4e-38, 1e-36, 1e-30, 1e30, 3.0e38, 0.0]
    for s in scales:
        W = rng.standard_normal((H, 64)).astype(np.float32)
        with np.errstate(over="ignore", under="ignore"):
            W = (W.astype(np.float64) * s).astype(np.float32)
        W[0, :8] = 0.0
        W[1, 8:16] = -W[2, 8:16]  # exact cancellations -> +0 sums
        if s == 3.0e38:
            W[H - 1, 16:24] = np.inf
        emb = ops.lsh_embed(T(ids, dev), T(feat, dev), T(planes, dev), T(W, dev)).cpu().numpy()
        want, bits = oracle.lsh_embed(ids, feat, planes, W, want_bits=True)
        assert bits_equal(emb, want), f"scale {s}"
        assert len(set(bits.sum(1).tolist())) >= min(H, 14)  # (almost) every popcount occurs (binomial tails thin out)


def test_empty_batches(ops, dev):
    ids = torch.empty((0,), dtype=torch.int64, device=dev)
    feat = torch.randn(10, 8, device=dev)
    planes = torch.randn(4, 8, device=dev)
    buckets = torch.randn(4, 16, device=dev)
    assert ops.lsh_embed(ids, feat, planes, buckets).shape == (0, 16)
    assert ops.gather_rows(ids, buckets).shape == (0, 16)
    assert ops.mapper_map(ids, "3round", 5, 3).shape == (0,)
    assert ops.rowdot(torch.empty(0, 16, device=dev), torch.empty(0, 16, device=dev)).shape == (0,)


@pytest.mark.parametrize("B,N,D,k", [(5, 300, 64, 10), (130, 1000, 64, 20), (33, 257, 22, 2), (200, 5000, 128, 7)])
def test_scores_topk_vs_oracle(B, N, D, k, oracle, ops, dev):
    rng = np.random.default_rng(N + D)
    U = rng.standard_normal((B, D), dtype=np.float32)
    E = rng.standard_normal((N, D), dtype=np.float32)
    E[5] = E[9]  # exact ties -> lower index first
    S = ops.full_sort_scores(T(U, dev), T(E, dev)).cpu().numpy()
    assert bits_equal(S, oracle.full_sort_scores(U, E))
    vals, idx = ops.score_topk(T(U, dev), T(E, dev), k, 1)
    o_vals, o_idx = oracle.score_topk(U, E, k, 1)
    assert np.array_equal(idx.cpu().numpy(), o_idx)
    assert bits_equal(vals.cpu().numpy(), o_vals)


@pytest.mark.parametrize("B,K,N_out", [(7, 16, 512), (300, 1024, 512), (513, 512, 64), (130, 22, 8), (64, 100, 130)])
def test_linear_act_vs_oracle(B, K, N_out, oracle, ops, dev):
    """Linear layers of the dhe/fdhe/dnn hash nets: pre-activation bit-exact (same fmaf chain),
    GELU / sigmoid within a few ulp (device erff/expf vs libm)."""
    rng = np.random.default_rng(K + N_out)
    X = (rng.standard_normal((B, K)) * (1000.0 if K == 16 else 1.0)).astype(np.float32)
    W = (rng.standard_normal((N_out, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N_out).astype(np.float32)
    got = ops.linear_act(T(X, dev), T(W, dev), T(b, dev), None).cpu().numpy()
    assert bits_equal(got, oracle.linear_act(X, W, b, 0))
    for name, act in (("gelu", 1), ("sigmoid", 2)):
        got = ops.linear_act(T(X, dev), T(W, dev), T(b, dev), name).cpu().numpy()
        want = oracle.linear_act(X, W, b, act)
        assert np.allclose(got, want, rtol=2e-6, atol=1e-7), name
    ref = torch.nn.functional.gelu(torch.nn.functional.linear(torch.from_numpy(X), torch.from_numpy(W), torch.from_numpy(b)))
    got = ops.linear_act(T(X, dev), T(W, dev), T(b, dev), "gelu").cpu()
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-4 * float(ref.abs().max()))


# mi_oov_linear_x3: the same layer on the bf16 matrix cores, every operand as three bf16 planes.  NOT the oracle's summation
# order, so parity is a tolerance: with u = 2^-24 (half an f32 ulp) the f32 chain's own error against the exact value is
# bounded by K u sum|x||w| and is ~sqrt(K) u of it in practice; the split form drops three cross terms below u |x||w| each and
# otherwise rounds only in the f32 accumulator.  Bound asserted: 8 u (sum_k |x_k||w_k| + |b|) against the ORACLE (both sides'
# error together; measured: 2.3e-7 = 4 u at K = 1024), and against an f64 product the split form may not be further off than
# twice the f32 kernel.
X3_SHAPES = [(300, 1024, 512),   # the pipelined 256 x 256 form (K % 16 == 0, N_out > 128): two row blocks, ragged
             (2100, 512, 512),   # nine row blocks: the XCD remap leaves seven slots of the last round empty
             (700, 64, 300),     # four stages per tile, the last n-block ragged
             (40, 32, 129),      # two stages: the shortest stream the pipelined kernel takes
             (513, 512, 64),     # narrow output: 128 x 64 tiles
             (257, 70, 130),     # K with a tail chunk, N_out ragged
             (64, 22, 512),      # dnn embedder's first layer (22 feature columns)
             (5, 1030, 33),      # K % 4 != 0: scalar loads
             (1, 16, 1)]


@pytest.mark.parametrize("form", ["0", "4"])  # MI_OOV_X3_SHAPE: by shape (small batches: 128 x 128 tiles) / the pipelined kernel wherever it applies
@pytest.mark.parametrize("B,K,N_out", X3_SHAPES)
def test_linear_x3_vs_oracle(B, K, N_out, form, oracle, ops, dev, monkeypatch):
    if form == "4" and not (K % 16 == 0 and K >= 32 and N_out > 128):
        pytest.skip("shape outside the pipelined kernel")
    monkeypatch.setenv("MI_OOV_X3_SHAPE", form)
    rng = np.random.default_rng(B + K + N_out)
    X = (rng.random((B, K)) * 2 - 1).astype(np.float32)
    X[0, : min(K, 5)] = [1e-30, -3e4, 0.0, 2.0 ** -100, 1e20][: min(K, 5)]  # small, large and zero operands
    W = (rng.standard_normal((N_out, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N_out).astype(np.float32)
    want = oracle.linear_act(X, W, b, 0)
    den = np.abs(X).astype(np.float64) @ np.abs(W).astype(np.float64).T + np.abs(b)
    u = 2.0 ** -24
    got = ops.linear_act_x3(T(X, dev), T(W, dev), T(b, dev), None).cpu().numpy()
    assert np.all(np.abs(got.astype(np.float64) - want) <= 8 * u * den)
    truth = X.astype(np.float64) @ W.astype(np.float64).T + b
    e_x3, e_f32 = np.abs(got - truth) / den, np.abs(want - truth) / den
    assert e_x3.max() <= 2 * e_f32.max() + 2 * u and np.sqrt((e_x3 ** 2).mean()) <= 2 * np.sqrt((e_f32 ** 2).mean()) + u
    for name, act in (("gelu", 1), ("sigmoid", 2)):  # both activations are 1.13-Lipschitz at most
        got = ops.linear_act_x3(T(X, dev), T(W, dev), T(b, dev), name).cpu().numpy()
        w_act = oracle.linear_act(X, W, b, act)
        assert np.all(np.abs(got.astype(np.float64) - w_act) <= 10 * u * den + 2e-6 * np.abs(w_act) + 1e-7), name


def test_linear_x3_padded_rows_and_full_size(oracle, ops, dev, monkeypatch):
    """(a) An input padded to a multiple of 16 columns (what hash_net_forward hands over for fdhe's K + F columns) goes
    through the pipelined kernel: bit-identical to the generic kernel on the unpadded rows (the same arithmetic in the same
    order), whatever finite values the padding columns hold -- they meet zero weights.  (b) BASELINE's dhe shape, 65536 x 1024 -> 512: every
    row agrees with the f32 kernel within that bound, and the result does not depend on where a row sits in the batch
    (rows of a 65536-row call == the same rows as a 32768-row call: the persistent workgroups walk other tiles)."""
    rng = np.random.default_rng(77)
    B, K, N_out = 700, 1046, 300
    X = (rng.random((B, K)) * 2 - 1).astype(np.float32)
    W = (rng.standard_normal((N_out, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N_out).astype(np.float32)
    Xt, Wt, bt = T(X, dev), T(W, dev), T(b, dev)
    den = Xt.abs() @ Wt.abs().T + bt.abs()
    plain = ops.linear_act_x3(Xt, Wt, bt, None)  # K % 4 != 0: the generic kernel
    monkeypatch.setenv("MI_OOV_X3_SHAPE", "4")  # (by shape a batch this small takes 128 x 128 tiles)
    padded = ops.linear_act_x3(torch.nn.functional.pad(Xt, (0, -K % 16)), Wt, bt, None)
    sevens = ops.linear_act_x3(torch.nn.functional.pad(Xt, (0, -K % 16), value=7.0), Wt, bt, None)
    monkeypatch.delenv("MI_OOV_X3_SHAPE")
    assert torch.equal(plain, padded) and torch.equal(padded, sevens)  # (the two kernels do the same arithmetic in the same order)
    assert bool(((plain - ops.linear_act(Xt, Wt, bt, None)).abs() <= 8 * 2.0 ** -24 * den).all())
    with pytest.raises(ValueError):
        ops.linear_act_x3(torch.nn.functional.pad(Xt, (0, 1)), Wt, bt, None)
    g = torch.Generator(device=dev).manual_seed(5)
    B, K, N_out = 65536, 1024, 512
    Xb = torch.rand((B, K), generator=g, device=dev) * 2 - 1
    Wb = torch.randn((N_out, K), generator=g, device=dev) / 32
    bb = torch.randn((N_out,), generator=g, device=dev)
    y = ops.linear_act_x3(Xb, Wb, bb, None)
    ref = ops.linear_act(Xb, Wb, bb, None)  # the bit-exact kernel (pinned on the oracle at small sizes)
    den = Xb.abs() @ Wb.abs().T + bb.abs()
    assert bool(((y - ref).abs() <= 8 * 2.0 ** -24 * den).all())
    part = ops.linear_act_x3(Xb[32768:], Wb, bb, None)
    assert torch.equal(part, y[32768:])


@pytest.mark.parametrize("M,K,N_out,ksplit", [(512, 2048, 1024, 8), (70, 999, 33, 5), (130, 64, 200, 7), (3, 16, 5, 4)])
def test_linear_x3_splitk_vs_oracle(M, K, N_out, ksplit, oracle, ops, dev):
    """mi_oov_linear_x3_splitk (training's dW = dZ^T X shape: few output tiles, long K): K in ksplit shares, slabs added in
    order -- the bound of test_linear_x3_vs_oracle against the oracle, the same bits on a second run, ksplit = 1 equal
    to the un-split entry; shares that get no stage (ksplit = 7 over 4 stages) contribute zeros."""
    rng = np.random.default_rng(M + K)
    X = (rng.random((M, K)) * 2 - 1).astype(np.float32)
    W = (rng.standard_normal((N_out, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N_out).astype(np.float32)
    Xt, Wt, bt = T(X, dev), T(W, dev), T(b, dev)
    den = np.abs(X).astype(np.float64) @ np.abs(W).astype(np.float64).T + np.abs(b)
    for act, code in ((None, 0), ("gelu", 1), ("sigmoid", 2)):
        got = ops.linear_act_x3(Xt, Wt, bt, act, ksplit=ksplit)
        want = oracle.linear_act(X, W, b, code)
        assert np.all(np.abs(got.cpu().numpy().astype(np.float64) - want) <= 10 * 2.0 ** -24 * den + 2e-6 * np.abs(want) + 1e-7), act
        assert torch.equal(got, ops.linear_act_x3(Xt, Wt, bt, act, ksplit=ksplit))
    # the weights given as their transpose (mi_oov_linear_x3_prepare_t: how training's backward products find their operands)
    Wtt = Wt.t().contiguous()
    assert torch.equal(ops.linear_act_x3(Xt, Wtt, bt, None, ops.LinearX3Weights(Wtt, transposed=True), ksplit), ops.linear_act_x3(Xt, Wt, bt, None, ksplit=ksplit))
    lib = ops.C.lib()
    assert lib.mi_oov_linear_x3_prepare_t(None, 8, 16, None, None) == -1 and lib.mi_oov_linear_x3_prepare_t(None, 0, 16, None, None) == -2
    assert lib.mi_oov_linear_x3_splitk_workspace(M, N_out, ksplit) == ksplit * M * N_out * 4
    assert lib.mi_oov_linear_x3_splitk(None, 4, 16, None, None, 8, 1, None, 0, None, None) == -2
    assert lib.mi_oov_linear_x3_splitk(None, 4, 16, None, None, 8, 1, None, 4, None, None) == -1


@pytest.mark.parametrize("B,K,N_out", [(33300, 32, 512), (66000, 48, 300), (35000, 160, 257)])
def test_linear_x3_persistent_stream_of_short_tiles(B, K, N_out, ops, dev, monkeypatch):
    """More 256 x 256 tiles than CUs, two to ten stages each: the persistent workgroups of the pipelined kernel cross
    tile boundaries every few stages (the next tile's loads issued in this tile's last two stages, K = 32: in every
    stage).  Bit-identical to the generic kernel's 128 x 128 tiles."""
    g = torch.Generator(device=dev).manual_seed(B + K)
    X = torch.rand((B, K), generator=g, device=dev) * 2 - 1
    W = torch.randn((N_out, K), generator=g, device=dev) / K ** 0.5
    b = torch.randn((N_out,), generator=g, device=dev)
    monkeypatch.setenv("MI_OOV_X3_SHAPE", "1")
    want = ops.linear_act_x3(X, W, b, "gelu")
    monkeypatch.setenv("MI_OOV_X3_SHAPE", "4")
    got = ops.linear_act_x3(X, W, b, "gelu")
    monkeypatch.delenv("MI_OOV_X3_SHAPE")
    assert torch.equal(got, want) and torch.equal(ops.linear_act_x3(X, W, b, "gelu"), want)
    ref = torch.nn.functional.gelu(X @ W.T + b)
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-5)


def test_siphash_into_wider_rows(oracle, ops, dev):
    """mi_oov_siphash24_mod_ld: the hashes written into the first K columns of wider rows (fdhe's net input), the other
    columns untouched; the integers are those of the plain entry (pinned on the oracle and the reference's fixture)."""
    rng = np.random.default_rng(4)
    ids = T(rng.integers(0, 1 << 40, 777), dev)
    keys = T(rng.integers(0, 256, (70, 16), dtype=np.uint8), dev)
    plain = ops.siphash24_mod(ids, keys)
    buf = torch.full((777, 96), -3.0, device=dev)
    out = ops.siphash24_mod(ids, keys, out=buf)
    assert out.data_ptr() == buf.data_ptr() and torch.equal(buf[:, :70], plain) and bool((buf[:, 70:] == -3.0).all())
    assert np.array_equal(plain.cpu().numpy(), oracle.siphash24_mod(ids.cpu().numpy(), keys.cpu().numpy(), 16777216))
    with pytest.raises(ValueError):
        ops.siphash24_mod(ids, keys, out=torch.empty((777, 69), device=dev))
    assert ops.C.lib().mi_oov_siphash24_mod_ld(None, 4, None, 8, 16777216, None, 7, None) == -2


def test_linear_x3_random_shapes(ops, dev, monkeypatch):
    """Thirty random (rows, K, N_out, activation) through every tile form the shape admits, against the bit-exact f32 kernel
    (itself pinned on the oracle): the bound of test_linear_x3_vs_oracle, and all forms bit-identical to one another."""
    rng = np.random.default_rng(2024)
    g = torch.Generator(device=dev).manual_seed(7)
    for _ in range(30):
        B = int(rng.choice([1, 3, 64, 129, 700, 2049, 9000]))
        K = int(rng.choice([1, 7, 16, 22, 32, 48, 100, 256, 1000, 1024]))
        N_out = int(rng.choice([1, 5, 64, 65, 129, 300, 512, 777]))
        act = [None, "gelu", "sigmoid"][int(rng.integers(0, 3))]
        X = torch.rand((B, K), generator=g, device=dev) * 2 - 1
        W = torch.randn((N_out, K), generator=g, device=dev) / K ** 0.5
        b = torch.randn((N_out,), generator=g, device=dev)
        den = X.abs() @ W.abs().T + b.abs()
        pre = ops.linear_act(X, W, b, None)
        outs = []
        for form in ("0", "1", "2", "3", "4"):
            monkeypatch.setenv("MI_OOV_X3_SHAPE", form)
            outs.append(ops.linear_act_x3(X, W, b, None))
            assert torch.equal(outs[-1], outs[0]), (B, K, N_out, form)
            got = ops.linear_act_x3(X, W, b, act)
            want = ops.linear_act(X, W, b, act)
            assert bool(((got - want).abs() <= 10 * 2.0 ** -24 * den + 2e-6 * want.abs() + 1e-7).all()), (B, K, N_out, act, form)
        assert bool(((outs[0] - pre).abs() <= 8 * 2.0 ** -24 * den).all()), (B, K, N_out)
    monkeypatch.delenv("MI_OOV_X3_SHAPE")


def test_linear_x3_same_sign_operands(ops, dev):
    """All-positive operands (no cancellation: every partial sum is as large as it can be, the worst case for any f32
    accumulation -- the f32 chain's own error grows to several 1e-6 of sum|x||w| here, beyond the 8 u of the zero-mean
    tests): the split form stays below the f32 kernel's error against an f64 product, and within 2e-5 of it relatively
    at K = 4096 (north_star's bar: 1e-5 at the widths the path uses, K <= 1024: asserted at 1024 too)."""
    g = torch.Generator(device=dev).manual_seed(3)
    for K, bar in ((1024, 1e-5), (4096, 2e-5)):
        X = torch.rand((512, K), generator=g, device=dev) + 0.5
        W = torch.exp(torch.randn((256, K), generator=g, device=dev))
        b = torch.zeros((256,), device=dev)
        truth = X.double() @ W.double().T
        e3 = ((ops.linear_act_x3(X, W, b, None).double() - truth) / truth).abs().max().item()
        e32 = ((ops.linear_act(X, W, b, None).double() - truth) / truth).abs().max().item()
        assert e3 <= e32 and e3 <= bar, (K, e3, e32)


def test_linear_x3_split_is_exact(ops, dev):
    """The three bf16 planes mi_oov_linear_x3_prepare makes add up to the f32 weight EXACTLY (each the round-to-nearest bf16
    of what the ones before it left: 3 x 8 significand bits), from the plain and from the transposed source; rows beyond
    N_out and columns beyond K are zero.  Decoded on the CPU from the split's documented layout [K/16][N -> 256s][3][16]."""
    rng = np.random.default_rng(11)
    N_out, K = 70, 50
    W = (rng.standard_normal((N_out, K)) * 10.0 ** rng.integers(-20, 20, (N_out, K))).astype(np.float32)
    W[0, :4] = [0.0, -0.0, 1.0, -3.0e38]
    Wt = T(W, dev)
    for transposed in (False, True):
        src = Wt.t().contiguous() if transposed else Wt
        raw = ops.LinearX3Weights(src, transposed=transposed).get().cpu().numpy()
        chunks, Np = -(-K // 16), 256
        planes = raw.view(np.uint16).reshape(chunks, Np, 3, 16).astype(np.uint32) << 16
        planes = planes.view(np.float32)                                   # bf16 -> f32: exact
        total = (planes[:, :, 0] + planes[:, :, 1]) + planes[:, :, 2]        # h + m exact in f32, + l exact
        full = total.transpose(1, 0, 2).reshape(Np, chunks * 16)            # [n, k]
        assert np.array_equal(full[:N_out, :K], W), transposed  # (value equality: the planes of -0 are -0, +0, +0)
        assert not full[N_out:].any() and not full[:, K:].any()
        assert np.all(np.abs(planes[:, :N_out, 1]) <= np.abs(planes[:, :N_out, 0]) * 2.0 ** -8 + 1e-45)  # m below half an ulp of h


def test_linear_x3_special_values_and_weights_cache(oracle, ops, dev):
    """Non-finite operands give non-finite results exactly where the f32 product does (NaN where that holds +-inf: the lower
    planes of an infinite value are inf - inf); the split weights follow the weight tensor's version counter."""
    rng = np.random.default_rng(9)
    X = rng.standard_normal((130, 64)).astype(np.float32)
    W = rng.standard_normal((200, 64)).astype(np.float32)
    b = np.zeros(200, np.float32)
    X[3, 7], X[5, 0], X[9, 1] = np.inf, np.nan, -np.inf
    got = ops.linear_act_x3(T(X, dev), T(W, dev), T(b, dev), None).cpu().numpy()
    want = oracle.linear_act(X, W, b, 0)
    assert np.array_equal(np.isfinite(got), np.isfinite(want)) and not np.isfinite(want[[3, 5, 9]]).any() and np.isfinite(want[10:]).all()
    Wt, Xt, bt = T(W, dev), T(X[10:], dev), T(b, dev)
    cache = ops.LinearX3Weights(Wt)
    y0 = ops.linear_act_x3(Xt, Wt, bt, None, cache)
    first = cache.get()
    assert cache.get() is first and cache.version == Wt._version
    with torch.no_grad():
        Wt.mul_(2.0)
    y1 = ops.linear_act_x3(Xt, Wt, bt, None, cache)  # re-split: an exact doubling
    assert torch.equal(y1, 2 * y0)
    Wt.data.mul_(0.5)  # a write through .data moves no counter: invalidate() by hand
    cache.invalidate()
    assert torch.equal(ops.linear_act_x3(Xt, Wt, bt, None, cache), y0)
    # a Parameter re-pointed at other memory (`p.data = new`: same object, same version counter) is seen through its address
    P = torch.nn.Parameter(Wt.clone())
    pc = ops.LinearX3Weights(P)
    assert torch.equal(ops.linear_act_x3(Xt, P, bt, None, pc), y0)
    v0 = P._version
    P.data = (2.0 * Wt).contiguous()
    assert P._version == v0
    assert torch.equal(ops.linear_act_x3(Xt, P, bt, None, pc), 2 * y0)
    # contiguous outputs only: a non-contiguous `out` would be written into a temporary copy
    ids = torch.arange(4, device=dev)
    keys = torch.zeros((3, 16), dtype=torch.uint8, device=dev)
    with pytest.raises(ValueError):
        ops.siphash24_mod(ids, keys, out=torch.empty((8, 4), device=dev)[:, :4].t())


def test_hash_net_forward_runs_the_split_layers_whatever_the_batch(ops, dev, monkeypatch):
    """Inference runs on mi_oov_linear_x3 at every batch size, and a row's result does not depend on the batch it sits in
    (1500 rows: 128 x 128 tiles; the same rows inside a 40000-row batch: the pipelined 256 x 256 kernel for the wide layer);
    MI_OOV_LINEAR_X3=0 selects the f32 kernel."""
    monkeypatch.delenv("MI_OOV_LINEAR_X3", raising=False)  # (the default route, whatever this environment sets)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(48, 512), torch.nn.GELU(), torch.nn.Linear(512, 64), torch.nn.Sigmoid()).to(dev)
    calls = {"x3": 0, "f32": 0}
    real_x3, real_f32 = ops.linear_act_x3, ops.linear_act
    monkeypatch.setattr(ops, "linear_act_x3", lambda *a, **k: (calls.__setitem__("x3", calls["x3"] + 1), real_x3(*a, **k))[1])
    monkeypatch.setattr(ops, "linear_act", lambda *a, **k: (calls.__setitem__("f32", calls["f32"] + 1), real_f32(*a, **k))[1])
    x = torch.rand((40000, 48), device=dev) * 2 - 1
    with torch.no_grad():
        big, mid, small, ref = ops.hash_net_forward(net, x), ops.hash_net_forward(net, x[:1500]), ops.hash_net_forward(net, x[:7]), net(x)
    assert calls == {"x3": 6, "f32": 0}
    assert torch.allclose(big, ref, rtol=1e-5, atol=1e-6)
    assert torch.equal(mid, big[:1500]) and torch.equal(small, big[:7])
    monkeypatch.setenv("MI_OOV_LINEAR_X3", "0")
    with torch.no_grad():
        exact = ops.hash_net_forward(net, x[:1500])
    assert calls == {"x3": 6, "f32": 2}
    assert (exact - mid).abs().max().item() <= 1e-6


def test_topk_edge_cases(oracle, ops, dev):
    rng = np.random.default_rng(3)
    U = rng.standard_normal((17, 8), dtype=np.float32)
    E = rng.standard_normal((300, 8), dtype=np.float32)
    E[7] = np.nan           # NaN scores sort first (torch.topk)
    E[40:60] = E[40]        # a run of exact ties straddling the k-th place
    E[100] = np.inf
    for k, skip in ((1, 0), (5, 1), (50, 1), (256, 0), (300, 0), (256, 290)):  # k > 256 -> multi-pass kernel; k > columns
        vals, idx = ops.score_topk(T(U, dev), T(E, dev), k, skip)
        o_vals, o_idx = oracle.score_topk(U, E, k, skip)
        assert np.array_equal(idx.cpu().numpy(), o_idx), (k, skip)
        assert bits_equal(vals.cpu().numpy(), o_vals), (k, skip)
    # many columns, heavy duplication of values (stress the radix passes)
    E2 = np.round(rng.standard_normal((20000, 8)), 1).astype(np.float32)
    U2 = np.round(rng.standard_normal((9, 8)), 1).astype(np.float32)
    vals, idx = ops.score_topk(T(U2, dev), T(E2, dev), 100, 1)
    o_vals, o_idx = oracle.score_topk(U2, E2, 100, 1)
    assert np.array_equal(idx.cpu().numpy(), o_idx) and bits_equal(vals.cpu().numpy(), o_vals)


@pytest.mark.parametrize("B,N,D,k,skip", [(70, 3000, 64, 10, 1), (9, 1300, 22, 5, 0), (130, 4159, 64, 20, 1),
                                          (33, 20000, 32, 100, 1), (5, 40000, 64, 256, 7), (70, 5000, 4, 10, 1),
                                          (40, 9000, 48, 20, 0), (12, 3000, 63, 7, 1), (20, 4000, 1, 3, 0),
                                          (6, 2000, 96, 8, 1)])
def test_fused_topk_vs_oracle(B, N, D, k, skip, oracle, ops, dev):
    """N >= 128 k takes the fused two-pass path (tile maxima -> tau -> filter -> rank): exact top-k with
    the [B,N] matrix never written, including ties across tiles, NaN rows and duplicated scores."""
    rng = np.random.default_rng(N + k)
    U = rng.standard_normal((B, D), dtype=np.float32)
    E = rng.standard_normal((N, D), dtype=np.float32)
    E[100:160] = E[100]            # 60 exact ties inside one tile and across its neighbour
    E[N - 1] = E[3]                # tie between the first and the last tile
    E[700] = np.nan                # NaN sorts first
    U[1] = 0.0                     # a whole row of equal scores (+0): candidate list overflows -> fallback
    vals, idx = ops.score_topk(T(U, dev), T(E, dev), k, skip)
    o_vals, o_idx = oracle.score_topk(U, E, k, skip)
    assert np.array_equal(idx.cpu().numpy(), o_idx)
    assert bits_equal(vals.cpu().numpy(), o_vals)


@pytest.mark.parametrize("case", ["queue_overflow", "popular_columns", "ties_300", "ties_450"])
def test_fused_topk_direct_filter_queue(case, oracle, ops, dev):
    """The second pass (bf16_filter_direct_kernel) keeps passing scores in a per-wave queue of 256 records that lives
    across column blocks.  queue_overflow: 40 all-zero user rows inside one 128-row block tie everywhere, a wave finds
    far more than 256 passing lane-groups in a block, records are dropped and every row of that workgroup -- the
    ordinary ones too -- must come out of the exact fallback unchanged.  popular_columns: a few items score high for
    every user, so single lanes push in most of the 22 groups of a block, block after block.  ties_300 / ties_450: that
    many items share one row, so for the users that like it all of them are candidates inside three column blocks: the
    (row, strip) lists run over into the row's overflow list (512 slots, of which the finalize kernel keeps 128 in LDS and
    reads the rest in place), and with 450 of them more survive the bf16 cut than the kernel can rank in LDS: fallback."""
    rng = np.random.default_rng(11)
    B, N, k = 300, 9000, 20
    U = rng.standard_normal((B, 64), dtype=np.float32)
    E = rng.standard_normal((N, 64), dtype=np.float32)
    if case == "queue_overflow":
        U[130:170] = 0.0
    elif case.startswith("ties_"):
        n_ties = int(case.split("_")[1])
        E[100:100 + n_ties] = 2.0 * U[:40].mean(0) / np.linalg.norm(U[:40].mean(0)) * 8.0
    else:
        E[[5, 777, 4100, 8999]] = 3.0 * np.abs(U).mean(0)   # large positive scores for most users
        U = np.abs(U)
    vals, idx = ops.score_topk(T(U, dev), T(E, dev), k, 1)
    o_vals, o_idx = oracle.score_topk(U, E, k, 1)
    assert np.array_equal(idx.cpu().numpy(), o_idx)
    assert bits_equal(vals.cpu().numpy(), o_vals)


@pytest.mark.parametrize("case", ["cluster_small", "cluster_big", "neg_nan", "many_strips", "skip_mid_tile"])
def test_fused_topk_bf16_lists(case, oracle, ops, dev):
    """The D = 64 fused path filters on bf16 matrix-core scores into per-(row, strip) lists and re-scores the
    survivors exactly.  Shapes that stress the list plumbing: a cluster of dominant columns inside one strip (lists run
    over into the row's overflow list), a cluster too big for both (exact fallback for every row), NaNs of either sign,
    few rows (128 strips, 8-entry lists), a skip bound inside a tile.  Bit-exact against the oracle in all of them."""
    rng = np.random.default_rng(sum(map(ord, case)))
    B, N, k, skip = 96, 20000, 20, 1
    U = rng.standard_normal((B, 64), dtype=np.float32)
    E = rng.standard_normal((N, 64), dtype=np.float32)
    if case == "cluster_small":
        E[5000:5060] *= 4.0            # 60 adjacent columns carry every row's top scores (positive or negative)
    elif case == "cluster_big":
        E[5000:5600] *= 4.0
    elif case == "neg_nan":
        E[77] = np.float32(np.nan)
        E[9000] = np.frombuffer(np.uint32(0xFFC00000).tobytes(), np.float32)[0]   # x86's default (negative) NaN
        E[12000, 3] = np.inf
        U[5] = 0.0
        U[6, 0] = np.inf               # inf - inf / 0 * inf rows
    elif case == "many_strips":
        B, N, k = 7, 60000, 50
        U = rng.standard_normal((B, 64), dtype=np.float32)
        E = rng.standard_normal((N, 64), dtype=np.float32)
    elif case == "skip_mid_tile":
        skip, k = 333, 33
    vals, idx = ops.score_topk(T(U, dev), T(E, dev), k, skip)
    o_vals, o_idx = oracle.score_topk(U, E, k, skip)
    assert np.array_equal(idx.cpu().numpy(), o_idx)
    assert bits_equal(vals.cpu().numpy(), o_vals)


@pytest.mark.parametrize("su,se", [(1e-20, 1e-20), (1e-3, 1e3), (1e3, 1e2), (1e15, 1e15), (1e-30, 1.0)])
def test_fused_topk_bf16_bound_over_magnitudes(su, se, oracle, ops, dev):
    """The bf16 prefilter's error bound scales with the operands' norms (and carries a term for subnormal operands the
    matrix cores may flush): the top-k stays bit-exact from subnormal products to squared norms that overflow f32
    (there the bound is infinite and the rows take the exact path)."""
    rng = np.random.default_rng(5)
    U = (rng.standard_normal((40, 64)) * su).astype(np.float32)
    E = (rng.standard_normal((5000, 64)) * se).astype(np.float32)
    vals, idx = ops.score_topk(T(U, dev), T(E, dev), 10, 1)
    o_vals, o_idx = oracle.score_topk(U, E, 10, 1)
    assert np.array_equal(idx.cpu().numpy(), o_idx)
    assert bits_equal(vals.cpu().numpy(), o_vals)


def test_fused_topk_large_catalogue(oracle, ops, dev):
    """3 M items: 11 719 sampled tile maxima per row (the radix tau kernel), 128 strips of 183 blocks, column
    indices far beyond 2^16; a few user rows keep the oracle to a couple of seconds."""
    rng = np.random.default_rng(12)
    U = rng.standard_normal((6, 64), dtype=np.float32)
    E = rng.standard_normal((3_000_000, 64), dtype=np.float32)
    vals, idx = ops.score_topk(T(U, dev), T(E, dev), 10, 1)
    o_vals, o_idx = oracle.score_topk(U, E, 10, 1)
    assert np.array_equal(idx.cpu().numpy(), o_idx)
    assert bits_equal(vals.cpu().numpy(), o_vals)


def test_topk_prepared_catalogue(oracle, ops, dev):
    """A catalogue prepared once (mi_oov_topk_catalogue_prepare) gives the same top-k as the per-call path, with and
    without exclusions; writing to the table invalidates it."""
    rng = np.random.default_rng(31)
    U = rng.standard_normal((150, 64), dtype=np.float32)
    E = rng.standard_normal((7000, 64), dtype=np.float32)
    Eg = T(E, dev)
    cat = ops.TopkCatalogue(Eg)
    vals, idx = ops.score_topk(T(U, dev), cat, 15, 1)
    o_vals, o_idx = oracle.score_topk(U, E, 15, 1)
    assert np.array_equal(idx.cpu().numpy(), o_idx) and bits_equal(vals.cpu().numpy(), o_vals)
    lens = rng.integers(0, 400, 150)
    ptr = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    cols = rng.integers(1, 7000, int(ptr[-1])).astype(np.int64)
    v2, i2 = ops.score_topk_excl(T(U, dev), cat, 15, T(ptr, dev), T(cols, dev), n_skip_low=1)
    ov, oi = oracle.score_topk_excl(U, E, 15, ptr, cols, 1)
    assert np.array_equal(i2.cpu().numpy(), oi) and np.array_equal(v2.cpu().numpy(), ov)
    assert ops.TopkCatalogue.of(torch.zeros((7000, 80), device=dev)) is not None   # rows of up to 128 floats (round 3)
    assert ops.TopkCatalogue.of(torch.zeros((7000, 129), device=dev)) is None      # a width the fused path does not take
    Eg[5] += 1.0
    assert not cat.fresh()
    with pytest.raises(ValueError):
        ops.score_topk(T(U, dev), cat, 15, 1)


@pytest.mark.parametrize("D", [22, 40, 65, 100, 128])
def test_topk_narrow_rows_take_the_bf16_path(D, oracle, ops, dev):
    """Rows that are not 64 floats wide -- a knn search over 22 feature columns, a 32-d model, and from round 3 on 65..128
    floats (a 128-d model: BASELINE config 4's row width): the bf16 copies are zero-padded to 64 or 128 k, the exact
    re-score runs the oracle's chain for the real width -- plain, with exclusions, and through a prepared catalogue; same
    results as the oracle, bit for bit."""
    rng = np.random.default_rng(D)
    B, N, k = 200, 12000, 12
    U = rng.standard_normal((B, D), dtype=np.float32)
    E = rng.standard_normal((N, D), dtype=np.float32)
    E[50:90] = E[50]                      # ties
    U[3] = 0.0                            # a row of equal scores -> exact fallback with the narrow chain
    Ug, Eg = T(U, dev), T(E, dev)
    o_vals, o_idx = oracle.score_topk(U, E, k, 1)
    vals, idx = ops.score_topk(Ug, Eg, k, 1)
    assert np.array_equal(idx.cpu().numpy(), o_idx) and bits_equal(vals.cpu().numpy(), o_vals)
    cat = ops.TopkCatalogue.of(Eg)
    assert cat is not None
    vals, idx = ops.score_topk(Ug, cat, k, 1)
    assert np.array_equal(idx.cpu().numpy(), o_idx) and bits_equal(vals.cpu().numpy(), o_vals)
    lens = rng.integers(0, 300, B)
    ptr = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    cols = rng.integers(1, N, int(ptr[-1])).astype(np.int64)
    ov, oi = oracle.score_topk_excl(U, E, k, ptr, cols, 1)
    for table in (Eg, cat):
        v2, i2 = ops.score_topk_excl(Ug, table, k, T(ptr, dev), T(cols, dev), n_skip_low=1)
        assert np.array_equal(i2.cpu().numpy(), oi) and np.array_equal(v2.cpu().numpy(), ov)
    from mi_oov import _cabi as C
    assert C.lib().mi_oov_score_topk_masked_workspace(B, N, D, k) > 0 and C.lib().mi_oov_score_topk_masked_workspace(B, N, 129, k) == 0


def test_fused_topk_random_shapes(oracle, ops, dev):
    """Sixty random shapes through the fused path -- user batches that are not multiples of the 64-row workgroups,
    catalogues that end inside a tile, widths 1..128 (two k-halves above 64), k up to 64, skipped low columns, tie runs and
    a zero user row -- against the oracle, bit for bit."""
    rng = np.random.default_rng(2026)
    for case in range(60):
        D = int(rng.choice([1, 3, 8, 16, 22, 31, 32, 40, 48, 63, 64] if case < 40 else [65, 72, 96, 100, 127, 128]))
        k = int(rng.choice([1, 2, 5, 10, 20, 33, 64]))
        N = int(rng.integers(128 * k, 128 * k + 30000))
        B = int(rng.integers(1, 200))
        skip = int(rng.choice([0, 1, 1, 5]))
        U = rng.standard_normal((B, D), dtype=np.float32)
        E = rng.standard_normal((N, D), dtype=np.float32)
        if case % 3 == 0:
            a = int(rng.integers(0, N - 70))
            E[a:a + int(rng.integers(2, 70))] = E[a]
        if case % 5 == 0:
            U[int(rng.integers(0, B))] = 0.0
        vals, idx = ops.score_topk(T(U, dev), T(E, dev), k, skip)
        o_vals, o_idx = oracle.score_topk(U, E, k, skip)
        assert np.array_equal(idx.cpu().numpy(), o_idx), (case, B, N, D, k, skip)
        assert bits_equal(vals.cpu().numpy(), o_vals), (case, B, N, D, k, skip)


def test_score_topk_user_chunks(oracle, ops, dev, monkeypatch):
    """ops.score_topk bounds its workspace by going through big user batches in chunks (multiples of 128 rows)."""
    rng = np.random.default_rng(77)
    U = rng.standard_normal((700, 64), dtype=np.float32)
    E = rng.standard_normal((9000, 64), dtype=np.float32)
    monkeypatch.setattr(ops, "_TOPK_WORKSPACE_MAX_BYTES", 3 << 20)   # 700 rows need ~9 MB: three chunks
    vals, idx = ops.score_topk(T(U, dev), T(E, dev), 12, 1)
    o_vals, o_idx = oracle.score_topk(U, E, 12, 1)
    assert np.array_equal(idx.cpu().numpy(), o_idx)
    assert bits_equal(vals.cpu().numpy(), o_vals)


def test_gather_splice_vs_oracle(oracle, ops, dev):
    rng = np.random.default_rng(11)
    for D in (64, 1, 50, 200):
        table = rng.standard_normal((300, D), dtype=np.float32)
        ids = rng.integers(0, 400, size=1000, dtype=np.int64)
        n_oov = int((ids >= 300).sum())
        oov_rows = rng.standard_normal((n_oov, D), dtype=np.float32)
        got = ops.splice_rows(T(ids, dev), T(table, dev), T(oov_rows, dev)).cpu().numpy()
        assert bits_equal(got, oracle.splice_rows(ids, table, oov_rows))
        iv = ids[ids < 300]
        assert bits_equal(ops.gather_rows(T(iv, dev), T(table, dev)).cpu().numpy(), table[iv])
        idx = rng.integers(0, 300, size=(77, 3), dtype=np.int64)
        for g in (2, 3, 5):
            got = ops.gather_mean(T(idx, dev), T(table, dev), g).cpu().numpy()
            assert bits_equal(got, oracle.gather_mean(idx, table, g))


@pytest.mark.parametrize("fused", ["2", "1", "0"])  # round 4: all planes + coalesced final (default) / ONE launch, the last workgroups finish / round 3's launches
@pytest.mark.parametrize("B,H,D", [(1, 8, 64), (300, 8, 64), (70000, 8, 64), (5000, 12, 64), (999, 3, 22), (4097, 40, 50), (1 << 20, 8, 64), (65536, 17, 128),
                                   (700, 8, 300), (333, 300, 512), (100, 1000, 64)])  # embedding_size > 256: windows of columns; hundreds of buckets
def test_lsh_backward_vs_oracle(B, H, D, fused, oracle, ops, dev, monkeypatch):
    """grad of (bits @ W)/popcount w.r.t. W: bit-exact against the oracle (same two-pass order), within 1e-5
    of torch autograd on the reference's op sequence (lsh_embedder.py:158,178), NaN rows as in the reference."""
    monkeypatch.setenv("MI_OOV_BWD_FUSED", fused)
    rng = np.random.default_rng(B + H)
    bits = (rng.random((B, H)) < 0.5).astype(np.uint8)
    bits[bits.sum(1) == 0, 0] = 1
    g = rng.standard_normal((B, D)).astype(np.float32)
    got = ops.lsh_embed_backward(T(bits, dev), T(g, dev)).cpu().numpy()
    assert bits_equal(got, oracle.lsh_embed_backward(bits, g))
    W = torch.zeros((H, D), requires_grad=True)
    bt = torch.from_numpy(bits).float()
    ((bt @ W) / bt.sum(1, keepdim=True) * torch.from_numpy(g)).sum().backward()
    assert np.abs(got - W.grad.numpy()).max() <= RTOL * np.abs(W.grad.numpy()).max()
    if B > 1:  # an all-zero code poisons the gradient exactly like the reference's autograd
        bits[B // 2] = 0
        got = ops.lsh_embed_backward(T(bits, dev), T(g, dev)).cpu().numpy()
        assert bits_equal(got, oracle.lsh_embed_backward(bits, g)) and np.isnan(got).all()


@pytest.mark.parametrize("fused", ["2", "1", "0"])
@pytest.mark.parametrize("B,nb,D", [(1, 8, 64), (3000, 9, 64), (70000, 64, 64), (5000, 777, 24), (4097, 65, 50), (65536, 9, 64), (300000, 33, 128), (900, 9, 300)])
def test_slsh_backward_and_scatter(B, nb, D, fused, oracle, ops, dev, monkeypatch):
    monkeypatch.setenv("MI_OOV_BWD_FUSED", fused)
    rng = np.random.default_rng(B + nb)
    idx = rng.integers(0, nb, B)
    if B > 100:
        idx[::13] = -1  # in-vocabulary rows of the training lookup: skipped
    g = rng.standard_normal((B, D)).astype(np.float32)
    got = ops.slsh_embed_backward(T(idx, dev), T(g, dev), nb).cpu().numpy()
    live = idx >= 0
    want = np.zeros((nb, D), np.float64)
    np.add.at(want, idx[live], g[live].astype(np.float64))
    assert np.abs(got - want).max() <= RTOL * np.abs(want).max()
    if nb <= 64:  # deterministic path: identical to the lsh backward on one-hot codes (a skipped row = a zero row somewhere:
        # adding +0 changes no bit, and the oracle's 0 / popcount-0 would be NaN where the kernel simply has no bucket)
        g0 = np.where(live[:, None], g, np.float32(0.0)).astype(np.float32)
        onehot = (np.where(live, idx, 0)[:, None] == np.arange(nb)[None]).astype(np.uint8)
        assert bits_equal(got, oracle.lsh_embed_backward(onehot, g0))
        assert bits_equal(got, ops.slsh_embed_backward(T(idx, dev), T(g, dev), nb).cpu().numpy())
    # generic scatter: out-of-range entries are skipped, accumulation goes into the given tensor
    idx2 = np.where(live, idx, 0)
    idx2[::7] = nb + 5
    idx2[1::11] = -1
    base = rng.standard_normal((nb, D)).astype(np.float32)
    out = ops.scatter_add_rows(T(idx2, dev), T(g, dev), nb, out=T(base, dev)).cpu().numpy()
    keep = (idx2 >= 0) & (idx2 < nb)
    want = base.astype(np.float64)
    np.add.at(want, idx2[keep], g[keep].astype(np.float64))
    assert np.abs(out - want).max() <= RTOL * np.abs(want).max()


@pytest.mark.parametrize("B,H,D,nb", [(4099, 3, 64, 8), (1000, 10, 64, 1000), (65, 27, 128, 100_000), (16, 32, 128, 40),
                                      (333, 8, 128, 5), (50, 9, 96, 77)])
def test_slsh_hot_tile_vs_oracle(B, H, D, nb, oracle, ops, dev):
    """slsh at F = 64 with up to 32 planes and D = 64 / 128 takes slsh64_kernel (planes eight at a time, bank-masked
    reduce); D = 96 stays on the generic kernel.  Bucket ids identical, rows verbatim, invalid ids -> -1 / NaN."""
    rng = np.random.default_rng(B + H)
    N = 3000
    feat = rng.standard_normal((N, 64), dtype=np.float32)
    feat[0] = 0
    planes = rng.standard_normal((H, 64), dtype=np.float32)
    big = rng.standard_normal((nb, D), dtype=np.float32)
    ids = rng.integers(0, N, size=B, dtype=np.int64)
    ids[0] = 0
    if B > 20:
        ids[7], ids[11] = N + 1, -5
    want, widx = oracle.slsh_embed(ids, feat, planes, big)
    got = ops.slsh_embed(T(ids, dev), T(feat, dev), T(planes, dev), T(big, dev)).cpu().numpy()
    assert bits_equal(got, want)
    assert np.array_equal(ops.slsh_index(T(ids, dev), T(feat, dev), T(planes, dev), nb).cpu().numpy(), widx)


def test_hot_kernel_chunked_launch_over_8m_lookups(ops, dev):
    """B above 2^23 lookups is split into several launches (32-bit row arithmetic inside the kernel): the result must
    equal the concatenation of two independent calls on the halves, for the score, row and slsh variants."""
    g = torch.Generator(device=dev).manual_seed(9)
    N, B = 100_000, (1 << 23) + 4099
    feat = torch.randn((N, 64), generator=g, device=dev)
    planes, buckets = torch.randn((8, 64), generator=g, device=dev), torch.randn((8, 64), generator=g, device=dev)
    ids = torch.randint(0, N, (B,), generator=g, device=dev)
    users = torch.randn((B, 64), generator=g, device=dev)
    h = B // 2 + 5
    s = ops.lsh_embed_score(ids, feat, planes, buckets, users)
    s2 = torch.cat((ops.lsh_embed_score(ids[:h], feat, planes, buckets, users[:h]),
                    ops.lsh_embed_score(ids[h:], feat, planes, buckets, users[h:].contiguous())))
    assert torch.equal(torch.nan_to_num(s), torch.nan_to_num(s2))
    del users, s, s2
    e = ops.lsh_embed(ids, feat, planes, buckets)
    assert torch.equal(torch.nan_to_num(e[h:]), torch.nan_to_num(ops.lsh_embed(ids[h:], feat, planes, buckets)))
    assert torch.equal(torch.nan_to_num(e[-3:]), torch.nan_to_num(ops.lsh_embed(ids[-3:], feat, planes, buckets)))
    del e
    big = torch.randn((1000, 64), generator=g, device=dev)
    p10 = torch.randn((10, 64), generator=g, device=dev)
    r = ops.slsh_embed(ids, feat, p10, big)
    assert torch.equal(r[h:], ops.slsh_embed(ids[h:], feat, p10, big))
    assert torch.equal(ops.slsh_index(ids, feat, p10, 1000)[-5:], ops.slsh_index(ids[-5:], feat, p10, 1000))


# ---- K batches in one persistent launch (mi_oov_lsh_embed_score_multi, csrc/lsh64p.hip) --------------------------
@pytest.mark.parametrize("K,B,N,H", [(1, 1, 7, 8), (1, 16, 9, 8), (3, 63, 100, 8), (2, 4097, 3000, 8), (7, 1000, 500, 8),
                                     (40, 333, 200, 8), (5, 65, 50, 1), (4, 130, 90, 3), (3, 257, 60, 5), (2, 48, 33, 7),
                                     (300, 17, 40, 8), (2, 70001, 5000, 8)])
def test_lsh_multi_vs_oracle(K, B, N, H, oracle, ops, dev):
    """Every batch of a multi-batch launch equals the oracle's lsh_embed_score of that batch, bit for bit: full and
    partial tiles, fewer tiles than resident waves and many more, batch boundaries inside a wave's stride,
    out-of-range ids (NaN score, never a fault), every plane count the persistent kernel serves."""
    rng = np.random.default_rng(K * 1000 + B)
    feat = rng.standard_normal((N, 64), dtype=np.float32)
    feat[0] = 0  # the padding row: all projections 0 -> all bits 1
    planes = rng.standard_normal((H, 64), dtype=np.float32)
    buckets = rng.standard_normal((H, 64), dtype=np.float32)
    ids = rng.integers(0, N, size=(K, B), dtype=np.int64)
    ids[0, 0] = 0
    if B > 5:
        ids[K - 1, 3] = N + 5
        ids[0, 4] = -1
    other = rng.standard_normal((K, B, 64), dtype=np.float32)
    feat_d, planes_d, buckets_d = T(feat, dev), T(planes, dev), T(buckets, dev)
    ids_d, other_d = T(ids, dev), T(other, dev)
    scores = ops.lsh_embed_score_multi([ids_d[k] for k in range(K)], feat_d, planes_d, buckets_d,
                                       [other_d[k] for k in range(K)])
    assert len(scores) == K
    for k in range(K):
        o_score, _ = oracle.lsh_embed_score(ids[k], feat, planes, buckets, other[k])
        assert bits_equal(scores[k].cpu().numpy(), o_score), f"batch {k}"


def test_lsh_multi_queue_slices_and_fallback(oracle, ops, dev):
    """A queue can be run in slices (bench.py launches its K steps in chunks); caller-owned score buffers are written
    and nothing else is; a shape the persistent kernel does not serve goes through K single launches."""
    rng = np.random.default_rng(5)
    K, B, N = 9, 200, 300
    feat = rng.standard_normal((N, 64), dtype=np.float32)
    planes = rng.standard_normal((8, 64), dtype=np.float32)
    buckets = rng.standard_normal((8, 64), dtype=np.float32)
    ids = rng.integers(0, N, size=(K, B), dtype=np.int64)
    other = rng.standard_normal((K, B, 64), dtype=np.float32)
    ids_d, other_d = T(ids, dev), T(other, dev)
    block = torch.full((K, B + 8), -7.0, dtype=torch.float32, device=dev)
    outs = [block[k, :B] for k in range(K)]
    q = ops.LshBatchQueue([ids_d[k] for k in range(K)], [other_d[k] for k in range(K)], outs)
    scorer = ops.LshMultiScorer(T(feat, dev), T(planes, dev), T(buckets, dev))
    assert scorer.persistent
    scorer.run(q, 2, 4)
    got = block.cpu().numpy()
    assert (got[:, B:] == -7.0).all() and (got[:2, :B] == -7.0).all() and (got[6:, :B] == -7.0).all()
    for k in range(2, 6):
        assert bits_equal(got[k, :B], oracle.lsh_embed_score(ids[k], feat, planes, buckets, other[k])[0])
    scorer.run(q)
    got = block.cpu().numpy()
    for k in range(K):
        assert bits_equal(got[k, :B], oracle.lsh_embed_score(ids[k], feat, planes, buckets, other[k])[0])
    with pytest.raises(ValueError):
        scorer.run(q, 5, 5)
    block.fill_(-7.0)
    launch = scorer.bind(q, 3, 4)  # a prevalidated launch of batches 3..6 (what bench.py issues)
    launch()
    launch()
    got = block.cpu().numpy()
    assert (got[:3, :B] == -7.0).all() and (got[7:, :B] == -7.0).all()
    for k in range(3, 7):
        assert bits_equal(got[k, :B], oracle.lsh_embed_score(ids[k], feat, planes, buckets, other[k])[0])
    with pytest.raises(ValueError):
        scorer.bind(q, 8, 2)
    # 12 planes: not a persistent-kernel shape
    planes12 = rng.standard_normal((12, 64), dtype=np.float32)
    buckets12 = rng.standard_normal((12, 64), dtype=np.float32)
    s12 = ops.LshMultiScorer(T(feat, dev), T(planes12, dev), T(buckets12, dev))
    assert not s12.persistent
    with pytest.raises(ValueError):
        s12.bind(ops.LshBatchQueue([ids_d[k] for k in range(3)], [other_d[k] for k in range(3)]))
    res = s12.run(ops.LshBatchQueue([ids_d[k] for k in range(3)], [other_d[k] for k in range(3)]))
    for k in range(3):
        assert bits_equal(res[k].cpu().numpy(), oracle.lsh_embed_score(ids[k], feat, planes12, buckets12, other[k])[0])


def _multi_case(rng, K, B, N, dev):
    ids = rng.integers(0, N, size=(K, B), dtype=np.int64)
    other = rng.standard_normal((K, B, 64), dtype=np.float32)
    return T(ids, dev), T(other, dev)


def test_lsh_multi_tile_pool_concurrent_streams_and_recycling(ops, dev):
    """The persistent kernel deals part of its tiles from a ticket pool (csrc/lsh64p.hip): a counter set serves one
    launch at a time and is handed back by the launch's last wave.  Launches in flight on four streams at once, and 3x
    more launches than there are counter sets, all give the per-batch kernel's scores bit for bit."""
    rng = np.random.default_rng(77)
    K, B, N = 6, 20000, 4000     # 7500 tiles: one static pair per wave of 2048, the rest from the pool
    feat = T(rng.standard_normal((N, 64), dtype=np.float32), dev)
    planes = T(rng.standard_normal((8, 64), dtype=np.float32), dev)
    buckets = T(rng.standard_normal((8, 64), dtype=np.float32), dev)
    scorer = ops.LshMultiScorer(feat, planes, buckets)
    cases = [_multi_case(rng, K, B, N, dev) for _ in range(8)]
    want = [[ops.lsh_embed_score(ids[k], feat, planes, buckets, other[k]) for k in range(K)] for ids, other in cases]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
    queues = []
    for rnd in range(24):                      # 96 launches > 32 counter sets
        for si, st in enumerate(streams):
            ids, other = cases[(rnd + 2 * si) % 8]
            with torch.cuda.stream(st):
                q = ops.LshBatchQueue([ids[k] for k in range(K)], [other[k] for k in range(K)])
                scorer.run(q)
            queues.append(((rnd + 2 * si) % 8, q))
    torch.cuda.synchronize()
    for ci, q in queues:
        for k in range(K):
            assert torch.equal(torch.nan_to_num(q.scores[k]), torch.nan_to_num(want[ci][k]))


def test_lsh_multi_captured_launch_deals_tiles_statically(ops, dev):
    """A launch captured into a HIP graph gets no counter set (the host cannot see its replays): the kernel deals the
    same tickets in a fixed order.  Replays with new ids in place equal the eager results."""
    rng = np.random.default_rng(78)
    K, B, N = 5, 30011, 6000
    feat = T(rng.standard_normal((N, 64), dtype=np.float32), dev)
    planes = T(rng.standard_normal((8, 64), dtype=np.float32), dev)
    buckets = T(rng.standard_normal((8, 64), dtype=np.float32), dev)
    scorer = ops.LshMultiScorer(feat, planes, buckets)
    ids, other = _multi_case(rng, K, B, N, dev)
    q = ops.LshBatchQueue([ids[k] for k in range(K)], [other[k] for k in range(K)])
    scorer.run(q)                              # warm-up outside the capture (occupancy query, LDS attribute)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        scorer.run(q)
    for rep in range(3):
        new_ids, new_other = _multi_case(rng, K, B, N, dev)
        ids.copy_(new_ids)
        other.copy_(new_other)
        for s in q.scores:
            s.fill_(-1.0)
        g.replay()
        torch.cuda.synchronize()
        for k in range(K):
            eager = ops.lsh_embed_score(ids[k], feat, planes, buckets, other[k])
            assert torch.equal(torch.nan_to_num(q.scores[k]), torch.nan_to_num(eager)), (rep, k)


def test_lsh_multi_pool_disabled_process(dev):
    """MI_OOV_POOL=0 (read once per process): every launch deals its tiles in the fixed order; same scores."""
    import os
    import subprocess
    import sys
    code = (
        "import numpy as np, torch, mi_oov\n"
        "from mi_oov import ops\n"
        "rng = np.random.default_rng(3); dev = 'cuda:0'\n"
        "T = lambda a: torch.from_numpy(a).to(dev)\n"
        "for K, B, N in [(7, 20000, 4000), (1, 16, 9), (40, 333, 200), (3, 65536, 50000)]:\n"
        "    feat, planes, buckets = T(rng.standard_normal((N, 64), dtype=np.float32)), T(rng.standard_normal((8, 64), dtype=np.float32)), T(rng.standard_normal((8, 64), dtype=np.float32))\n"
        "    ids, other = T(rng.integers(0, N, size=(K, B), dtype=np.int64)), T(rng.standard_normal((K, B, 64), dtype=np.float32))\n"
        "    got = ops.lsh_embed_score_multi([ids[k] for k in range(K)], feat, planes, buckets, [other[k] for k in range(K)])\n"
        "    for k in range(K):\n"
        "        want = ops.lsh_embed_score(ids[k], feat, planes, buckets, other[k])\n"
        "        assert torch.equal(torch.nan_to_num(got[k]), torch.nan_to_num(want)), (K, B, k)\n"
        "print('POOL-OFF-OK')\n")
    env = dict(os.environ, MI_OOV_POOL="0", PYTHONPATH=os.pathsep.join(sys.path))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "POOL-OFF-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_lsh_multi_raw_cabi_errors(ops, dev):
    """Status codes of the raw entry point: unsupported shape, NULL table, misaligned table."""
    import ctypes
    from mi_oov import _cabi as C
    lib = C.lib()
    feat = torch.zeros((10, 64), device=dev)
    planes = torch.zeros((8, 64), device=dev)
    buckets = torch.zeros((8, 64), device=dev)
    tab = torch.zeros((3, 4), dtype=torch.int64, device=dev)
    st = C.stream_of(feat)
    fn = lib.mi_oov_lsh_embed_score_multi
    base, step = tab.data_ptr(), 32
    args = lambda F, H, D: (base, base + step, base + 2 * step, 1, 16, feat.data_ptr(), 10, F, planes.data_ptr(), H,  # noqa: E731
                            buckets.data_ptr(), D, st)
    assert fn(*args(32, 8, 64)) == -2 and fn(*args(64, 9, 64)) == -2 and fn(*args(64, 8, 128)) == -2   # MI_OOV_ERR_SHAPE
    assert fn(None, base, base, 1, 16, feat.data_ptr(), 10, 64, planes.data_ptr(), 8, buckets.data_ptr(), 64, st) == -1
    assert fn(base + 4, base, base, 1, 16, feat.data_ptr(), 10, 64, planes.data_ptr(), 8, buckets.data_ptr(), 64, st) == -5
    assert fn(base, base, base, 0, 16, feat.data_ptr(), 10, 64, planes.data_ptr(), 8, buckets.data_ptr(), 64, st) == 0  # K = 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("B", [262144, 300001])
def test_lsh_bits_large_batch_persistent_codes_kernel(B, oracle, ops, dev):
    """Codes-only calls of >= 262144 lookups (F = 64, H = 8) take the persistent, software-pipelined kernel
    (lsh64p.hip, MODE_CODES): same bits as the oracle, 0xFF rows for ids outside the table, partial last tile."""
    rng = np.random.default_rng(B)
    N = 50_000
    feat = rng.standard_normal((N, 64), dtype=np.float32)
    feat[0] = 0
    planes = rng.standard_normal((8, 64), dtype=np.float32)
    ids = rng.integers(0, N, size=B, dtype=np.int64)
    ids[0], ids[7], ids[B - 1], ids[B // 2] = 0, -1, N, N + 12345
    bits = ops.lsh_bits(T(ids, dev), T(feat, dev), T(planes, dev)).cpu().numpy()
    _, o_bits = oracle.lsh_embed(ids, feat, planes, np.zeros((8, 1), np.float32), want_bits=True)
    assert bits.shape == (B, 8) and np.array_equal(bits, o_bits)
    assert (bits[7] == 0xFF).all() and (bits[B - 1] == 0xFF).all() and (bits[0] == 1).all()


# ---- round 3: rows / lookup modes of the persistent launch, prepared table of aggregates (mi_oov_lsh_multi) ------------
@pytest.mark.parametrize("K,B,N,H", [(1, 1, 7, 8), (3, 63, 100, 8), (2, 4097, 3000, 8), (40, 333, 200, 8), (5, 65, 50, 1),
                                     (4, 130, 90, 3), (3, 257, 60, 5), (2, 48, 33, 7), (300, 17, 40, 8), (2, 70001, 5000, 8)])
@pytest.mark.parametrize("prepared", [False, True])
def test_lsh_multi_rows_and_lookup_modes_vs_oracle(K, B, N, H, prepared, oracle, ops, dev):
    """The four per-batch calls queued K times in one persistent launch, with the table of aggregates built inside the
    launch or prepared once: rows (embed_*_ids' [B, 64] output), lookup rows (BPR.get_*_embedding: in-vocabulary rows
    verbatim), scores and lookup scores -- every batch bit-identical to the oracle; ids outside the tables give NaN."""
    rng = np.random.default_rng(K * 1000 + B + H)
    n_vocab = max(1, N // 3)
    feat = rng.standard_normal((N, 64), dtype=np.float32)
    feat[0] = 0
    vtab = rng.standard_normal((n_vocab, 64), dtype=np.float32)
    planes = rng.standard_normal((H, 64), dtype=np.float32)
    buckets = rng.standard_normal((H, 64), dtype=np.float32)
    ids = rng.integers(0, N, size=(K, B), dtype=np.int64)
    ids[0, 0] = 0
    if B > 5:
        ids[K - 1, 3], ids[0, 4] = N + 5, -1
    other = rng.standard_normal((K, B, 64), dtype=np.float32)
    feat_d, planes_d, buckets_d, vtab_d = T(feat, dev), T(planes, dev), T(buckets, dev), T(vtab, dev)
    ids_d, other_d = T(ids, dev), T(other, dev)
    il, ol = [ids_d[k] for k in range(K)], [other_d[k] for k in range(K)]
    tab = ops.LshTable(buckets_d) if prepared else None
    if prepared:
        assert tab.get() is not None and tab.get().shape == (1 << H, 64)
    rows = ops.lsh_embed_multi(il, feat_d, planes_d, buckets_d, table=tab)
    lrows = ops.lsh_lookup_multi(il, vtab_d, feat_d, planes_d, buckets_d, lsh_table=tab)
    lscore = ops.lsh_lookup_multi(il, vtab_d, feat_d, planes_d, buckets_d, other_list=ol, lsh_table=tab)
    sc = ops.LshMultiScorer(feat_d, planes_d, buckets_d, prepared=prepared).run(ops.LshBatchQueue(il, ol))
    for k in range(K):
        o_rows = oracle.lsh_embed(ids[k], feat, planes, buckets)
        assert bits_equal(rows[k].cpu().numpy(), o_rows), f"rows, batch {k}"
        o_lrows = oracle.lsh_lookup(ids[k], vtab, feat, planes, buckets)
        got = lrows[k].cpu().numpy()
        assert bits_equal(got, o_lrows), f"lookup rows, batch {k}"
        iv = (ids[k] >= 0) & (ids[k] < n_vocab)
        assert bits_equal(got[iv], vtab[ids[k][iv]])
        assert bits_equal(lscore[k].cpu().numpy(), oracle.rowdot(other[k], o_lrows)), f"lookup score, batch {k}"
        assert bits_equal(sc[k].cpu().numpy(), oracle.lsh_embed_score(ids[k], feat, planes, buckets, other[k])[0]), f"score, batch {k}"
    if B > 5:
        assert np.isnan(rows[K - 1][3].cpu().numpy()).all() and np.isnan(lrows[0][4].cpu().numpy()).all()


def test_lsh_table_prepared_equals_per_lookup_arithmetic_and_tracks_updates(oracle, ops, dev):
    """mi_oov_lsh_table_prepare: row c of the table is the embedding of code c -- the oracle's lsh_embed of a feature row
    whose projections have exactly those signs -- for every H the persistent kernel takes; row 0 is the reference's NaN
    row; a non-finite bucket weight poisons exactly the rows the per-lookup chain poisons; and ops.LshTable re-prepares
    after an in-place update of the bucket tensor (an optimizer step)."""
    rng = np.random.default_rng(11)
    for H in range(1, 9):
        buckets = rng.standard_normal((H, 64), dtype=np.float32)
        if H == 5:
            buckets[2, 7] = np.inf
        if H == 6:
            buckets[0, :] *= 1e-38  # subnormal quotients take the IEEE division branch
        b_d = T(buckets, dev)
        tab = ops.LshTable(b_d)
        got = tab.get().cpu().numpy()
        # a 64-wide feature row e_h projects onto plane h alone: planes = +-identity rows give every code
        planes = np.zeros((H, 64), np.float32)
        planes[np.arange(H), np.arange(H)] = 1.0
        codes = np.arange(1 << H)
        feat = np.full((1 << H, 64), 0.0, np.float32)
        feat[:, :H] = np.where((codes[:, None] >> np.arange(H)) & 1, 1.0, -1.0)
        want = oracle.lsh_embed(codes.astype(np.int64), feat, planes, buckets)
        assert bits_equal(got, want), f"H = {H}"
        assert np.isnan(got[0]).all()
        b_d.mul_(2.0)  # in place: the version counter moves
        got2 = tab.get().cpu().numpy()
        want2 = oracle.lsh_embed(codes.astype(np.int64), feat, planes, (buckets * np.float32(2.0)).astype(np.float32))
        assert bits_equal(got2, want2), f"H = {H} after an in-place update"
    assert ops.LshTable(torch.zeros((9, 64), device=dev)).get() is None
    assert ops.LshTable(torch.zeros((8, 32), device=dev)).get() is None
    # a Parameter re-pointed at other memory keeps its version counter: the table follows the address
    P = torch.nn.Parameter(T(buckets, dev))
    tab = ops.LshTable(P)
    t0 = tab.get().clone()
    P.data = (P.data * 2.0).contiguous()
    assert not torch.equal(torch.nan_to_num(tab.get()), torch.nan_to_num(t0))


def test_reinitialising_a_model_after_a_forward_call_is_seen(dev):
    """the package's initialiser used to write through `.data`, which moves no version counter -- a sweep that
    re-initialises a model after its first inference call kept scoring with the old split weights / the old table of
    aggregates.  It now writes on the Parameter (under no_grad): the counter moves and both caches re-make themselves."""
    import mi_oov
    from mi_oov import ops
    from mi_oov.model import xavier_normal_initialization
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(32, 48), torch.nn.GELU(), torch.nn.Linear(48, 16), torch.nn.Sigmoid()).to(dev)
    x = torch.randn((70, 32), device=dev)
    with torch.no_grad():
        y0 = ops.hash_net_forward(net, x)
        net.apply(xavier_normal_initialization)
        y1 = ops.hash_net_forward(net, x)
        ref = torch.sigmoid(torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(x, net[0].weight, net[0].bias)),
                                                       net[2].weight, net[2].bias))
    assert not torch.equal(y0, y1) and float((y1 - ref).abs().max()) < 1e-5
    emb = torch.nn.Embedding(8, 64).to(dev)
    tab = ops.LshTable(emb.weight)
    t0 = tab.get().clone()
    emb.apply(xavier_normal_initialization)
    assert not torch.equal(torch.nan_to_num(tab.get()), torch.nan_to_num(t0))


@pytest.mark.parametrize("B", [524288 + 37, 600000])
def test_lsh_large_single_batch_takes_the_persistent_kernel(B, oracle, ops, dev):
    """One call of >= 524288 scored lookups (F = D = 64, H <= 8) runs on the persistent kernel with a single batch
    (TAB = false): lsh_embed_score and lsh_lookup_score, bit-identical to the oracle, partial last tile, invalid ids."""
    rng = np.random.default_rng(B)
    N, n_vocab, H = 40_000, 15_000, 8
    feat = rng.standard_normal((N, 64), dtype=np.float32)
    vtab = rng.standard_normal((n_vocab, 64), dtype=np.float32)
    planes = rng.standard_normal((H, 64), dtype=np.float32)
    buckets = rng.standard_normal((H, 64), dtype=np.float32)
    ids = rng.integers(0, N, size=B, dtype=np.int64)
    ids[0], ids[7], ids[B - 1] = 0, -1, N
    other = rng.standard_normal((B, 64), dtype=np.float32)
    f, p, w, v, o, i = T(feat, dev), T(planes, dev), T(buckets, dev), T(vtab, dev), T(other, dev), T(ids, dev)
    want, _ = oracle.lsh_embed_score(ids, feat, planes, buckets, other)
    assert bits_equal(ops.lsh_embed_score(i, f, p, w, o).cpu().numpy(), want)
    lrows = oracle.lsh_lookup(ids, vtab, feat, planes, buckets)
    assert bits_equal(ops.lsh_lookup_score(i, v, f, p, w, o).cpu().numpy(), oracle.rowdot(other, lrows))


def test_lsh_multi_generic_entry_raw_cabi_errors(ops, dev):
    """Status codes of mi_oov_lsh_multi / mi_oov_lsh_table_prepare: unknown mode, missing tables, shapes, alignment."""
    from mi_oov import _cabi as C
    lib = C.lib()
    feat = torch.zeros((10, 64), device=dev)
    planes = torch.zeros((8, 64), device=dev)
    buckets = torch.zeros((8, 64), device=dev)
    table = torch.zeros((256, 64), device=dev)
    tab = torch.zeros((3, 4), dtype=torch.int64, device=dev)
    st = C.stream_of(feat)
    fn, base = lib.mi_oov_lsh_multi, tab.data_ptr()
    common = (feat.data_ptr(), 10, 64, planes.data_ptr(), 8)
    assert fn(7, base, base, base, 1, 16, None, 0, *common, buckets.data_ptr(), 64, None, st) == -3       # MI_OOV_ERR_KIND
    assert fn(0, base, None, base, 1, 16, None, 0, *common, buckets.data_ptr(), 64, None, st) == -1       # score mode needs rows
    assert fn(1, base, None, base, 1, 16, None, 0, *common, None, 64, None, st) == -1                     # neither buckets nor table
    assert fn(1, base, None, base, 1, 16, None, 0, *common, None, 64, table.data_ptr() + 4, st) == -5     # misaligned table
    assert fn(1, base, None, base, 1, 16, None, 0, feat.data_ptr(), 10, 32, planes.data_ptr(), 8, buckets.data_ptr(), 64, None, st) == -2
    assert fn(1, base, None, base, 0, 16, None, 0, *common, buckets.data_ptr(), 64, None, st) == 0        # K = 0
    assert lib.mi_oov_lsh_table_bytes(8, 64) == 65536 and lib.mi_oov_lsh_table_bytes(9, 64) == 0 and lib.mi_oov_lsh_table_bytes(8, 128) == 0
    assert lib.mi_oov_lsh_table_prepare(buckets.data_ptr(), 9, 64, table.data_ptr(), st) == -2
    assert lib.mi_oov_lsh_table_prepare(None, 8, 64, table.data_ptr(), st) == -1
    assert lib.mi_oov_lsh_table_prepare(buckets.data_ptr(), 8, 64, table.data_ptr() + 8, st) == -5
    assert lib.mi_oov_init() == 0 and lib.mi_oov_init() == 0  # idempotent
    torch.cuda.synchronize()


@pytest.mark.parametrize("K,B,N,D", [(1, 1, 5, 64), (3, 63, 100, 64), (5, 4097, 3000, 64), (40, 333, 200, 128), (7, 130, 90, 32),
                                     (2, 17, 40, 50), (300, 17, 40, 64)])
def test_gather_multi_entries_vs_oracle(K, B, N, D, oracle, ops, dev):
    """K queued batches of gather_rows / gather_mean in one launch: every batch equals the oracle's result for that batch
    (and the single-batch kernels), invalid ids give NaN rows, odd index counts keep torch.split's short last group."""
    rng = np.random.default_rng(K * 100 + B + D)
    W = rng.standard_normal((N, D), dtype=np.float32)
    ids = rng.integers(0, N, size=(K, B), dtype=np.int64)
    if B > 5:
        ids[0, 2], ids[K - 1, 4] = -1, N
    W_d, ids_d = T(W, dev), T(ids, dev)
    rows = ops.gather_rows_multi([ids_d[k] for k in range(K)], W_d)
    for k in range(K):
        assert bits_equal(rows[k].cpu().numpy(), oracle.gather_rows(ids[k], W)), f"gather_rows, batch {k}"
        assert bits_equal(ops.gather_rows(ids_d[k], W_d).cpu().numpy(), oracle.gather_rows(ids[k], W))
    for g in (2, 3):
        for M in (B, B - 1 if B > 1 else 1):  # an odd count leaves a short last group
            idx = [ids_d[k][:M] for k in range(K)]
            means = ops.gather_mean_multi(idx, W_d, g)
            for k in range(K):
                want = oracle.gather_mean(ids[k][:M], W, g)
                assert bits_equal(means[k].cpu().numpy(), want), f"gather_mean g={g} M={M}, batch {k}"
                assert bits_equal(ops.gather_mean(idx[k], W_d, g).cpu().numpy(), want)


@pytest.mark.parametrize("K,B,H,D,nb", [(1, 1, 3, 64, 8), (3, 4099, 3, 64, 8), (4, 1000, 10, 64, 1000), (2, 65, 27, 128, 100_000),
                                        (5, 16, 32, 128, 40), (3, 300, 8, 64, 5), (2, 77, 12, 32, 100)])
def test_slsh_multi_and_lds_bucket_rows_vs_oracle(K, B, H, D, nb, oracle, ops, dev):
    """slsh on the hot tile reads its bucket rows from an LDS copy of the H + 1 rows the reference's arithmetic can reach
    ((bits_req + popcount) % n_buckets): rows and bucket ids of single launches and of K queued batches equal the
    oracle's, for bucket tables smaller than, equal to and far larger than that set, D = 64 / 128 (D = 32: fallback)."""
    rng = np.random.default_rng(K + B + H + nb)
    N = 500
    feat = rng.standard_normal((N, 64), dtype=np.float32)
    feat[0] = 0
    planes = rng.standard_normal((H, 64), dtype=np.float32)
    buckets = rng.standard_normal((nb, D), dtype=np.float32)
    ids = rng.integers(0, N, size=(K, B), dtype=np.int64)
    ids[0, 0] = 0
    if B > 5:
        ids[0, 3], ids[K - 1, 1] = N + 2, -7
    f, p, w, i = T(feat, dev), T(planes, dev), T(buckets, dev), T(ids, dev)
    rows, idxs = ops.slsh_embed_multi([i[k] for k in range(K)], f, p, w, want_idx=True)
    rows_only = ops.slsh_embed_multi([i[k] for k in range(K)], f, p, w)
    for k in range(K):
        o_emb, o_idx = oracle.slsh_embed(ids[k], feat, planes, buckets)
        assert bits_equal(rows[k].cpu().numpy(), o_emb) and bits_equal(rows_only[k].cpu().numpy(), o_emb), f"batch {k}"
        assert np.array_equal(idxs[k].cpu().numpy(), o_idx)
        assert bits_equal(ops.slsh_embed(i[k], f, p, w).cpu().numpy(), o_emb)
        assert np.array_equal(ops.slsh_index(i[k], f, p, nb).cpu().numpy(), o_idx)


@pytest.mark.parametrize("H,D,nb", [(3, 64, 8), (27, 128, 100_000), (24, 64, 20_000)])
def test_slsh_pipelined_waves_equal_one_tile_per_wave_launches(H, D, nb, oracle, ops, dev):
    """slsh64_kernel keeps the rows of a wave's next tile and the ids of the one after in flight (round 4): launches whose
    waves walk 4-5 tiles each (12 queued batches of 50001 lookups; one call over their concatenation) against launches
    whose waves have a single tile (one batch per call: no pipeline), ragged last tiles and invalid ids included; batch 0
    against the oracle."""
    g = torch.Generator(device=dev).manual_seed(H + D)
    N, K, B = 20_000, 12, 50_001
    feat = torch.randn((N, 64), generator=g, device=dev)
    planes = torch.randn((H, 64), generator=g, device=dev)
    big = torch.randn((nb, D), generator=g, device=dev)
    ids = [torch.randint(0, N, (B,), generator=g, device=dev) for _ in range(K)]
    for k in range(K):
        ids[k][k::997] = N + k
        ids[k][B - 1 - k] = -1

    def same(a, b):
        return torch.equal(torch.nan_to_num(a, 7.0), torch.nan_to_num(b, 7.0))

    single = [ops.slsh_embed(i, feat, planes, big) for i in ids]
    single_idx = [ops.slsh_index(i, feat, planes, nb) for i in ids]
    rows, idx = ops.slsh_embed_multi(ids, feat, planes, big, want_idx=True)
    rows_only = ops.slsh_embed_multi(ids, feat, planes, big)
    for k in range(K):
        assert same(rows[k], single[k]) and same(rows_only[k], single[k]), k
        assert torch.equal(idx[k], single_idx[k]), k
    flat = torch.cat(ids)
    assert same(ops.slsh_embed(flat, feat, planes, big), torch.cat(single))
    assert torch.equal(ops.slsh_index(flat, feat, planes, nb), torch.cat(single_idx))
    want, widx = oracle.slsh_embed(ids[0].cpu().numpy(), feat.cpu().numpy(), planes.cpu().numpy(), big.cpu().numpy())
    assert bits_equal(single[0].cpu().numpy(), want) and np.array_equal(single_idx[0].cpu().numpy(), widx)
    assert int((single_idx[0] == -1).sum()) >= 50


@pytest.mark.parametrize("B,N,D,k,skip", [(300, 128 * 40 + 1, 64, 20, 1), (64, 128 * 129 + 1, 64, 5, 0), (130, 50_000, 22, 20, 1),
                                          (65, 128 * 9 + 127, 64, 2, 3)])
@pytest.mark.parametrize("poison", [0xFF, 0x7F, 0x00])
def test_fused_topk_reads_nothing_it_has_not_written(B, N, D, k, skip, poison, oracle, ops, dev):
    """The fused top-k's kernels form addresses from counters and candidate entries they keep in the caller's workspace
    (list lengths, overflow counts, column numbers).  Whatever the workspace held before the call -- here every byte
    0xFF / 0x7F / 0x00 -- the result is the oracle's: each stage overwrites what the next one reads, on shapes whose last
    128-row block of the bf16 catalogue copy is one row or 127 rows full (the padding rows are written too).  The raw C
    ABI is called so that the workspace is the test's own buffer.  (The finalize kernel additionally clamps every count
    and column it reads: DESIGN.md section 5a, "the fault of record".)"""
    from mi_oov import _cabi as C
    lib = C.lib()
    rng = np.random.default_rng(B + N + k)
    U = rng.standard_normal((B, D), dtype=np.float32)
    E = rng.standard_normal((N, D), dtype=np.float32)
    U_d, E_d = T(U, dev), T(E, dev)
    need = int(lib.mi_oov_score_topk_workspace(B, N, k))
    ws = torch.full((need + 4096,), poison, dtype=torch.uint8, device=dev)
    vals = torch.empty((B, k), dtype=torch.float32, device=dev)
    idx = torch.empty((B, k), dtype=torch.int64, device=dev)
    rc = lib.mi_oov_score_topk(U_d.data_ptr(), B, E_d.data_ptr(), N, D, k, skip, vals.data_ptr(), idx.data_ptr(), ws.data_ptr(),
                               C.stream_of(U_d))
    assert rc == 0
    torch.cuda.synchronize()
    assert (ws[need:] == poison).all(), "the call wrote past the workspace size it asked for"
    o_vals, o_idx = oracle.score_topk(U, E, k, skip)
    assert np.array_equal(idx.cpu().numpy(), o_idx) and bits_equal(vals.cpu().numpy(), o_vals)
