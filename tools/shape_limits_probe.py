#!/usr/bin/env python3
"""Which shapes the reference accepts does the library refuse (MI_OOV_ERR_SHAPE) or get wrong?  Developer probe, GPU box:
every op on a few extreme shapes against the oracle."""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402

import mi_oov  # noqa: E402,F401
import oov_oracle as oracle  # noqa: E402
from mi_oov import ops  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind == "f":
        return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)) or (
            np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a), np.nan_to_num(b)))
    return np.array_equal(a, b)


def case(name, fn):
    try:
        ok = fn()
        print(f"{'ok  ' if ok else 'DIFF'} {name}", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"FAIL {name}: {type(e).__name__}: {str(e)[:140]}", flush=True)
        if os.environ.get("PROBE_TRACE"):
            traceback.print_exc()


def lsh(B, N, F, H, D):
    feat, planes, W = (rng.standard_normal(s, dtype=np.float32) for s in ((N, F), (H, F), (H, D)))
    ids = rng.integers(0, N, size=B, dtype=np.int64)
    o_emb, o_bits = oracle.lsh_embed(ids, feat, planes, W, want_bits=True)
    other = rng.standard_normal((B, D), dtype=np.float32)
    table = rng.standard_normal((N // 2, D), dtype=np.float32)
    o_score, _ = oracle.lsh_embed_score(ids, feat, planes, W, other)
    emb_b, bits_b = ops._lsh_forward(d(ids), d(feat), d(planes), d(W), want_bits=True)
    return same(ops.lsh_embed(d(ids), d(feat), d(planes), d(W)).cpu().numpy(), o_emb) and \
        same(ops.lsh_bits(d(ids), d(feat), d(planes)).cpu().numpy(), o_bits) and \
        same(emb_b.cpu().numpy(), o_emb) and same(bits_b.cpu().numpy(), o_bits) and \
        same(ops.lsh_embed_score(d(ids), d(feat), d(planes), d(W), d(other)).cpu().numpy(), o_score) and \
        same(ops.lsh_lookup(d(ids), d(table), d(feat), d(planes), d(W)).cpu().numpy(), oracle.lsh_lookup(ids, table, feat, planes, W)) and \
        same(ops.lsh_lookup_score(d(ids), d(table), d(feat), d(planes), d(W), d(other)).cpu().numpy(),
             oracle.rowdot(other, oracle.lsh_lookup(ids, table, feat, planes, W)))


def slsh(B, N, F, nb, D):
    H = int(np.ceil(np.log2(nb)))
    feat, planes, W = (rng.standard_normal(s, dtype=np.float32) for s in ((N, F), (H, F), (nb, D)))
    ids = rng.integers(0, N, size=B, dtype=np.int64)
    o_emb, o_idx = oracle.slsh_embed(ids, feat, planes, W)
    return same(ops.slsh_embed(d(ids), d(feat), d(planes), d(W)).cpu().numpy(), o_emb) and \
        same(ops.slsh_index(d(ids), d(feat), d(planes), nb).cpu().numpy(), o_idx)


def gmean(B, N, D, k):
    W = rng.standard_normal((N, D), dtype=np.float32)
    idx = rng.integers(0, N, size=B * k, dtype=np.int64)
    return same(ops.gather_mean(d(idx), d(W), 2).cpu().numpy(), oracle.gather_mean(idx, W, 2))


def topk(B, N, D, k):
    U, E = rng.standard_normal((B, D), dtype=np.float32), rng.standard_normal((N, D), dtype=np.float32)
    v, i = ops.score_topk(d(U), d(E), k, 0)
    ov, oi = oracle.score_topk(U, E, k, 0)
    return same(i.cpu().numpy(), oi) and same(v.cpu().numpy(), ov)


def rowdot(B, D):
    a, b = rng.standard_normal((B, D), dtype=np.float32), rng.standard_normal((B, D), dtype=np.float32)
    return same(ops.rowdot(d(a), d(b)).cpu().numpy(), oracle.rowdot(a, b))


for args in ((100, 50, 64, 8, 300), (100, 50, 64, 8, 512), (100, 50, 64, 8, 1024), (100, 50, 300, 100, 64), (100, 50, 300, 400, 64),
             (100, 50, 1000, 16, 64), (100, 50, 2000, 30, 16), (100, 50, 1, 1, 1), (100, 50, 64, 2000, 64), (100, 50, 7, 5000, 50),
             (100, 50, 768, 1000, 64), (100, 50, 770, 300, 300), (100, 50, 64, 8, 257), (100, 50, 22, 40, 1030)):
    case(f"lsh   B,N,F,H,D = {args}", lambda a=args: lsh(*a))
for args in ((100, 50, 64, 8, 300), (100, 50, 64, 100_000, 512), (100, 50, 1000, 1 << 20, 64), (100, 50, 3000, 1 << 30, 16), (100, 50, 5, 2, 7), (100, 50, 5, 1, 7)):
    case(f"slsh  B,N,F,nb,D = {args}", lambda a=args: slsh(*a))
for args in ((100, 50, 64, 2), (100, 50, 300, 2), (100, 50, 1000, 5), (100, 50, 3, 50), (100, 50, 64, 1)):
    case(f"gather_mean B,N,D,k = {args}", lambda a=args: gmean(*a))
for args in ((10, 500, 64, 100), (10, 500, 64, 500), (10, 500, 200, 20), (10, 500, 700, 20), (10, 50, 64, 64), (3, 70000, 64, 300), (1, 1, 1, 1)):
    case(f"score_topk B,N,D,k = {args}", lambda a=args: topk(*a))
for args in ((100, 300), (100, 1000), (100, 1), (1, 4097)):
    case(f"rowdot B,D = {args}", lambda a=args: rowdot(*a))
