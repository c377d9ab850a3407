"""ctypes front-end of oracle/liboov_oracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
Inputs and outputs are numpy arrays (C-contiguous); see oov_oracle.c for the reference
file:line each function restates.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboov_oracle.so")

HASH_KINDS = {"mod": 0, "fast": 1, "3round": 2, "64bit": 3}


def build(force=False):
    """Compile the oracle with gcc (called by __graft_entry__.build())."""
    src = os.path.join(_HERE, "oov_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oov_siphash24.restype = ctypes.c_uint64
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


_c = ctypes.c_int64


def set_threads(n):
    lib().oov_set_threads(int(n))


def max_threads():
    return int(lib().oov_get_max_threads())


def lsh_embed(ids, feat, planes, buckets=None, want_bits=False):
    ids, feat, planes = _i64(ids), _f32(feat), _f32(planes)
    B, (N, F), H = ids.shape[0], feat.shape, planes.shape[0]
    out = bits = None
    D = 0
    if buckets is not None:
        buckets = _f32(buckets)
        D = buckets.shape[1]
        out = np.empty((B, D), np.float32)
    if want_bits:
        bits = np.empty((B, H), np.uint8)
    lib().oov_lsh_embed(_p(ids), _c(B), _p(feat), _c(N), _c(F), _p(planes), _c(H), _p(buckets), _c(D), _p(out), _p(bits))
    if want_bits:
        return out, bits
    return out


def lsh_embed_score(ids, feat, planes, buckets, other, want_emb=True):
    ids, feat, planes, buckets, other = _i64(ids), _f32(feat), _f32(planes), _f32(buckets), _f32(other)
    B, (N, F), H, D = ids.shape[0], feat.shape, planes.shape[0], buckets.shape[1]
    score = np.empty((B,), np.float32)
    out = np.empty((B, D), np.float32) if want_emb else None
    lib().oov_lsh_embed_score(_p(ids), _c(B), _p(feat), _c(N), _c(F), _p(planes), _c(H), _p(buckets), _c(D),
                              _p(other), _p(score), _p(out))
    return score, out


def lsh_lookup(ids, table, feat, planes, buckets):
    ids, table, feat, planes, buckets = _i64(ids), _f32(table), _f32(feat), _f32(planes), _f32(buckets)
    B, (N, F), H, D = ids.shape[0], feat.shape, planes.shape[0], buckets.shape[1]
    out = np.empty((B, D), np.float32)
    lib().oov_lsh_lookup(_p(ids), _c(B), _p(table), _c(table.shape[0]), _p(feat), _c(N), _c(F), _p(planes), _c(H),
                         _p(buckets), _c(D), _p(out))
    return out


def slsh_embed(ids, feat, planes, buckets):
    ids, feat, planes, buckets = _i64(ids), _f32(feat), _f32(planes), _f32(buckets)
    B, (N, F), H, (nb, D) = ids.shape[0], feat.shape, planes.shape[0], buckets.shape
    out = np.empty((B, D), np.float32)
    idx = np.empty((B,), np.int64)
    lib().oov_slsh_embed(_p(ids), _c(B), _p(feat), _c(N), _c(F), _p(planes), _c(H), _p(buckets), _c(nb), _c(D),
                         _p(out), _p(idx))
    return out, idx


def siphash24(key: bytes, msg: bytes) -> int:
    k = (ctypes.c_uint8 * 16).from_buffer_copy(key)
    m = (ctypes.c_uint8 * max(1, len(msg))).from_buffer_copy(msg.ljust(1, b"\0"))
    return int(lib().oov_siphash24(k, m, _c(len(msg))))


def siphash24_mod(ids, keys, mod=16777216):
    ids = _i64(ids)
    keys = np.ascontiguousarray(keys, dtype=np.uint8).reshape(-1, 16)
    B, K = ids.shape[0], keys.shape[0]
    out = np.empty((B, K), np.float32)
    lib().oov_siphash24_mod(_p(ids), _c(B), _p(keys), _c(K), ctypes.c_uint32(mod), _p(out))
    return out


def mapper_hash(ids, kind):
    ids = _i64(ids)
    out = np.empty_like(ids)
    rc = lib().oov_mapper_hash(_p(ids), _c(ids.shape[0]), ctypes.c_int(HASH_KINDS[kind]), _p(out))
    assert rc == 0
    return out


def mapper_map(ids, kind, n_orig, n_buckets):
    ids = _i64(ids)
    out = np.empty_like(ids)
    rc = lib().oov_mapper_map(_p(ids), _c(ids.shape[0]), ctypes.c_int(HASH_KINDS[kind]), _c(n_orig), _c(n_buckets),
                              _p(out))
    assert rc == 0
    return out


def bucket_by_owner(ids, n_rows, per, world, cap):
    """-> (send int64[world, cap], slot int32[B], counts int32[world]); stable order inside a segment."""
    ids = _i64(ids)
    send = np.empty((world, cap), np.int64)
    slot = np.empty((ids.shape[0],), np.int32)
    counts = np.empty((world,), np.int32)
    lib().oov_bucket_by_owner(_p(ids), _c(ids.shape[0]), _c(n_rows), _c(per), _c(world), _c(cap), _p(send), _p(slot),
                              _p(counts))
    return send, slot, counts


def lsh_codes_embed(codes, slot, buckets, other=None):
    """-> (score or None, emb) from codes u8[M,H] that came back from the owners and the slot of every lookup."""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    slot = np.ascontiguousarray(slot, dtype=np.int32)
    buckets = _f32(buckets)
    (M, H), D, B = codes.shape, buckets.shape[1], slot.shape[0]
    out = np.empty((B, D), np.float32)
    score = None
    if other is not None:
        other = _f32(other)
        score = np.empty((B,), np.float32)
    lib().oov_lsh_codes_embed(_p(codes), _c(M), _p(slot), _c(B), _c(H), _p(buckets), _c(D), _p(other), _p(score), _p(out))
    return score, out


def gather_mean(idx, W, g=2):
    idx, W = _i64(idx).ravel(), _f32(W)
    M, (N, D) = idx.shape[0], W.shape
    out = np.empty(((M + g - 1) // g, D), np.float32)
    lib().oov_gather_mean(_p(idx), _c(M), _c(g), _p(W), _c(N), _c(D), _p(out))
    return out


def gather_rows(ids, W):
    ids, W = _i64(ids), _f32(W)
    out = np.empty((ids.shape[0], W.shape[1]), np.float32)
    lib().oov_gather_rows(_p(ids), _c(ids.shape[0]), _p(W), _c(W.shape[0]), _c(W.shape[1]), _p(out))
    return out


def splice_rows(ids, table, oov_rows):
    ids, table, oov_rows = _i64(ids), _f32(table), _f32(oov_rows)
    n_vocab, D = table.shape
    rank = np.cumsum(ids >= n_vocab) - (ids >= n_vocab)
    rank = _i64(rank)
    out = np.empty((ids.shape[0], D), np.float32)
    lib().oov_splice_rows(_p(ids), _p(rank), _c(ids.shape[0]), _p(table), _c(n_vocab), _p(oov_rows),
                          _c(oov_rows.shape[0]), _c(D), _p(out))
    return out


def token_fields_embed(tokens, offsets, table, n_users, n_items, oov_user_rows, oov_item_rows, sum_fields=False):
    tokens, offsets, table = _i64(tokens), _i64(offsets), _f32(table)
    B, nf = tokens.shape
    T, D = table.shape
    ou, oi = tokens[:, 0] >= n_users, tokens[:, 1] >= n_items
    ru, ri = _i64(np.cumsum(ou) - ou), _i64(np.cumsum(oi) - oi)
    oov_user_rows = _f32(oov_user_rows).reshape(-1, D)
    oov_item_rows = _f32(oov_item_rows).reshape(-1, D)
    out = np.empty((B, D) if sum_fields else (B, nf, D), np.float32)
    lib().oov_token_fields_embed(_p(tokens), _c(B), _c(nf), _p(offsets), _p(table), _c(T), _c(D), _c(n_users),
                                 _c(n_items), _p(oov_user_rows), _p(ru), _c(oov_user_rows.shape[0]),
                                 _p(oov_item_rows), _p(ri), _c(oov_item_rows.shape[0]),
                                 ctypes.c_int(1 if sum_fields else 0), _p(out))
    return out


def lsh_embed_backward(bits, grad_out):
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    g = _f32(grad_out)
    B, H = bits.shape
    D = g.shape[1]
    out = np.empty((H, D), np.float32)
    lib().oov_lsh_embed_backward(_p(bits), _p(g), _c(B), _c(H), _c(D), _p(out))
    return out


def col_mean(W):
    W = _f32(W)
    mean = np.empty((W.shape[1],), np.float32)
    lib().oov_col_mean(_p(W), _c(W.shape[0]), _c(W.shape[1]), _p(mean))
    return mean


def broadcast_rows(vec, B, D):
    vec = None if vec is None else _f32(vec)
    out = np.empty((B, D), np.float32)
    lib().oov_broadcast_rows(_p(vec), _c(B), _c(D), _p(out))
    return out


def rowdot(U, E):
    U, E = _f32(U), _f32(E)
    out = np.empty((U.shape[0],), np.float32)
    lib().oov_rowdot(_p(U), _p(E), _c(U.shape[0]), _c(U.shape[1]), _p(out))
    return out


def full_sort_scores(U, E):
    U, E = _f32(U), _f32(E)
    out = np.empty((U.shape[0], E.shape[0]), np.float32)
    lib().oov_full_sort_scores(_p(U), _c(U.shape[0]), _p(E), _c(E.shape[0]), _c(U.shape[1]), _p(out))
    return out


def linear_act(X, W, bias, act=0):
    X, W, bias = _f32(X), _f32(W), _f32(bias)
    Y = np.empty((X.shape[0], W.shape[0]), np.float32)
    lib().oov_linear_act(_p(X), _c(X.shape[0]), _c(X.shape[1]), _p(W), _p(bias), _c(W.shape[0]), ctypes.c_int(act), _p(Y))
    return Y


def score_topk(U, E, k, n_skip_low=0):
    U, E = _f32(U), _f32(E)
    vals = np.empty((U.shape[0], k), np.float32)
    idx = np.empty((U.shape[0], k), np.int64)
    lib().oov_score_topk(_p(U), _c(U.shape[0]), _p(E), _c(E.shape[0]), _c(U.shape[1]), _c(k), _c(n_skip_low),
                         _p(vals), _p(idx))
    return vals, idx


def segment_topk(scores, cols, seg_ptr, k, col_lo=0, col_hi=2 ** 62):
    scores = _f32(scores)
    cols = np.ascontiguousarray(cols, dtype=np.int64)
    seg_ptr = np.ascontiguousarray(seg_ptr, dtype=np.int64)
    S = seg_ptr.shape[0] - 1
    vals = np.empty((S, k), np.float32)
    idx = np.empty((S, k), np.int64)
    lib().oov_segment_topk(_p(scores), _p(cols), _p(seg_ptr), _c(S), _c(k), _c(col_lo), _c(col_hi), _p(vals), _p(idx))
    return vals, idx


def topk_hits(idx, pos_ptr, pos_cols):
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    pos_ptr = np.ascontiguousarray(pos_ptr, dtype=np.int64)
    pos_cols = np.ascontiguousarray(pos_cols, dtype=np.int64)
    S, k = idx.shape
    out = np.empty((S, k + 1), np.int32)
    lib().oov_topk_hits(_p(idx), _c(S), _c(k), _p(pos_ptr), _p(pos_cols), _p(out))
    return out


def topk_hits_range(idx, pos_ptr, pos_cols, col_lo, col_hi):
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    pos_ptr = np.ascontiguousarray(pos_ptr, dtype=np.int64)
    pos_cols = np.ascontiguousarray(pos_cols, dtype=np.int64)
    S, k = idx.shape
    out = np.empty((S, k + 1), np.int32)
    lib().oov_topk_hits_range(_p(idx), _c(S), _c(k), _p(pos_ptr), _p(pos_cols), _c(col_lo), _c(col_hi), _p(out))
    return out


def eval_rows_build(pos_ptr, user_ids, pos_items, neg_items, n_neg):
    """-> (row_user, row_item, seg_ptr, pos_user)"""
    pos_ptr = np.ascontiguousarray(pos_ptr, dtype=np.int64)
    user_ids = np.ascontiguousarray(user_ids, dtype=np.int64)
    pos_items = np.ascontiguousarray(pos_items, dtype=np.int64)
    neg_items = np.ascontiguousarray(neg_items, dtype=np.int64)
    U, P = len(user_ids), len(pos_items)
    M = P * (1 + n_neg)
    row_user, row_item = np.empty(M, np.int64), np.empty(M, np.int64)
    seg_ptr, pos_user = np.empty(U + 1, np.int64), np.empty(P, np.int64)
    lib().oov_eval_rows_build(_p(pos_ptr), _c(U), _p(user_ids), _p(pos_items), _p(neg_items), _c(n_neg), _p(row_user),
                              _p(row_item), _p(seg_ptr), _p(pos_user))
    return row_user, row_item, seg_ptr, pos_user


def segment_dedup(cols, seg_ptr):
    cols = np.ascontiguousarray(cols, dtype=np.int64)
    seg_ptr = np.ascontiguousarray(seg_ptr, dtype=np.int64)
    out = np.empty_like(cols)
    lib().oov_segment_dedup(_p(cols), _p(seg_ptr), _c(len(seg_ptr) - 1), _p(out))
    return out


def topk_metric_sums(rec, disc, idcg_base, uids=None, n_old_users=0):
    """-> (sums f64[n_sides, 6, K], counts i64[n_sides, 6]); n_sides = 3 when uids is given (all / old / new users)"""
    rec = np.ascontiguousarray(rec, dtype=np.int32)
    disc = np.ascontiguousarray(disc, dtype=np.float64)
    idcg_base = np.ascontiguousarray(idcg_base, dtype=np.float64)
    U, K = rec.shape[0], rec.shape[1] - 1
    n_sides = 1 if uids is None else 3
    uids = None if uids is None else np.ascontiguousarray(uids, dtype=np.int64)
    sums, counts = np.empty((n_sides, 6, K), np.float64), np.empty((n_sides, 6), np.int64)
    lib().oov_topk_metric_sums(_p(rec), _c(U), _c(K), _p(disc), _p(idcg_base), _p(uids), _c(n_old_users), ctypes.c_int(n_sides),
                               _p(sums), _p(counts))
    return sums, counts


def score_topk_excl(U, E, k, excl_ptr, excl_cols, n_skip_low=0):
    U, E = _f32(U), _f32(E)
    excl_ptr = np.ascontiguousarray(excl_ptr, dtype=np.int64)
    excl_cols = np.ascontiguousarray(excl_cols, dtype=np.int64)
    vals = np.empty((U.shape[0], k), np.float32)
    idx = np.empty((U.shape[0], k), np.int64)
    lib().oov_score_topk_excl(_p(U), _c(U.shape[0]), _p(E), _c(E.shape[0]), _c(U.shape[1]), _c(k), _c(n_skip_low),
                              _p(excl_ptr), _p(excl_cols), _p(vals), _p(idx))
    return vals, idx

