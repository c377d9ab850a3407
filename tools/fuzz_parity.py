#!/usr/bin/env python3
"""Random shapes for the path's kernels against the oracle (bit for bit), to catch a shape rule the fixed parity cases miss
(the plane chunks, column windows, scalar tails, partial tiles).  Developer tool, GPU box:
    python3 tools/fuzz_parity.py [--cases 300] [--seed 0] [--only multi]
Prints one line per failing case with everything needed to repeat it, and a summary."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402

import mi_oov  # noqa: E402,F401
import oov_oracle as oracle  # noqa: E402
from mi_oov import ops  # noqa: E402

dev = torch.device("cuda:0")
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    if a.dtype.kind == "f":
        return bool(np.array_equal(a.view(np.uint32), b.view(np.uint32)) or
                    (np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a), np.nan_to_num(b))))
    return bool(np.array_equal(a, b))


def pick(rng, small, big, p_big=0.25):
    lo, hi = big if rng.random() < p_big else small
    return int(rng.integers(lo, hi + 1))


def ids_of(rng, B, N):
    ids = rng.integers(0, N, size=B, dtype=np.int64)
    if B > 3:
        ids[rng.integers(0, B)] = N + int(rng.integers(0, 5))
        ids[rng.integers(0, B)] = -int(rng.integers(1, 5))
    return ids


def case_lsh(rng):
    B, N = pick(rng, (1, 300), (301, 5000)), pick(rng, (1, 200), (201, 3000))
    F = int(rng.choice([1, 3, 4, 20, 21, 63, 64, 65, 128, 130, 256, 257, 300, 768, 1000]))
    H = pick(rng, (1, 40), (41, 1200), 0.3)
    D = int(rng.choice([1, 2, 4, 7, 32, 50, 64, 65, 128, 130, 256, 257, 300, 512, 600]))
    if F * H > 600_000:
        H = max(1, 600_000 // F)
    feat, planes, W = (rng.standard_normal(s, dtype=np.float32) for s in ((N, F), (H, F), (H, D)))
    other, table = rng.standard_normal((B, D), dtype=np.float32), rng.standard_normal((max(1, N // 2), D), dtype=np.float32)
    feat[0] = 0
    ids = ids_of(rng, B, N)
    desc = f"lsh B={B} N={N} F={F} H={H} D={D}"
    o_emb, o_bits = oracle.lsh_embed(ids, feat, planes, W, want_bits=True)
    o_score, _ = oracle.lsh_embed_score(ids, feat, planes, W, other)
    o_look = oracle.lsh_lookup(ids, table, feat, planes, W)
    emb_b, bits_b = ops._lsh_forward(d(ids), d(feat), d(planes), d(W), want_bits=True)
    ok = {"rows": same(ops.lsh_embed(d(ids), d(feat), d(planes), d(W)).cpu().numpy(), o_emb),
          "bits": same(ops.lsh_bits(d(ids), d(feat), d(planes)).cpu().numpy(), o_bits),
          "rows+bits": same(emb_b.cpu().numpy(), o_emb) and same(bits_b.cpu().numpy(), o_bits),
          "score": same(ops.lsh_embed_score(d(ids), d(feat), d(planes), d(W), d(other)).cpu().numpy(), o_score),
          "lookup": same(ops.lsh_lookup(d(ids), d(table), d(feat), d(planes), d(W)).cpu().numpy(), o_look),
          "lookup_score": same(ops.lsh_lookup_score(d(ids), d(table), d(feat), d(planes), d(W), d(other)).cpu().numpy(), oracle.rowdot(other, o_look)),
          "backward": same(ops.lsh_embed_backward(d(o_bits), d(other)).cpu().numpy(), oracle.lsh_embed_backward(o_bits, other))}
    return desc, ok


def case_slsh(rng):
    B, N = pick(rng, (1, 300), (301, 5000)), pick(rng, (1, 200), (201, 3000))
    F = int(rng.choice([1, 3, 4, 21, 64, 65, 128, 300, 1000, 3000]))
    H = int(rng.integers(0, 41))
    nb = int(rng.choice([1, 2, 5, 8, 9, 64, 65, 1000, 5000]))
    D = int(rng.choice([1, 7, 50, 64, 128, 130, 300, 512]))
    feat, planes, W = (rng.standard_normal(s, dtype=np.float32) for s in ((N, F), (H, F), (nb, D)))
    ids = ids_of(rng, B, N)
    o_emb, o_idx = oracle.slsh_embed(ids, feat, planes, W)
    return f"slsh B={B} N={N} F={F} H={H} nb={nb} D={D}", {
        "rows": same(ops.slsh_embed(d(ids), d(feat), d(planes), d(W)).cpu().numpy(), o_emb),
        "idx": same(ops.slsh_index(d(ids), d(feat), d(planes), nb).cpu().numpy(), o_idx)}


def case_gather(rng):
    B, N = pick(rng, (1, 300), (301, 5000)), pick(rng, (1, 200), (201, 3000))
    D = int(rng.choice([1, 3, 4, 50, 64, 65, 128, 300, 1000]))
    g = int(rng.choice([1, 2, 3]))
    W = rng.standard_normal((N, D), dtype=np.float32)
    idx = rng.integers(0, N, size=B * g + int(rng.integers(0, g)), dtype=np.int64)
    ids = ids_of(rng, B, N)
    a, b = rng.standard_normal((B, D), dtype=np.float32), rng.standard_normal((B, D), dtype=np.float32)
    return f"gather B={B} N={N} D={D} g={g}", {
        "gather_mean": same(ops.gather_mean(d(idx), d(W), g).cpu().numpy(), oracle.gather_mean(idx, W, g)),
        "gather_rows": same(ops.gather_rows(d(ids), d(W)).cpu().numpy(), oracle.gather_rows(ids, W)),
        "rowdot": same(ops.rowdot(d(a), d(b)).cpu().numpy(), oracle.rowdot(a, b))}


def case_topk(rng):
    B, N = pick(rng, (1, 100), (101, 600)), pick(rng, (1, 3000), (3001, 60000))
    D = int(rng.choice([1, 5, 22, 64, 65, 128, 130, 300]))
    k = min(N, pick(rng, (1, 30), (31, 300)))
    skip = int(rng.integers(0, 3))
    U, E = rng.standard_normal((B, D), dtype=np.float32), rng.standard_normal((N, D), dtype=np.float32)
    if rng.random() < 0.3:
        E[rng.integers(0, N, size=max(1, N // 10))] = E[0]  # ties
    v, i = ops.score_topk(d(U), d(E), k, skip)
    ov, oi = oracle.score_topk(U, E, k, skip)
    return f"score_topk B={B} N={N} D={D} k={k} skip={skip}", {"idx": same(i.cpu().numpy(), oi), "vals": same(v.cpu().numpy(), ov)}


def case_hash(rng):
    B, K = pick(rng, (1, 500), (501, 5000)), int(rng.choice([1, 2, 7, 16, 64, 100]))
    ids = rng.integers(-2 ** 62, 2 ** 62, size=B, dtype=np.int64)
    keys = rng.integers(0, 256, size=(K, 16), dtype=np.uint8)
    kind = str(rng.choice(["fast", "3round", "mod", "64bit"]))
    n_orig, nb = int(rng.integers(1, 1000)), int(rng.integers(1, 5000))
    mids = rng.integers(0, 3 * n_orig + 5, size=B, dtype=np.int64)
    return f"hash B={B} K={K} kind={kind} n_orig={n_orig} nb={nb}", {
        "siphash": same(ops.siphash24_mod(d(ids), d(keys)).cpu().numpy(), oracle.siphash24_mod(ids, keys)),
        "mapper": same(ops.mapper_map(d(mids), kind, n_orig, nb).cpu().numpy(), oracle.mapper_map(mids, kind, n_orig, nb))}


def case_multi(rng):
    """K queued batches per launch (the persistent kernel's tile schedule: static part, ticket pool, odd tile counts, partial
    last tiles) against the oracle, batch by batch."""
    K = pick(rng, (1, 6), (7, 40))
    B = pick(rng, (1, 300), (301, 20000), 0.4)
    N = pick(rng, (1, 500), (501, 50000))
    H = int(rng.integers(1, 9))
    F = D = 64
    if rng.random() < 0.3:  # shapes the persistent kernel does not take: the wrappers' K single launches
        F, D, H = int(rng.choice([3, 20, 64, 65, 130])), int(rng.choice([1, 50, 64, 65, 128, 257])), int(rng.choice([1, 8, 9, 40, 70]))
        B = min(B, 3000)
    feat = rng.standard_normal((N, F), dtype=np.float32)
    feat[0] = 0
    planes, W = rng.standard_normal((H, F), dtype=np.float32), rng.standard_normal((H, D), dtype=np.float32)
    ids = np.stack([ids_of(rng, B, N) for _ in range(K)])
    users = rng.standard_normal((K, B, D), dtype=np.float32)
    vt = rng.standard_normal((max(1, N // 2), D), dtype=np.float32)
    idx2 = rng.integers(0, N, size=(K, 2 * B - int(rng.integers(0, 2))), dtype=np.int64)
    planes24, big = rng.standard_normal((int(rng.integers(1, 33)), F), dtype=np.float32), rng.standard_normal((int(rng.choice([5, 40, 1000])), int(rng.choice([64, 128]) if F == 64 else D)), dtype=np.float32)
    f, p, w, i, u, v = d(feat), d(planes), d(W), d(ids), d(users), d(vt)
    il, ul = [i[k] for k in range(K)], [u[k] for k in range(K)]
    tab = ops.LshTable(w) if rng.random() < 0.5 else None
    sc = ops.lsh_embed_score_multi(il, f, p, w, ul)
    rows = ops.lsh_embed_multi(il, f, p, w, table=tab)
    look = ops.lsh_lookup_multi(il, v, f, p, w, lsh_table=tab)
    looks = ops.lsh_lookup_multi(il, v, f, p, w, other_list=ul, lsh_table=tab)
    gr = ops.gather_rows_multi(il, f)
    gm = ops.gather_mean_multi([d(idx2[k]) for k in range(K)], f, 2)
    sl, sidx = ops.slsh_embed_multi(il, f, d(planes24), d(big), want_idx=True)
    ok = {n: True for n in ("score", "rows", "lookup", "lookup_score", "gather_rows", "gather_mean", "slsh", "slsh_idx")}
    for k in range(K):
        o_emb = oracle.lsh_embed(ids[k], feat, planes, W)
        o_look = oracle.lsh_lookup(ids[k], vt, feat, planes, W)
        o_sl, o_sidx = oracle.slsh_embed(ids[k], feat, planes24, big)
        ok["score"] &= same(sc[k].cpu().numpy(), oracle.lsh_embed_score(ids[k], feat, planes, W, users[k])[0])
        ok["rows"] &= same(rows[k].cpu().numpy(), o_emb)
        ok["lookup"] &= same(look[k].cpu().numpy(), o_look)
        ok["lookup_score"] &= same(looks[k].cpu().numpy(), oracle.rowdot(users[k], o_look))
        ok["gather_rows"] &= same(gr[k].cpu().numpy(), oracle.gather_rows(ids[k], feat))
        ok["gather_mean"] &= same(gm[k].cpu().numpy(), oracle.gather_mean(idx2[k], feat, 2))
        ok["slsh"] &= same(sl[k].cpu().numpy(), o_sl)
        ok["slsh_idx"] &= same(sidx[k].cpu().numpy(), o_sidx)
    return f"multi K={K} B={B} N={N} F={F} D={D} H={H} slsh_planes={planes24.shape[0]} slsh_D={big.shape[1]} table={'prepared' if tab is not None else 'built'}", ok


def case_eval(rng):
    """The sampled-ranking evaluation's kernels: rows of a group of batches, duplicate handling, per-user top-k over column
    ranges, hit blocks, the metric sums; and the full-sort route's top-k under a history mask."""
    U = pick(rng, (1, 60), (61, 3000))
    n_neg = int(rng.choice([0, 1, 3, 50, 250]))
    n_items = pick(rng, (2, 300), (301, 100000))
    npos = rng.integers(0 if U > 1 else 1, 5, size=U)
    if npos.sum() == 0:
        npos[0] = 1
    pos_ptr = np.concatenate([[0], np.cumsum(npos)]).astype(np.int64)
    P = int(pos_ptr[-1])
    user_ids = rng.permutation(10 * U)[:U].astype(np.int64)
    pos_items = rng.integers(0, n_items, size=P, dtype=np.int64)
    neg_items = rng.integers(0, n_items, size=(P, n_neg), dtype=np.int64).reshape(-1)
    o_ru, o_ri, o_seg, o_pu = oracle.eval_rows_build(pos_ptr, user_ids, pos_items, neg_items, n_neg)
    ru, ri, seg, pu = ops.eval_rows_build(d(pos_ptr), d(user_ids), d(pos_items), d(neg_items), n_neg, want_pos_user=True)
    ok = {"eval_rows": same(ru.cpu().numpy(), o_ru) and same(ri.cpu().numpy(), o_ri) and same(seg.cpu().numpy(), o_seg) and same(pu.cpu().numpy(), o_pu)}
    o_dd = oracle.segment_dedup(o_ri, o_seg)
    ok["dedup"] = same(ops.segment_dedup(d(o_ri), d(o_seg)).cpu().numpy(), o_dd)
    scores = rng.standard_normal(len(o_ri)).astype(np.float32)
    if rng.random() < 0.3:
        scores[rng.integers(0, len(scores), size=max(1, len(scores) // 7))] = scores[0]
    k = int(rng.choice([1, 5, 10, 20, 50, 256]))
    lo = int(rng.integers(0, n_items // 2 + 1))
    hi = int(rng.integers(lo + 1, n_items + 2))
    # (the whole range reads only the winners' columns and takes item ids >= 0: the raw columns; the de-duplicated ones,
    #  with their -1 entries, go with an explicit range -- the evaluator's "everything" is [0, 2^62 - 1))
    for name, cols_, (a, b) in (("whole", o_ri, (0, 2 ** 62)), ("all", o_dd, (0, 2 ** 62 - 1)), ("range", o_dd, (lo, hi))):
        ov, oi = oracle.segment_topk(scores, cols_, o_seg, k, a, b)
        v, i = ops.segment_topk(d(scores), d(cols_), d(o_seg), k, a, b)
        ok["segment_topk_" + name] = same(v.cpu().numpy(), ov) and same(i.cpu().numpy(), oi)
        ok["hits_" + name] = same(ops.topk_hits(d(oi), d(pos_ptr), d(pos_items), a, b).cpu().numpy(), oracle.topk_hits_range(oi, pos_ptr, pos_items, a, b))
    rec = oracle.topk_hits(oracle.segment_topk(scores, o_dd, o_seg, k)[1], pos_ptr, pos_items)
    disc = 1.0 / np.log2(np.arange(2, k + 2, dtype=np.float64))
    idcg = np.cumsum(disc)
    n_old = int(rng.integers(0, 10 * U + 1))
    for uids in (None, user_ids):
        os_, oc = oracle.topk_metric_sums(rec, disc, idcg, uids, n_old)
        s_, c_ = ops.topk_metric_sums(d(rec), d(disc), d(idcg), None if uids is None else d(uids), n_old)
        ok["metric_sums" + ("" if uids is None else "_sides")] = bool(np.array_equal(s_.cpu().numpy().view(np.uint64), os_.view(np.uint64))) and same(c_.cpu().numpy(), oc)
    # full-sort route: top-k with excluded (history) columns
    B, N, D = pick(rng, (1, 40), (41, 300)), pick(rng, (30, 3000), (3001, 40000)), int(rng.choice([8, 64, 100, 128, 200]))
    kk = min(N // 2, int(rng.choice([1, 10, 20, 100])))
    Uq, E = rng.standard_normal((B, D), dtype=np.float32), rng.standard_normal((N, D), dtype=np.float32)
    cnt = rng.integers(0, int(rng.choice([2, 30, 400])), size=B)
    cnt = np.minimum(cnt, N - kk - 1)
    eptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    ecols = np.concatenate([np.sort(rng.choice(N, size=int(c), replace=False)) for c in cnt] + [np.empty(0, np.int64)]).astype(np.int64)
    skip = int(rng.integers(0, 2))
    ov, oi = oracle.score_topk_excl(Uq, E, kk, eptr, ecols, skip)
    v, i = ops.score_topk_excl(d(Uq), d(E), kk, d(eptr), d(ecols), skip, h_max=int(cnt.max()) if B else 0)
    ok["topk_excl"] = same(i.cpu().numpy(), oi) and same(v.cpu().numpy(), ov)
    return f"eval U={U} n_neg={n_neg} items={n_items} k={k} range=[{lo},{hi}) | excl B={B} N={N} D={D} k={kk} hist<={int(cnt.max())}", ok


def case_misc(rng):
    """Exchange bucketing, codes -> rows, the row splice, column mean / broadcast, the materialised full sort, an f32 layer."""
    B, world = pick(rng, (1, 500), (501, 70000)), int(rng.choice([1, 2, 3, 8, 16, 20]))
    n_rows = pick(rng, (world, 5000), (5001, 10 ** 7))
    per = -(-n_rows // world)
    cap = max(1, int(B / world * rng.choice([0.5, 1.3, 4.0])) + int(rng.integers(0, 3)))
    ids = rng.integers(-2, n_rows + 2, size=B, dtype=np.int64)
    o_send, o_slot, o_cnt = oracle.bucket_by_owner(ids, n_rows, per, world, cap)
    send, slot, cnt = ops.bucket_by_owner(d(ids), n_rows, per, world, cap)[:3]
    fits = bool((o_cnt <= cap).all())  # beyond the capacity WHICH lookups are dropped is not defined: counts only
    ok = {"bucket_counts": same(cnt.cpu().numpy(), o_cnt)}
    if fits:
        # the order inside a segment is the oracle's choice (stable), not the kernel's (atomic reservations): what must hold is
        # that slot[b] is where lookup b's owner-local row sits, every slot taken once, invalid ids coded as the oracle codes them
        got, gs = send.cpu().numpy(), slot.cpu().numpy()
        valid = o_slot >= 0
        ok["bucket_slots"] = bool(np.array_equal(gs[~valid], o_slot[~valid]) and (gs[valid] >= 0).all() and
                                  len(np.unique(gs[valid])) == int(valid.sum()) and
                                  np.array_equal(got.reshape(-1)[gs[valid]], o_send.reshape(-1)[o_slot[valid]]) and
                                  np.array_equal(gs[valid] // cap, o_slot[valid] // cap))
        ok["bucket_send"] = all(same(np.sort(got[r, :o_cnt[r]]), np.sort(o_send[r, :o_cnt[r]])) and (got[r, o_cnt[r]:] == -1).all() for r in range(world))
    M, H, D = pick(rng, (1, 400), (401, 5000)), int(rng.integers(1, 41)), int(rng.choice([1, 7, 64, 65, 128, 256]))
    codes = (rng.random((M, H)) < 0.5).astype(np.uint8)
    codes[rng.integers(0, M)] = 0xFF
    sl = rng.integers(-2, M, size=B if B < 3000 else 3000).astype(np.int32)
    W, other = rng.standard_normal((H, D), dtype=np.float32), rng.standard_normal((len(sl), D), dtype=np.float32)
    o_sc, o_emb = oracle.lsh_codes_embed(codes, sl, W, other)
    sc, emb = ops.lsh_codes_embed(d(codes), d(sl), d(W), d(other), want_emb=True)
    ok["codes_embed"] = same(emb.cpu().numpy(), o_emb) and same(sc.cpu().numpy(), o_sc)
    n_vocab = int(rng.integers(1, 300))
    table = rng.standard_normal((n_vocab, D), dtype=np.float32)
    sid = rng.integers(0, 2 * n_vocab, size=min(B, 2000), dtype=np.int64)
    oov_rows = rng.standard_normal((int((sid >= n_vocab).sum()), D), dtype=np.float32)
    if len(oov_rows):
        ok["splice"] = same(ops.splice_rows(d(sid), d(table), d(oov_rows)).cpu().numpy(), oracle.splice_rows(sid, table, oov_rows))
    ok["col_mean"] = same(ops.col_mean(d(table)).cpu().numpy(), oracle.col_mean(table))
    Uq, E = rng.standard_normal((int(rng.integers(1, 70)), D), dtype=np.float32), rng.standard_normal((int(rng.integers(1, 700)), D), dtype=np.float32)
    ok["full_sort"] = same(ops.full_sort_scores(d(Uq), d(E)).cpu().numpy(), oracle.full_sort_scores(Uq, E))
    Kin, Nout = int(rng.choice([1, 5, 16, 40, 64, 130])), int(rng.choice([1, 7, 64, 96, 200]))
    X, Wl, bl = rng.standard_normal((Uq.shape[0] * 3, Kin), dtype=np.float32), rng.standard_normal((Nout, Kin), dtype=np.float32), rng.standard_normal(Nout).astype(np.float32)
    ok["linear_f32"] = same(ops.linear_act(d(X), d(Wl), d(bl), None).cpu().numpy(), oracle.linear_act(X, Wl, bl, 0))
    return f"misc B={B} world={world} n_rows={n_rows} cap={cap} | codes M={M} H={H} D={D} | linear {Kin}->{Nout}", ok


def case_plugin(rng):
    """The lsh / slsh plugin CLASSES on random feature widths, embedding sizes and bucket counts (zero-padded hot operands,
    plane chunks, column windows all at once) against the oracle on the reference-shaped operands; train mode strips the
    prime pad in place."""
    import mi_oov as mi
    n_new, n_orig = int(rng.integers(50, 800)), int(rng.integers(10, 50))
    widths = [int(rng.choice([0, 1, 3, 17, 40, 64, 70])) for _ in range(int(rng.integers(1, 4)))]
    D = int(rng.choice([1, 16, 50, 64, 64, 64, 128, 130, 300]))
    nb = int(rng.choice([1, 2, 8, 8, 9, 33, 100, 400]))
    g = torch.Generator().manual_seed(int(rng.integers(0, 2 ** 31)))

    def feats():
        cols = {"id": torch.arange(n_new)}
        for j, w in enumerate(widths):
            cols[f"f{j}"] = torch.randn((n_new,) if w == 0 else (n_new, w), generator=g)
        return mi.FeatureTable(cols)

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.user_oov_buckets = torch.nn.Embedding.from_pretrained(torch.randn((nb, D), generator=g).to(dev), freeze=True)
            self.item_oov_buckets = torch.nn.Embedding.from_pretrained(torch.randn((nb, D), generator=g).to(dev), freeze=True)

    prime = 112062759511
    norm = str(rng.choice(["per-feature", "global", "none"]))
    model = M()
    ids = torch.from_numpy(rng.integers(0, n_new, size=int(rng.integers(1, 600)), dtype=np.int64)).to(dev)
    ok = {}
    for name, emb in (("lsh", mi.LSHInductiveEmbedder(feats(), feats(), n_orig, n_orig, nb, nb, D, dev, prime, norm, mi.InductiveFeatureCache())),
                      ("slsh", mi.SingleLSHInductiveEmbedder(feats(), feats(), n_orig, n_orig, nb, nb, D, dev, prime, norm))):
        feat = emb.item_feature_mat.cpu().numpy()
        planes = emb.item_lsh.uniform_planes[0].data.cpu().numpy()
        W = model.item_oov_buckets.weight.cpu().numpy()
        emb.set_eval()
        rows = emb.embed_item_ids(ids.clone(), model).cpu().numpy()
        codes = emb._hash_items(ids).cpu().numpy()
        emb.set_train()
        padded = ids.clone() + prime * (torch.arange(ids.numel(), device=dev) % 2)
        rows_t = emb.embed_item_ids(padded, model).cpu().numpy()
        if name == "lsh":
            o_rows, o_codes = oracle.lsh_embed(ids.cpu().numpy(), feat, planes, W, want_bits=True)
            ok["lsh_codes"] = same(codes.astype(np.uint8), o_codes)
        else:
            o_rows, o_codes = oracle.slsh_embed(ids.cpu().numpy(), feat, planes, W)
            ok["slsh_idx"] = same(codes, o_codes)
        ok[name + "_rows"] = same(rows, o_rows)
        ok[name + "_train_rows"] = same(rows_t, o_rows) and bool(torch.equal(padded, ids))
    return f"plugin widths={widths} D={D} buckets={nb} norm={norm} n_new={n_new} B={ids.numel()}", ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--only", default="", help="lsh / slsh / gather / topk / hash / multi / eval / misc / plugin")
    args = ap.parse_args()
    makers = [case_lsh, case_lsh, case_slsh, case_gather, case_topk, case_hash, case_multi, case_eval, case_misc, case_plugin]
    if args.only:
        makers = [m for m in makers if m.__name__ == "case_" + args.only]
    bad = 0
    for c in range(args.cases):
        rng = np.random.default_rng([args.seed, c])
        mk = makers[c % len(makers)]
        try:
            with torch.no_grad():
                desc, ok = mk(rng)
            wrong = [k for k, v in ok.items() if not v]
            if wrong:
                bad += 1
                print(f"DIFF seed={args.seed} case={c} {desc}: {wrong}", flush=True)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print(f"FAIL seed={args.seed} case={c} {mk.__name__}: {type(e).__name__}: {str(e)[:200]}", flush=True)
        if (c + 1) % 50 == 0:
            print(f"... {c + 1} cases, {bad} bad", flush=True)
    print(f"{args.cases} cases, {bad} bad", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
