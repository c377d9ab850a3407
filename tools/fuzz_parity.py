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
    feat = rng.standard_normal((N, 64), dtype=np.float32)
    feat[0] = 0
    planes, W = rng.standard_normal((H, 64), dtype=np.float32), rng.standard_normal((H, 64), dtype=np.float32)
    ids = np.stack([ids_of(rng, B, N) for _ in range(K)])
    users = rng.standard_normal((K, B, 64), dtype=np.float32)
    vt = rng.standard_normal((max(1, N // 2), 64), dtype=np.float32)
    idx2 = rng.integers(0, N, size=(K, 2 * B - int(rng.integers(0, 2))), dtype=np.int64)
    planes24, big = rng.standard_normal((int(rng.integers(1, 33)), 64), dtype=np.float32), rng.standard_normal((int(rng.choice([5, 40, 1000])), int(rng.choice([64, 128]))), dtype=np.float32)
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
    return f"multi K={K} B={B} N={N} H={H} slsh_planes={planes24.shape[0]} slsh_D={big.shape[1]} table={'prepared' if tab is not None else 'built'}", ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--only", default="", help="lsh / slsh / gather / topk / hash / multi")
    args = ap.parse_args()
    makers = [case_lsh, case_lsh, case_slsh, case_gather, case_topk, case_hash, case_multi]
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
