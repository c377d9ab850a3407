#!/usr/bin/env python3
"""Per-kernel timing on the BASELINE-sized workloads (developer tool, GPU box only).

Launches on one stream, fresh ids per launch, replayed as one HIP graph with HIP events at both ends (run
under `rocprofv3 --kernel-trace --stats` for per-kernel durations: tools/profile_all.sh).  Prints one JSON
line per case with the algorithmic bytes / flops of SURVEY.md section 8(d)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mi_oov  # noqa: E402,F401
from mi_oov import ops  # noqa: E402


def timeit(fn, n_iter, warm=5):
    """us per launch: the n_iter launches are captured into one HIP graph and replayed (no host launch cost,
    reproducible to ~0.01 us); ops that cannot be captured fall back to back-to-back eager launches."""
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    try:
        gr = torch.cuda.CUDAGraph()
        keep = []
        with torch.cuda.graph(gr):
            for i in range(n_iter):
                keep.append(fn(warm + i)) if i < 4 else fn(warm + i)
        gr.replay()
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            a.record()
            gr.replay()
            b.record()
            torch.cuda.synchronize()
            t = a.elapsed_time(b) * 1e3 / n_iter
            best = t if best is None else min(best, t)
        return best
    except Exception:  # noqa: BLE001 -- capture refused (a sync inside the op): time it eagerly
        torch.cuda.synchronize()
    a.record()
    for i in range(n_iter):
        fn(warm + i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n_iter


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=10_000_000)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    N, B, F, D, H = args.items, args.batch, 64, 64, 8
    g = torch.Generator(device=dev).manual_seed(0)
    feat = torch.nn.functional.normalize(torch.randn((N, F), generator=g, device=dev), dim=-1)
    planes = torch.randn((H, F), generator=g, device=dev)
    buckets = torch.randn((H, D), generator=g, device=dev)
    total = max(args.iters + 5, 55)
    ids = torch.randint(0, N, (total, B), generator=g, device=dev)
    users = torch.randn((8, B, D), generator=g, device=dev)
    emb = torch.randn((B, D), generator=g, device=dev)
    keys = torch.randint(0, 256, (1024, 16), generator=g, device=dev, dtype=torch.uint8)
    planes24 = torch.randn((24, F), generator=g, device=dev)
    idx2 = torch.randint(0, N, (total, B, 2), generator=g, device=dev)
    Bs, Ns = 4096, 50_000
    U = torch.randn((Bs, D), generator=g, device=dev)
    E = torch.randn((Ns, D), generator=g, device=dev)
    planes16 = torch.randn((16, F), generator=g, device=dev)
    buckets16 = torch.randn((16, D), generator=g, device=dev)
    hashes = torch.rand((B, 1024), generator=g, device=dev)
    Ws = [torch.randn((o, i_), generator=g, device=dev) / i_ ** 0.5 for i_, o in ((1024, 512), (512, 512), (512, 512), (512, D))]
    bs = [torch.zeros((w.shape[0],), device=dev) for w in Ws]

    def mlp(x):
        for j, (w, b_) in enumerate(zip(Ws, bs)):
            x = ops.linear_act(x, w, b_, "gelu" if j < 3 else "sigmoid")
        return x

    x3w = [ops.LinearX3Weights(w) for w in Ws]

    def mlp_x3(x):
        for j, (w, b_) in enumerate(zip(Ws, bs)):
            x = ops.linear_act_x3(x, w, b_, "gelu" if j < 3 else "sigmoid", x3w[j])
        return x

    bits8 = (torch.rand((B, H), generator=g, device=dev) < 0.5).to(torch.uint8)
    bits8[:, 0] = 1
    idx9 = torch.randint(0, 9, (8, B), generator=g, device=dev)
    gtab = torch.zeros((N, D), device=dev)
    n_seg, per_seg = 4096, 1506  # 6 positives x (1 + 250) candidates per user
    seg_scores = torch.randn((n_seg * per_seg,), generator=g, device=dev)
    seg_cols = torch.randint(1, N, (n_seg * per_seg,), generator=g, device=dev)
    seg_ptr = torch.arange(0, n_seg * per_seg + 1, per_seg, device=dev)
    # name -> (callable, units per launch, bytes per unit, flops per unit, unit)
    cases = {
        "lsh_embed H=8 (hot kernel, rows stored)": (lambda i: ops.lsh_embed(ids[i], feat, planes, buckets), B, 8 + 4 * F + 4 * D, 0),
        "lsh_embed_score H=8 (hot kernel, fused)": (lambda i: ops.lsh_embed_score(ids[i], feat, planes, buckets, users[i % 8]), B, 16 + 4 * F + 4 * D + 4, 0),
        "lsh_embed H=16 (generic LDS kernel)": (lambda i: ops.lsh_embed(ids[i], feat, planes16, buckets16), B, 8 + 4 * F + 4 * D, 0),
        "lsh_lookup H=8 (in-vocab splice, 50% OOV)": (lambda i: ops.lsh_lookup(ids[i], feat[:N // 2], feat, planes, buckets), B, 8 + 4 * F + 4 * D, 0),
        "lsh_bits H=8": (lambda i: ops.lsh_bits(ids[i], feat, planes), B, 8 + 4 * F + H, 0),
        "slsh_embed nb=8": (lambda i: ops.slsh_embed(ids[i], feat, planes[:3], buckets), B, 8 + 4 * F + 8 * D, 0),
        "slsh_embed nb=N (bucket row from HBM)": (lambda i: ops.slsh_embed(ids[i], feat, planes24, feat), B, 8 + 4 * F + 8 * D, 0),
        "gather_rows": (lambda i: ops.gather_rows(ids[i], feat), B, 8 + 8 * D, 0),
        "gather_mean k=2 (knn aggregate)": (lambda i: ops.gather_mean(idx2[i], feat, 2), B, 16 + 8 * D + 4 * D, 0),
        "rowdot (BPR.predict)": (lambda i: ops.rowdot(users[i % 8], emb), B, 8 * D + 4, 0),
        "mapper_map 3round": (lambda i: ops.mapper_map(ids[i], "3round", N // 2, 1000), B, 16, 0),
        "siphash24_mod K=1024 (dhe)": (lambda i: ops.siphash24_mod(ids[i], keys), B, 8 + 4 * 1024, 0),
        "col_mean N=10M (mean embedder, one-off)": (lambda i: ops.col_mean(feat), N, 4 * D, 0),
        "broadcast_rows": (lambda i: ops.broadcast_rows(emb[0], B), B, 4 * D, 0),
        "full_sort_scores B=4096 N=50000": (lambda i: ops.full_sort_scores(U, E), Bs * Ns, 4, 2 * D),
        "score_topk k=20 B=4096 N=50000": (lambda i: ops.score_topk(U, E, 20, 1), Bs * Ns, 0, 2 * D),
        "dhe MLP 1024-512-512-512-64 (4 x linear_act)": (lambda i: mlp(hashes), B, 0, 2 * (1024 * 512 + 2 * 512 * 512 + 512 * 64)),
        "dhe MLP 1024-512-512-512-64 (4 x linear_x3, split bf16)": (lambda i: mlp_x3(hashes), B, 0, 2 * (1024 * 512 + 2 * 512 * 512 + 512 * 64)),
        "linear_x3 65536 x 1024 -> 512 (gelu)": (lambda i: ops.linear_act_x3(hashes, Ws[0], bs[0], "gelu", x3w[0]), B, 0, 2 * 1024 * 512),
        "linear_act 65536 x 1024 -> 512 (gelu, f32 MFMA)": (lambda i: ops.linear_act(hashes, Ws[0], bs[0], "gelu"), B, 0, 2 * 1024 * 512),
        "lsh_embed_backward H=8 (bucket-table grad)": (lambda i: ops.lsh_embed_backward(bits8, users[i % 8]), B, H + 4 * D, 0),
        "slsh_embed_backward nb=9": (lambda i: ops.slsh_embed_backward(idx9[i % 8], users[i % 8], 9), B, 8 + 4 * D, 0),
        "scatter_add_rows into 10M x 64 (gather backward)": (lambda i: ops.scatter_add_rows(ids[i], users[i % 8], N, out=gtab), B, 8 + 12 * D, 0),
        "segment_topk 4096 users x 1506 candidates k=20": (lambda i: ops.segment_topk(seg_scores, seg_cols, seg_ptr, 20), seg_scores.numel(), 4, 0),
        "segment_topk 4096 users x 1506 candidates k=20, columns [1, 5M) only": (lambda i: ops.segment_topk(seg_scores, seg_cols, seg_ptr, 20, 1, N // 2), seg_scores.numel(), 12, 0),
    }
    # round 2: the persistent launch and the two ends of the sharded exchange (1 M lookups = 16 batches per call)
    BIG = 16 * B
    big_ids = ids[:16].reshape(-1)
    q50 = ops.LshBatchQueue([ids[i] for i in range(50)], [users[i % 8] for i in range(50)])  # (838 MB of rows per launch: no reuse)
    multi = ops.LshMultiScorer(feat, planes, buckets)
    codes_big = (torch.rand((BIG, H), generator=g, device=dev) < 0.5).to(torch.uint8)
    slot_big = torch.randperm(BIG, generator=g, device=dev).to(torch.int32)
    users_big = torch.randn((BIG, D), generator=g, device=dev)
    sc_big = torch.empty((BIG,), device=dev)
    over = torch.zeros((1,), dtype=torch.int32, device=dev)
    cases.update({
        "lsh_embed_score_multi 50 batches per launch (persistent kernel)": (lambda i: multi.run(q50), 50 * B, 16 + 4 * F + 4 * D + 4, 0),
        "lsh_bits 1M lookups (persistent codes kernel, sharded owner side)": (lambda i: ops.lsh_bits(big_ids, feat, planes), BIG, 8 + 4 * F + H, 0),
        "lsh_codes_embed 1M lookups score only (sharded requester side)": (lambda i: ops.lsh_codes_embed(codes_big, slot_big, buckets, users_big, want_emb=False, score_out=sc_big), BIG, 4 + H + 4 * D + 4, 0),
        "bucket_by_owner 1M lookups world=8": (lambda i: ops.bucket_by_owner(big_ids, N, -(-N // 8), 8, BIG // 8 + 8192, over), BIG, 8 + 8 + 4, 0),
        "bucket_by_owner 65536 lookups world=8": (lambda i: ops.bucket_by_owner(ids[i], N, -(-N // 8), 8, B // 8 + 2048, over), B, 8 + 8 + 4, 0),
    })
    # round 3: K = 20 queued batches per launch of the rows / lookup modes of the persistent kernel and of the gather family.
    # Consecutive launches alternate between TWO queues of 20 distinct id batches (and two sets of preallocated outputs: a
    # serving loop rotates over its buffers), so that a launch finds none of its 335 MB of table rows in the 256 MiB
    # Infinity Cache; pointer tables are built once (ops caches them by address).
    KQ = 20
    idsq = [[ids[j * KQ + i] for i in range(KQ)] for j in range(2)]
    usersq = [users[i % 8] for i in range(KQ)]
    vt = feat[:N // 2]
    q_rows = [ops.LshBatchQueue(idsq[j], rows=True) for j in range(2)]
    q_sc = [ops.LshBatchQueue(idsq[j], usersq) for j in range(2)]
    m_rows, m_lrows = ops.LshMultiScorer(feat, planes, buckets), ops.LshMultiScorer(feat, planes, buckets, vtable=vt)
    m_lall = ops.LshMultiScorer(feat, planes, buckets, vtable=feat)  # every id in the vocabulary: a gather through the lsh kernel
    idx2q = [[idx2[j * KQ + i] for i in range(KQ)] for j in range(2)]
    big_bk = torch.randn((1000, 128), generator=g, device=dev)
    planes10 = planes24[:10].contiguous()
    o64 = [[t for t in torch.empty((KQ, B, 64), device=dev)] for j in range(2)]
    o128 = [[t for t in torch.empty((KQ, B, 128), device=dev)] for j in range(2)]
    cases.update({
        f"lsh_embed_multi {KQ} batches per launch (persistent kernel, rows stored, prepared table)": (lambda i: m_rows.run(q_rows[i % 2]), KQ * B, 8 + 4 * F + 4 * D, 0),
        f"lsh_lookup_multi rows {KQ} batches per launch (persistent kernel, 50% OOV)": (lambda i: m_lrows.run(q_rows[i % 2]), KQ * B, 8 + 4 * F + 4 * D, 0),
        f"lsh_lookup_multi rows {KQ} batches per launch (persistent kernel, 0% OOV = a plain gather)": (lambda i: m_lall.run(q_rows[i % 2]), KQ * B, 8 + 4 * F + 4 * D, 0),
        f"lsh_lookup_multi score {KQ} batches per launch (persistent kernel, 50% OOV)": (lambda i: m_lrows.run(q_sc[i % 2]), KQ * B, 16 + 4 * F + 4 * D + 4, 0),
        f"lsh_embed_score_multi {KQ} batches per launch (persistent kernel, prepared table)": (lambda i: m_rows.run(q_sc[i % 2]), KQ * B, 16 + 4 * F + 4 * D + 4, 0),
        f"gather_rows_multi {KQ} batches per launch": (lambda i: ops.gather_rows_multi(idsq[i % 2], feat, out=o64[i % 2]), KQ * B, 8 + 8 * D, 0),
        f"gather_mean_multi k=2 {KQ} batches per launch": (lambda i: ops.gather_mean_multi(idx2q[i % 2], feat, 2, out=o64[i % 2]), KQ * B, 16 + 8 * D + 4 * D, 0),
        f"slsh_embed_multi nb=N {KQ} batches per launch (bucket rows from LDS)": (lambda i: ops.slsh_embed_multi(idsq[i % 2], feat, planes24, feat, out=o64[i % 2]), KQ * B, 8 + 4 * F + 8 * D, 0),
        f"slsh_embed_multi nb=1000 D=128 10 planes {KQ} batches per launch": (lambda i: ops.slsh_embed_multi(idsq[i % 2], feat, planes10, big_bk, out=o128[i % 2]), KQ * B, 8 + 4 * F + 8 * 128, 0),
        "slsh_embed nb=1000 D=128 10 planes": (lambda i: ops.slsh_embed(ids[i], feat, planes10, big_bk), B, 8 + 4 * F + 8 * 128, 0),
    })
    U128 = torch.randn((Bs, 128), generator=g, device=dev)
    E128 = torch.randn((Ns, 128), generator=g, device=dev)
    cases["score_topk k=20 B=4096 N=50000 D=128 (two k-halves on the bf16 path)"] = (lambda i: ops.score_topk(U128, E128, 20, 1), Bs * Ns, 0, 2 * 128)
    if args.only.startswith("score_topk_excl"):  # full-sort evaluation: histories masked (bitmap in the kernel vs top-(k + h_max))
        gh = torch.Generator(device=dev).manual_seed(5)
        for hmean, hmax in ((60, 120), (100, 230), (100, 1500)):
            lens = torch.randint(0, 2 * hmean, (Bs,), generator=gh, device=dev)
            lens[0] = hmax
            ptr = torch.cat((torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(lens, 0)))
            colsx = torch.randint(1, Ns, (int(ptr[-1]),), generator=gh, device=dev)
            for masked in (True, False):
                def run(i, ptr=ptr, colsx=colsx, masked=masked, hmax=hmax):
                    ops._USE_MASKED_TOPK = masked
                    return ops.score_topk_excl(U, E, 20, ptr, colsx, 1, h_max=hmax)
                cases[f"score_topk_excl k=20 B=4096 N=50000 hist~{hmean} max {hmax} {'masked' if masked else 'k+h_max'}"] = (run, Bs * Ns, 0, 2 * D)
    if args.only.startswith("score_topk prepared"):  # catalogue converted once (knn search table, evaluation item table)
        catE = ops.TopkCatalogue(E)
        cases["score_topk prepared k=20 B=4096 N=50000"] = (lambda i: ops.score_topk(U, catE, 20, 1), Bs * Ns, 0, 2 * D)
        Ebig = torch.randn(10_000_000, D, device=dev)
        catB = ops.TopkCatalogue(Ebig)
        cases["score_topk prepared k=2 B=4096 N=10000000"] = (lambda i: ops.score_topk(U, catB, 2, 1), Bs * 10_000_000, 0, 2 * D)
        cases["score_topk prepared: the same without preparing, k=2 B=4096 N=10000000"] = (lambda i: ops.score_topk(U, Ebig, 2, 1), Bs * 10_000_000, 0, 2 * D)
    if args.only.startswith("score_topk degenerate"):  # every row ties everywhere: all rows take the exact fallback
        Uz = torch.zeros(512, D, device=dev)
        cases["score_topk degenerate (512 zero user rows, N=50000, k=20: exact fallback for every row)"] = (lambda i: ops.score_topk(Uz, E, 20, 1), 512 * Ns, 0, 2 * D)
    if args.only.startswith("score_topk sweep"):  # no cliffs over k, user-batch and catalogue sizes
        for (b_, n_, k_) in ((4096, 50000, 1), (4096, 50000, 5), (4096, 50000, 50), (4096, 50000, 120), (4096, 50000, 256),
                             (512, 50000, 20), (65536, 50000, 20), (4096, 500000, 20), (4096, 10000, 20),
                             (4096, 500000, 5), (4096, 10_000_000, 2)):  # the last one: the knn search of BASELINE's 10 M-row table
            U_ = torch.randn(b_, D, device=dev)
            E_ = torch.randn(n_, D, device=dev)
            cases[f"score_topk sweep k={k_} B={b_} N={n_}"] = (lambda i, U_=U_, E_=E_, k_=k_: ops.score_topk(U_, E_, k_, 1), b_ * n_, 0, 2 * D)
    # Bytes a case really MOVES where that differs from SURVEY 8(d)'s algorithmic count: the slsh kernels
    # serve the bucket row from LDS (the reference's arithmetic reaches H + 1 rows only), so 8 + 4F + 4D cross HBM, not
    # 8 + 4F + 8D; the fused score entries are handed user ROWS, not user ids (524 of the 532 bytes).  A roofline fraction
    # is printed on the moved bytes, and never above 1: a figure beyond the peak is an accounting error, not evidence.
    moved = {}
    for name in cases:
        if name.startswith("slsh_embed"):
            d = 128 if "D=128" in name else D
            moved[name] = 8 + 4 * F + 4 * d
        elif "score" in name and name.startswith(("lsh_embed_score", "lsh_lookup_multi score")):
            moved[name] = 8 + 4 * F + 4 * D + 4
    HBM_PEAK = 8000.0
    with torch.no_grad():
        for name, (fn, units, bpu, fpu) in cases.items():
            if args.only and not any(o in name for o in args.only.split(",")):
                continue
            us = timeit(fn, args.iters if units < 10 ** 8 else 5)
            bmoved = moved.get(name, bpu)
            gbs_moved = units * bmoved / us / 1e3
            line = {"case": name, "us_per_launch": round(us, 2), "units_per_launch": units,
                    "GB_per_s_survey": round(units * bpu / us / 1e3, 1), "GB_per_s_moved": round(gbs_moved, 1),
                    "bytes_per_unit_survey": bpu, "bytes_per_unit_moved": bmoved,
                    "TFLOP_per_s": round(units * fpu / us / 1e6, 2), "M_units_per_s": round(units / us, 1)}
            if bpu:
                if gbs_moved > HBM_PEAK:  # refused: no fraction is printed for it
                    line["frac_of_hbm_peak_moved"] = None
                    line["accounting_error"] = (f"{gbs_moved:.0f} GB/s on MOVED bytes exceeds the {HBM_PEAK:.0f} GB/s peak: the byte "
                                                "count of this case is wrong, or its inputs were served by the Infinity Cache")
                else:
                    line["frac_of_hbm_peak_moved"] = round(gbs_moved / HBM_PEAK, 3)
            print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
