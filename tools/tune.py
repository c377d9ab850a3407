#!/usr/bin/env python3
"""Per-kernel timing on the BASELINE-sized workloads (developer tool, GPU box only).

Back-to-back launches on one stream, fresh ids per launch, HIP events at both ends (includes the
Python launch overhead when the kernel is shorter than ~8 us; run under `rocprofv3 --kernel-trace
--stats` for kernel-only durations).  Prints one JSON line per case with the algorithmic bytes /
flops of SURVEY.md section 8(d)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mi_oov  # noqa: E402,F401
from mi_oov import ops  # noqa: E402


def timeit(fn, n_iter, warm=5):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
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
    total = args.iters + 5
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
    }
    with torch.no_grad():
        for name, (fn, units, bpu, fpu) in cases.items():
            if args.only and args.only not in name:
                continue
            us = timeit(fn, args.iters if units < 10 ** 8 else 5)
            print(json.dumps({"case": name, "us_per_launch": round(us, 2), "units_per_launch": units,
                              "GB_per_s": round(units * bpu / us / 1e3, 1), "TFLOP_per_s": round(units * fpu / us / 1e6, 2),
                              "M_units_per_s": round(units / us, 1)}), flush=True)


if __name__ == "__main__":
    main()
