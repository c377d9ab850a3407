#!/usr/bin/env python3
"""Micro-timing of the hot kernels on the bench workload (developer tool, GPU box only).
Back-to-back launches on one stream, fresh ids per launch, HIP events at both ends."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mi_oov  # noqa: E402,F401
from mi_oov import ops  # noqa: E402


def timeit(fn, n_iter, warm=10):
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
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    N, B, F, D, H = args.items, args.batch, 64, 64, 8
    g = torch.Generator(device=dev).manual_seed(0)
    feat = torch.nn.functional.normalize(torch.randn((N, F), generator=g, device=dev), dim=-1)
    planes = torch.randn((H, F), generator=g, device=dev)
    buckets = torch.randn((H, D), generator=g, device=dev)
    total = args.iters + 10
    ids = torch.randint(0, N, (total, B), generator=g, device=dev)
    users = torch.randn((8, B, D), generator=g, device=dev)
    emb = torch.randn((B, D), generator=g, device=dev)
    cases = {
        "gather_rows(copy floor)": (lambda i: ops.gather_rows(ids[i], feat), 8 + 4 * F + 4 * D),
        "lsh_embed": (lambda i: ops.lsh_embed(ids[i], feat, planes, buckets), 8 + 4 * F + 4 * D),
        "lsh_embed_score": (lambda i: ops.lsh_embed_score(ids[i], feat, planes, buckets, users[i % 8]), 16 + 4 * F + 4 * D + 4),
        "lsh_bits": (lambda i: ops.lsh_bits(ids[i], feat, planes), 8 + 4 * F + H),
        "rowdot": (lambda i: ops.rowdot(users[i % 8], emb), 8 * D + 4),
        "slsh_embed": (lambda i: ops.slsh_embed(ids[i], feat, planes[:3], buckets), 8 + 4 * F + 8 * D),
    }
    with torch.no_grad():
        for name, (fn, bpl) in cases.items():
            if args.only and args.only not in name:
                continue
            us = timeit(fn, args.iters)
            print(f"{name:28s} {us:8.2f} us/launch  {B * bpl / us / 1e3:8.1f} GB/s  ({B / us:7.1f} M lookups/s)", flush=True)


if __name__ == "__main__":
    main()
