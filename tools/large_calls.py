#!/usr/bin/env python3
"""ONE large call per op (default 2 M lookups on the 10 M-row table: what "embed every OOV item" or a queued evaluation
hands over) -- us per call and the fraction of the HBM peak on the bytes the kernel moves.  Developer tool, GPU box:
    python3 tools/large_calls.py [--lookups 2097152] [--only substr]
Eager launches between two events, fresh ids per launch (5 id sets), best of 3 rounds of 4 launches."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mi_oov  # noqa: E402,F401
from mi_oov import ops  # noqa: E402


def timeit(fn, sets):
    for i in range(2):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = None
    for _ in range(3):
        a.record()
        for i in range(4):
            fn((i + 1) % sets)
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) * 1e3 / 4
        best = t if best is None else min(best, t)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=10_000_000)
    ap.add_argument("--lookups", type=int, default=1 << 21)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    N, B, F, D = args.items, args.lookups, 64, 64
    g = torch.Generator(device=dev).manual_seed(0)
    feat = torch.nn.functional.normalize(torch.randn((N, F), generator=g, device=dev), dim=-1)
    sets = 5
    ids = torch.randint(0, N, (sets, B), generator=g, device=dev)
    idx2 = torch.randint(0, N, (sets, B, 2), generator=g, device=dev)
    P = {h: torch.randn((h, F), generator=g, device=dev) for h in (8, 10, 16, 24, 32)}
    W = {h: torch.randn((h, D), generator=g, device=dev) for h in (8, 16, 32)}
    W128 = torch.randn((8, 128), generator=g, device=dev)
    big128 = torch.randn((1000, 128), generator=g, device=dev)
    users = torch.randn((B, D), generator=g, device=dev)
    out64 = torch.empty((B, D), device=dev)
    cases = {
        "lsh_embed H=8 (persistent kernel)": (lambda i: ops.lsh_embed(ids[i], feat, P[8], W[8]), 8 + 4 * F + 4 * D),
        "lsh_embed_score H=8 (persistent kernel)": (lambda i: ops.lsh_embed_score(ids[i], feat, P[8], W[8], users), 16 + 4 * F + 4 * D + 4),
        "lsh_lookup H=8, 50% OOV": (lambda i: ops.lsh_lookup(ids[i], feat[:N // 2], feat, P[8], W[8]), 8 + 4 * F + 4 * D),
        "lsh_bits H=8": (lambda i: ops.lsh_bits(ids[i], feat, P[8]), 8 + 4 * F + 8),
        "lsh_embed H=16 (lsh64g)": (lambda i: ops.lsh_embed(ids[i], feat, P[16], W[16]), 8 + 4 * F + 4 * D),
        "lsh_embed H=32 (lsh64g)": (lambda i: ops.lsh_embed(ids[i], feat, P[32], W[32]), 8 + 4 * F + 4 * D),
        "lsh_embed H=8 D=128 (generic)": (lambda i: ops.lsh_embed(ids[i], feat, P[8], W128), 8 + 4 * F + 4 * 128),
        "slsh_embed 24 planes nb=N D=64": (lambda i: ops.slsh_embed(ids[i], feat, P[24], feat), 8 + 4 * F + 4 * D),
        "slsh_embed 10 planes nb=1000 D=128": (lambda i: ops.slsh_embed(ids[i], feat, P[10], big128), 8 + 4 * F + 4 * 128),
        "slsh_index 24 planes": (lambda i: ops.slsh_index(ids[i], feat, P[24], N), 8 + 4 * F + 8),
        "gather_rows": (lambda i: ops.gather_rows(ids[i], feat), 8 + 8 * D),
        "gather_mean k=2": (lambda i: ops.gather_mean(idx2[i], feat, 2), 16 + 8 * D + 4 * D),
        "rowdot": (lambda i: ops.rowdot(users, out64), 8 * D + 4),
    }
    for name, (fn, bytes_per) in cases.items():
        if args.only and args.only not in name:
            continue
        with torch.no_grad():
            us = timeit(fn, sets)
        gbs = B * bytes_per / us / 1e3
        print(json.dumps({"case": name, "lookups": B, "us_per_call": round(us, 1), "us_per_65536": round(us * 65536 / B, 2),
                          "bytes_moved_per_lookup": bytes_per, "GB_per_s_moved": round(gbs, 1), "frac_of_hbm_peak_moved": round(gbs / 8000, 3)}),
              flush=True)


if __name__ == "__main__":
    main()
