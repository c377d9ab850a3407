#!/usr/bin/env python3
"""Developer tool: one lsh call (rows / fused score / lookup rows) at several batch sizes, HIP-graph replays of fresh id
batches -- run once per value of MI_OOV_PERSIST_MIN_B (read once per process) to find where the persistent kernel takes
over from the one-tile-per-wave kernel.   python3 tools/single_sizes.py [sizes...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mi_oov  # noqa: E402,F401
from mi_oov import ops  # noqa: E402
from tune import timeit  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [65536, 131072, 262144, 524288, 1048576]
    dev = torch.device("cuda:0")
    N, F, D, H = 10_000_000, 64, 64, 8
    g = torch.Generator(device=dev).manual_seed(0)
    feat = torch.nn.functional.normalize(torch.randn((N, F), generator=g, device=dev), dim=-1)
    planes = torch.randn((H, F), generator=g, device=dev)
    buckets = torch.randn((H, D), generator=g, device=dev)
    for B in sizes:
        n_it = max(8, min(40, (1 << 25) // B))
        ids = torch.randint(0, N, (n_it + 5, B), generator=g, device=dev)
        users = torch.randn((4, B, D), generator=g, device=dev)
        with torch.no_grad():
            for name, fn, bpu in (
                    ("lsh_embed rows", lambda i: ops.lsh_embed(ids[i], feat, planes, buckets), 8 + 4 * F + 4 * D),
                    ("lsh_embed_score", lambda i: ops.lsh_embed_score(ids[i], feat, planes, buckets, users[i % 4]), 16 + 4 * F + 4 * D + 4),
                    ("lsh_lookup rows 50% oov", lambda i: ops.lsh_lookup(ids[i], feat[:N // 2], feat, planes, buckets), 8 + 4 * F + 4 * D)):
                us = timeit(fn, n_it)
                print(json.dumps({"case": name, "B": B, "persist_min_b": os.environ.get("MI_OOV_PERSIST_MIN_B", "default"),
                                  "us": round(us, 2), "us_per_65536": round(us * 65536 / B, 3),
                                  "frac_of_8TBs": round(B * bpu / us / 1e3 / 8000, 3)}), flush=True)
        del ids, users


if __name__ == "__main__":
    main()
