#!/usr/bin/env python3
"""BASELINE config 4 on ONE GPU at its stated size: 100 M feature rows x 64 (25.6 GB), slsh with 27 planes, 128-d rows
-- us per 65536 lookups, single launch and 20 queued batches, and the same at 10 M rows (does the 25.6 GB table cost more
per gather than the 2.56 GB one: TLB reach).  Developer probe, GPU box.
Round 4: 10 M rows 20.8 us single (eager) / 213.8 us per 20 batches; 100 M rows 17.3 / 194.7 -- the larger table costs nothing extra."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import mi_oov  # noqa: E402,F401
from mi_oov import ops  # noqa: E402
from large_calls import timeit  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
B, F, D, K = 65536, 64, 128, 20
for N in (10_000_000, 100_000_000):
    feat = torch.empty((N, F), device=dev)
    for lo in range(0, N, 10_000_000):
        feat[lo:lo + 10_000_000].normal_(generator=g)
    H = 27
    planes = torch.randn((H, F), generator=g, device=dev)
    table = torch.randn((1000, D), generator=g, device=dev)  # (the reference's arithmetic reaches H + 1 rows of it whatever its size)
    ids = torch.randint(0, N, (5, K, B), generator=g, device=dev)
    with torch.no_grad():
        one = timeit(lambda i: ops.slsh_embed(ids[i, 0], feat, planes, table), 5)
        outs = [torch.empty((B, D), device=dev) for _ in range(K)]
        many = timeit(lambda i: ops.slsh_embed_multi([ids[i, k] for k in range(K)], feat, planes, table, out=outs), 5)
    moved = 8 + 4 * F + 4 * D
    print(json.dumps({"N": N, "table_GB": round(N * F * 4 / 1e9, 2), "slsh_single_us": round(one, 1), "slsh_20_batches_us": round(many, 1),
                      "slsh_20_frac_of_hbm_peak_moved": round(K * B * moved / many / 1e3 / 8000, 3)}), flush=True)
    del feat
    torch.cuda.empty_cache()
