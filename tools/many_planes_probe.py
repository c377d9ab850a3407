#!/usr/bin/env python3
"""lsh with MANY hyperplanes (n_oov_buckets = H: lsh_embedder.py:108-114 -- a model with 100 or 1000 OOV buckets has that many
planes): us per 65536 lookups on the generic kernel, against the two dense products' flop count.  Developer probe, GPU box."""
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
N, B, F, D = 1_000_000, 65536, 64, 64
feat = torch.randn((N, F), generator=g, device=dev)
ids = torch.randint(0, N, (5, B), generator=g, device=dev)
for H in (8, 32, 33, 64, 100, 256, 1000):
    planes = torch.randn((H, F), generator=g, device=dev)
    W = torch.randn((H, D), generator=g, device=dev)
    with torch.no_grad():
        us = timeit(lambda i: ops.lsh_embed(ids[i], feat, planes, W), 5)
        us_bits = timeit(lambda i: ops.lsh_bits(ids[i], feat, planes), 5)
    flop = 2.0 * B * H * (F + D)
    print(json.dumps({"H": H, "us_embed": round(us, 1), "us_bits": round(us_bits, 1), "TFLOP_per_s_embed": round(flop / us / 1e6, 2)}), flush=True)
