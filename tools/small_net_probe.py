#!/usr/bin/env python3
"""The hash nets at SMALL widths (dnn / fdhe with dhe_layer_size 32-128 on a 20-column feature matrix: shapes outside the
pipelined 256 x 256 layer kernel): us per 65536 lookups (eager, four launches) and the achieved flop rate.  Developer probe, GPU box.
Round 4: (F, layer) = (20, 32) 45 us, (20, 96) 75, (20, 128) 86, (20, 256) 163, (1024, 96) 172, (1024, 512) 759 -- the narrow nets are
bound by their [B, layer] activations going through HBM between the four launches (a whole-net kernel with the activations in
LDS would apply up to ~128 columns; not built)."""
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
B, D = 65536, 64
for F, L in ((20, 32), (20, 96), (20, 128), (64, 128), (20, 256), (1024, 96), (1024, 512)):
    Kp = -(-F // 16) * 16
    x = torch.randn((B, Kp), generator=g, device=dev)
    x[:, F:] = 0
    dims = [(Kp, L), (L, L), (L, L), (L, D)]
    Ws = [torch.randn((o, i), generator=g, device=dev) / i ** 0.5 for i, o in dims]
    bs = [torch.zeros((o,), device=dev) for _, o in dims]
    xw = [ops.LinearX3Weights(w) for w in Ws]

    def net(_):
        h = x
        for j, (w, b) in enumerate(zip(Ws, bs)):
            h = ops.linear_act_x3(h, w, b, "gelu" if j < 3 else "sigmoid", xw[j])
        return h

    with torch.no_grad():
        us = timeit(net, 1)
    flop = 2.0 * B * sum(i * o for i, o in dims)
    print(json.dumps({"F": F, "layer": L, "us_per_65536": round(us, 1), "TFLOP_per_s": round(flop / us / 1e6, 1)}), flush=True)
