#!/usr/bin/env python3
"""Measured error of the hash nets' pre-sigmoid activations (dhe / fdhe fixtures of the REAL reference, tests/golden/) on
both layer kernels -- mi_oov_linear_act (the oracle's f32 fmaf chain) and mi_oov_linear_x3 (split bf16) -- against the
reference's own output and against an f64 evaluation of the same net (the witness: how far the reference itself is from
the exact value).  GPU box:  python3 tools/dhe_error.py  -> one JSON line per fixture / side."""
import json
import os
import sys

import numpy as np
from scipy.special import erf

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mi_oov  # noqa: E402,F401
from mi_oov import ops  # noqa: E402

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def net64(x, Ws, bs):
    x = x.astype(np.float64)
    for j, (W, b) in enumerate(zip(Ws, bs)):
        x = x @ W.astype(np.float64).T + b.astype(np.float64)
        if j < len(Ws) - 1:
            x = 0.5 * x * (1 + erf(x / np.sqrt(2)))
    return x


def run(name, x, Ws, bs, ref):
    dev = torch.device("cuda:0")
    f64 = net64(x, Ws, bs)
    m = np.abs(ref).max()
    out = {"case": name, "max_abs_ref": float(m),
           "reference_vs_f64": {"over_max": float(np.abs(ref - f64).max() / m), "elementwise_rel": float((np.abs(ref - f64) / np.abs(f64)).max())}}
    for kern, fn in (("linear_act (f32 chain)", ops.linear_act), ("linear_x3 (split bf16)", ops.linear_act_x3)):
        h = torch.from_numpy(x.astype(np.float32)).to(dev)
        for j, (W, b) in enumerate(zip(Ws, bs)):
            if fn is ops.linear_act_x3 and h.shape[1] % 16:
                h = torch.nn.functional.pad(h, (0, -h.shape[1] % 16))
            h = fn(h, torch.from_numpy(W).to(dev), torch.from_numpy(b).to(dev), "gelu" if j < len(Ws) - 1 else None)
        got = h.cpu().numpy().astype(np.float64)
        out[kern] = {"vs_reference_over_max": float(np.abs(got - ref).max() / m),
                     "vs_reference_elementwise_rel": float((np.abs(got - ref) / np.abs(ref)).max()),
                     "vs_f64_over_max": float(np.abs(got - f64).max() / m),
                     "vs_f64_elementwise_rel": float((np.abs(got - f64) / np.abs(f64)).max())}
    print(json.dumps(out), flush=True)


def main():
    z = np.load(os.path.join(G, "dhe.npz"))
    Ws = [z[f"item_hash_net__{i}__weight"] for i in (0, 2, 4, 6)]
    bs = [z[f"item_hash_net__{i}__bias"] for i in (0, 2, 4, 6)]
    run("dhe item (K = 16, 512-wide, 12 ids)", z["hashes"], Ws, bs, z["item_pre_sigmoid"])
    z = np.load(os.path.join(G, "fdhe.npz"))
    for mode in ("eval", "train"):
        for side in ("user", "item"):
            Ws = [z[f"sd__{side}_hash_net__{i}__weight"] for i in (0, 2, 4, 6)]
            bs = [z[f"sd__{side}_hash_net__{i}__bias"] for i in (0, 2, 4, 6)]
            run(f"fdhe {mode} {side} (K + F = {Ws[0].shape[1]}, 96-wide, 10 ids)", z[f"{mode}_{side}_input"], Ws, bs, z[f"{mode}_{side}_pre_sigmoid"])


if __name__ == "__main__":
    main()
