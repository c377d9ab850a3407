#!/usr/bin/env python3
"""Accuracy of mi_oov_linear_x3 against f64 and against the f32 kernel (developer check; the pinned form is
tests/test_gpu_parity.py::test_linear_x3_vs_oracle)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib, torch
ops = importlib.import_module("improving-inductive-oov-recsys_amd.ops")
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
for B, K, N in ((300, 1024, 512), (1000, 512, 64), (257, 70, 130), (64, 22, 512), (5, 1030, 33)):
    X = (torch.rand((B, K), generator=g) * 2 - 1).to(dev)
    W = (torch.randn((N, K), generator=g) / K ** 0.5).to(dev)
    b = torch.randn((N,), generator=g).to(dev)
    truth = X.double() @ W.double().T + b.double()
    den = X.abs().double() @ W.abs().double().T + b.abs().double()
    for name, f in (("f32", ops.linear_act), ("x3", ops.linear_act_x3)):
        y = f(X, W, b, None)
        e = (y.double() - truth).abs() / den
        print(f"{B}x{K}->{N} {name}: max err/sum|xw| {e.max().item():.3e} rms {e.pow(2).mean().sqrt().item():.3e}")
    for act in ("gelu", "sigmoid"):
        d = (ops.linear_act_x3(X, W, b, act) - ops.linear_act(X, W, b, act)).abs().max().item()
        print(f"   {act}: max |x3 - f32| {d:.3e}")
