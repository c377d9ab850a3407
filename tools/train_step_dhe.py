#!/usr/bin/env python3
"""Developer probe: one training step of the dhe hash net (forward + backward of the 1024-512-512-512-64 net on a batch of
hashed ids) on the split-bf16 GEMM and, with MI_OOV_LINEAR_X3=0, on the f32 matrix instruction.
    python tools/train_step_dhe.py [B=2048]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mi_oov  # noqa: F401
from mi_oov import ops

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(1024, 512), torch.nn.GELU(), torch.nn.Linear(512, 512), torch.nn.GELU(),
                          torch.nn.Linear(512, 512), torch.nn.GELU(), torch.nn.Linear(512, 64), torch.nn.Sigmoid()).to(dev)
x = torch.randint(0, 1 << 24, (B, 1024), device=dev).float() / (1 << 23) - 1.0  # (scaled: raw hashes saturate an untrained net)
tgt = torch.rand((B, 64), device=dev)


def step():
    for p in net.parameters():
        p.grad = None
    y = ops.hash_net_train(net, x)
    ((y - tgt) ** 2).mean().backward()


def timeit(n=10):
    for _ in range(20):  # (the first steps grow the caching allocator's pools and load the kernels)
        step()
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            step()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n)
    return best


res = {}
for mode in ("1", "0"):
    os.environ["MI_OOV_LINEAR_X3"] = mode
    res[mode] = timeit()
    res["g" + mode] = [p.grad.clone() for p in net.parameters()]
os.environ.pop("MI_OOV_LINEAR_X3")
worst = max(((a - b).abs().max() / b.abs().max()).item() for a, b in zip(res["g1"], res["g0"]))
print(f"B={B}: forward+backward {res['1']*1e3:.0f} us on the split-bf16 GEMM, {res['0']*1e3:.0f} us on the f32 matrix instruction; "
      f"largest gradient difference {worst:.2e} of the tensor's largest entry")
