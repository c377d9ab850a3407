"""Developer probe: time one fused top-k shape: python tools/topk_case.py B N k [iters] [D]  (env knobs apply)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mi_oov  # noqa: F401
from mi_oov import ops
B, N, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
D = int(sys.argv[5]) if len(sys.argv) > 5 else 64
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
U = torch.randn((B, D), generator=g, device=dev)
E = torch.randn((N, D), generator=g, device=dev)
ops.score_topk(U, E, k, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    ops.score_topk(U, E, k, 1)
torch.cuda.synchronize()
print(f"B={B} N={N} k={k} D={D}: {(time.perf_counter() - t0) / iters * 1e6:.1f} us per call")
