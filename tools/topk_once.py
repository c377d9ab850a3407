"""Developer probe: a few calls of the fused top-k at the headline shape (4096 x 50 000, k = 20), for builds that print."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mi_oov  # noqa: F401
from mi_oov import ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
U = torch.randn((4096, 64), generator=g, device=dev)
E = torch.randn((50000, 64), generator=g, device=dev)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    ops.score_topk(U, E, 20, 1)
    torch.cuda.synchronize()
