import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
import mi_oov
from mi_oov import ops
sys.path.insert(0, "/root/repo/oracle")
import oov_oracle as oracle
dev = torch.device("cuda:0")
rng = np.random.default_rng(1)
for (B, N, F, H, D) in ((500, 300, 64, 300, 64), (200, 100, 20, 1000, 32)):
    feat = rng.standard_normal((N, F), dtype=np.float32); planes = rng.standard_normal((H, F), dtype=np.float32)
    W = rng.standard_normal((H, D), dtype=np.float32); g = rng.standard_normal((B, D), dtype=np.float32)
    ids = rng.integers(0, N, size=B, dtype=np.int64)
    Wt = torch.from_numpy(W).to(dev).requires_grad_(True)
    out = ops.lsh_embed(torch.from_numpy(ids).to(dev), torch.from_numpy(feat).to(dev), torch.from_numpy(planes).to(dev), Wt)
    out.backward(torch.from_numpy(g).to(dev))
    _, bits = oracle.lsh_embed(ids, feat, planes, W, want_bits=True)
    want = oracle.lsh_embed_backward(bits, g)
    got = Wt.grad.cpu().numpy()
    print(B, H, "grad equal bits:", np.array_equal(got.view(np.uint32), want.view(np.uint32)), "max abs diff", float(np.abs(got - want).max()))
