"""Developer tool: per-launch time of the path's kernels at the BASELINE shape, each measured by replaying a
HIP graph of K launches over K distinct id batches (reproducible to ~0.01 us; no host launch cost).
    python tools/graph_time.py [K] [name-substring ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mi_oov  # noqa: F401
from mi_oov import ops

dev = torch.device("cuda", 0)
args = sys.argv[1:]
K = int(args.pop(0)) if args and args[0].isdigit() else 1000
only = args
N, B, F, H, D = 10_000_000, 65536, 64, 8, 64
g = torch.Generator(device=dev).manual_seed(1)
feat = torch.empty((N, F), device=dev)
for lo in range(0, N, 1 << 20):
    hi = min(N, lo + (1 << 20))
    feat[lo:hi] = torch.nn.functional.normalize(torch.randn((hi - lo, F), generator=g, device=dev), dim=-1)
planes, buckets = torch.randn((H, F), generator=g, device=dev), torch.randn((H, D), generator=g, device=dev)
planes16, buckets16 = torch.randn((16, F), generator=g, device=dev), torch.randn((16, D), generator=g, device=dev)
ids = torch.randint(0, N, (K, B), generator=g, device=dev)
ids_mixed = torch.randint(0, 2 * N, (K, B), generator=g, device=dev)  # half in-vocabulary, half OOV for the lookups
users = torch.randn((8, B, D), generator=g, device=dev)
table = torch.randn((N, D), generator=g, device=dev)
big_buckets = torch.randn((1000, D), generator=g, device=dev)
planes10 = torch.randn((10, F), generator=g, device=dev)
feat2 = torch.cat([feat, feat[: N]], 0) if False else None

cases = {
    "lsh_embed_score (fused, headline)": (lambda k: ops.lsh_embed_score(ids[k], feat, planes, buckets, users[k % 8]), 532),
    "lsh_embed (rows stored)": (lambda k: ops.lsh_embed(ids[k], feat, planes, buckets), 520),
    "lsh_bits (codes only)": (lambda k: ops.lsh_bits(ids[k], feat, planes), 8 + 256 + 8),
    "lsh_embed H=16 (generic kernel)": (lambda k: ops.lsh_embed(ids[k], feat, planes16, buckets16), 520),
    "slsh_embed nb=8": (lambda k: ops.slsh_embed(ids[k], feat, planes[:3], buckets), 8 + 256 + 512),
    "slsh_embed nb=1000": (lambda k: ops.slsh_embed(ids[k], feat, planes10, big_buckets), 8 + 256 + 512),
    "gather_rows": (lambda k: ops.gather_rows(ids[k], table), 8 + 512),
    "rowdot": (lambda k: ops.rowdot(users[k % 8], users[(k + 1) % 8]), 516),
    "mapper_map 3round": (lambda k: ops.mapper_map(ids[k], "3round", N // 2, 1000), 16),
}
with torch.no_grad():
    for name, (fn, bytes_per) in cases.items():
        if only and not any(o in name for o in only):
            continue
        for k in range(3):
            fn(k)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        keep = []
        with torch.cuda.graph(gr):
            for k in range(K):
                keep.append(fn(k)) if k < 8 else fn(k)
        ts = []
        t_end = time.time() + 0.7
        while time.time() < t_end or len(ts) < 5:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); gr.replay(); b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / K * 1e3)
        us = sorted(ts[-5:])[2]
        print(f"{name:36s} {us:8.2f} us   {B * bytes_per / us / 1e6:6.2f} TB/s algorithmic")
        del gr, keep
