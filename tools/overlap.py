"""Developer probe: throughput of the hot path when consecutive batches are ALLOWED to overlap.

bench.py's contract serialises the K steps (one stream / one graph chain): every launch pays its own
dispatch -> ids -> rows -> store chain.  A serving loop that keeps C independent batches in flight (C
streams, one HIP graph chain each) hides the head of one launch under the tail of another.  Same kernels,
same id batches; prints us per step and lookups/s for C = 1..4.   python tools/overlap.py [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mi_oov  # noqa: F401
from mi_oov import ops

dev = torch.device("cuda", 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
N, B, F, H, D = 10_000_000, 65536, 64, 8, 64
g = torch.Generator(device=dev).manual_seed(1)
feat = torch.empty((N, F), device=dev)
for lo in range(0, N, 1 << 20):
    hi = min(N, lo + (1 << 20))
    feat[lo:hi] = torch.nn.functional.normalize(torch.randn((hi - lo, F), generator=g, device=dev), dim=-1)
planes, buckets = torch.randn((H, F), generator=g, device=dev), torch.randn((H, D), generator=g, device=dev)
ids = torch.randint(0, N, (K, B), generator=g, device=dev)
users = torch.randn((8, B, D), generator=g, device=dev)
scores = torch.empty((8, B), device=dev)
scorer = ops.LshScorer(feat, planes, buckets)
with torch.no_grad():
    for C in (1, 2, 3, 4):
        streams = [torch.cuda.Stream() for _ in range(C)]
        graphs = []
        for c, s in enumerate(streams):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for k in range(c, K, C):
                    scorer(ids[k], users[k % 8], score_out=scores[k % 8])
            graphs.append(gr)
        best = None
        t_end = time.time() + 1.0
        while time.time() < t_end or best is None:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for gr, s in zip(graphs, streams):
                with torch.cuda.stream(s):
                    gr.replay()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print(f"C={C} batches in flight: {best / K * 1e6:6.2f} us per step   {K * B / best / 1e9:6.2f} G lookups+scores/s   "
              f"{K * B * 532 / best / 1e12:5.2f} TB/s algorithmic")
