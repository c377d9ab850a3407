"""Developer probe: where does the host-clocked region of `bench.py --steps 20` go?  One persistent launch of 20
batches (~115 us of kernel) between two fences; every host-side segment timed with perf_counter."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mi_oov  # noqa: F401
from mi_oov import ops

dev = torch.device("cuda", 0)
N, B, K = 10_000_000, 65536, 20
g = torch.Generator(device=dev).manual_seed(1)
feat = torch.nn.functional.normalize(torch.randn((N, 64), generator=g, device=dev), dim=-1)
planes = torch.randn((8, 64), generator=g, device=dev)
buckets = torch.randn((8, 64), generator=g, device=dev)
ids = torch.randint(0, N, (64 + K * 12, B), generator=g, device=dev)
users = torch.randn((64, B, 64), generator=g, device=dev)
scores = torch.empty((64, B), device=dev)
n = ids.shape[0]
q = ops.LshBatchQueue([ids[i] for i in range(n)], [users[i % 64] for i in range(n)], [scores[i % 64] for i in range(n)])
scorer = ops.LshMultiScorer(feat, planes, buckets)
t_end = time.time() + 1.0
while time.time() < t_end:
    scorer.run(q, 0, 64)
    torch.cuda.synchronize()
rows = []
for rep in range(12):
    i0 = 64 + rep * K
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    t1 = time.perf_counter()
    scorer.run(q, i0, K)
    t2 = time.perf_counter()
    e1.record()
    t3 = time.perf_counter()
    while not e1.query():
        pass
    t4 = time.perf_counter()
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    rows.append([(b - a) * 1e6 for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t0, t5))] + [e0.elapsed_time(e1) * 1e3])
print("us: record0  launch  record1  spin-until-done  final-sync  TOTAL | HIP events")
for r in rows:
    print("   " + "  ".join(f"{v:8.1f}" for v in r[:6]) + f" | {r[6]:8.1f}")
