"""Developer probe: can one sharded exchange (bucket -> all_to_all -> owner -> all_to_all -> requester) be captured into
a HIP graph and replayed on this stack (RCCL through torch.distributed)?  One-rank RCCL group on cuda:0.

RESULT (ROCm 7.0.2, RCCL 2.26.6, torch 2.10): NO -- capturing the collectives ends in a segmentation fault of the
process (gpurun log of round 2).  The sharded path therefore issues its exchanges eagerly (~120 us of host time per
exchange, two collectives each), which is why one exchange carries many steps.  Kept to re-check on a newer stack;
expect it to crash."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import mi_oov  # noqa: F401
from mi_oov import ops, sharded

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
N, B, S = 2_000_000, 65536, 5
g = torch.Generator(device=dev).manual_seed(1)
feat = torch.nn.functional.normalize(torch.randn((N, 64), generator=g, device=dev), dim=-1)
planes, buckets = torch.randn((8, 64), generator=g, device=dev), torch.randn((8, 64), generator=g, device=dev)
ids = [torch.randint(0, N, (S * B,), generator=g, device=dev) for _ in range(4)]
oth = [torch.randn((S * B, 64), generator=g, device=dev) for _ in range(4)]
sc = [torch.empty((S * B,), device=dev) for _ in range(4)]
table = sharded.ShardedLSHTable(feat, N, cap_factor=1.0, uniform_batches=True)
pipe = sharded.LshPipeline(table, planes, buckets)
pipe.run(ids, oth, sc)  # eager warm-up (RCCL communicator, allocator)
torch.cuda.synchronize()
want = [s.clone() for s in sc]
for s in sc:
    s.zero_()
t0 = time.perf_counter()
for _ in range(10):
    pipe.run(ids, oth, sc)
torch.cuda.synchronize()
print("eager: %.1f us per 4 exchanges of %d batches" % ((time.perf_counter() - t0) / 10 * 1e6, S))
gr = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(gr):
        pipe.run(ids, oth, sc)
    for s in sc:
        s.zero_()
    gr.replay()
    torch.cuda.synchronize()
    ok = all(torch.equal(torch.nan_to_num(a, 7.0), torch.nan_to_num(b, 7.0)) for a, b in zip(sc, want))
    t0 = time.perf_counter()
    for _ in range(10):
        gr.replay()
    torch.cuda.synchronize()
    print("graph: capture ok, replay == eager: %s, %.1f us per replay" % (ok, (time.perf_counter() - t0) / 10 * 1e6))
except Exception as e:  # noqa: BLE001
    print("graph capture failed:", type(e).__name__, str(e)[:300])
dist.destroy_process_group()
