"""Developer probe: host time of every stage of ONE sharded exchange (one-rank RCCL group), and the GPU time of the chain."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import mi_oov  # noqa: F401
from mi_oov import ops, sharded

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29545")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
N, B, S = 10_000_000, 65536, 20
g = torch.Generator(device=dev).manual_seed(1)
feat = torch.nn.functional.normalize(torch.randn((N, 64), generator=g, device=dev), dim=-1)
planes, buckets = torch.randn((8, 64), generator=g, device=dev), torch.randn((8, 64), generator=g, device=dev)
ids = [torch.randint(0, N, (S * B,), generator=g, device=dev) for _ in range(12)]
oth = torch.randn((S * B, 64), generator=g, device=dev)
sc = torch.empty((S * B,), device=dev)
table = sharded.ShardedLSHTable(feat, N, cap_factor=1.0, uniform_batches=True)
for i in range(3):
    table.embed_score(ids[i], planes, buckets, oth, score_out=sc)
torch.cuda.synchronize()
pc = time.perf_counter
print("us: bucket  a2a_ids  owner  a2a_codes  requester | host total | GPU chain (events)")
for i in range(3, 12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record(); torch.cuda.synchronize()
    t = [pc()]
    e0.record()
    p = sharded._Pending()
    p.B = ids[i].numel(); p.cap = table.capacity(p.B)
    send, p.slot, p.counts = table.prims.bucket(ids[i], table.n_rows, table.per, table.world, p.cap, table.overflow)
    t.append(pc())
    p.recv = torch.empty_like(send)
    w = dist.all_to_all_single(p.recv, send, async_op=True)
    t.append(pc())
    w.wait()
    codes = table.prims.codes(p.recv.view(-1), table.feat_local, planes)
    t.append(pc())
    p.back = torch.empty_like(codes)
    w = dist.all_to_all_single(p.back, codes, async_op=True)
    t.append(pc())
    w.wait()
    table.prims.codes_embed(p.back, p.slot, buckets, oth, False, sc)
    t.append(pc())
    e1.record()
    torch.cuda.synchronize()
    d = [(b - a) * 1e6 for a, b in zip(t, t[1:])]
    print("   " + "  ".join(f"{v:7.1f}" for v in d) + f" | {sum(d):7.1f} | {e0.elapsed_time(e1) * 1e3:7.1f}")
dist.destroy_process_group()
