import sys, time, torch
sys.path.insert(0, '.')
import mi_oov
from mi_oov import ops
dev = torch.device('cuda:0')
N, B = 1_000_000, 65536
g = torch.Generator(device=dev).manual_seed(0)
feat = torch.randn((N, 64), generator=g, device=dev)
planes = torch.randn((8, 64), generator=g, device=dev); buckets = torch.randn((8, 64), generator=g, device=dev)
ids = torch.randint(0, N, (B,), generator=g, device=dev)
emb = torch.randn((B, 64), generator=g, device=dev)
table = feat[:N // 2]
def t(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
scorer = ops.LshScorer(feat, planes, buckets)
sbuf = torch.empty((B,), device=dev)
assert torch.equal(torch.nan_to_num(scorer(ids, emb)), torch.nan_to_num(ops.lsh_embed_score(ids, feat, planes, buckets, emb)))
for name, fn in [("lsh_embed_score", lambda: ops.lsh_embed_score(ids, feat, planes, buckets, emb)),
                 ("LshScorer", lambda: scorer(ids, emb)),
                 ("LshScorer(out=)", lambda: scorer(ids, emb, sbuf)),
                 ("lsh_lookup", lambda: ops.lsh_lookup(ids, table, feat, planes, buckets)),
                 ("mapper_map", lambda: ops.mapper_map(ids, "3round", N // 2, 1000)),
                 ("broadcast_rows", lambda: ops.broadcast_rows(emb[0], B)),
                 ("rowdot", lambda: ops.rowdot(emb, emb)),
                 ("torch.empty", lambda: torch.empty((B, 64), device=dev)),
                 ("gather_rows", lambda: ops.gather_rows(ids, feat))]:
    h, tot = t(fn)
    print(f"{name:18s} host {h:7.2f} us/call   incl. drain {tot:7.2f} us/call")
