#!/usr/bin/env python3
"""fdhe at BASELINE's size (developer tool): K = 1024 SipHash-2-4 hashes of the id ++ 22 feature columns -> MLP, 65536 lookups.
Times the embedder (hashes written straight into the net's padded input) against hstack + the net."""
import hashlib, json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mi_oov
from mi_oov import ops, embedders

dev = torch.device("cuda:0")
K, B, D, N, F = 1024, 65536, 64, 1_000_000, 22
os.chdir(tempfile.mkdtemp()); os.makedirs("hash_keys")
json.dump([hashlib.sha256(b"mi-oov-key-%d" % j).digest()[:16].hex() for j in range(K)], open(f"hash_keys/{K}.hashes", "w"))
g = torch.Generator().manual_seed(0)
ft = mi_oov.FeatureTable({"id": torch.arange(N), "f": torch.randn((N, F), generator=g)})
torch.manual_seed(0)
emb = mi_oov.FeatDeepHashEmbedder(ft, ft, N // 2, N // 2, 8, 8, D, dev, 112062759511, K, 512)
for m in emb.modules():
    if isinstance(m, torch.nn.Linear):
        torch.nn.init.normal_(m.weight, std=1e-4 if m.in_features > 512 else 0.05)
gd = torch.Generator(device=dev).manual_seed(3)
ids = torch.randint(N // 2, N, (20, B), generator=gd, device=dev)


def timeit(fn, n=10):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(3 + i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def old(i):
    x = ids[i]
    h = emb._hash_ids(x)
    extra = ops.gather_rows(emb._lookup(x), emb.item_feature_mat)
    return embedders._run_hash_net(emb.item_hash_net, torch.hstack((h, extra)))


with torch.no_grad():
    t_new = timeit(lambda i: emb.embed_item_ids(ids[i], None))
    t_old = timeit(old)
    same = torch.equal(emb.embed_item_ids(ids[0], None), old(0))
print(json.dumps({"fdhe_embed_ms": round(t_new, 3), "hstack_then_net_ms": round(t_old, 3), "identical": same, "lookups_per_s": B / t_new * 1e3}))
