#!/usr/bin/env python3
"""Config 3: synthetic dhe, K=1024 SipHash-2-4 per id -> MLP(1024,512,512,512,64), B=65536."""
import hashlib, os, sys, json, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mi_oov
from mi_oov import ops

dev = torch.device("cuda:0")
K, B, D, N = 1024, 65536, 64, 10_000_000
tmp = tempfile.mkdtemp(); os.chdir(tmp); os.makedirs("hash_keys")
json.dump([hashlib.sha256(b"mi-oov-key-%d" % j).digest()[:16].hex() for j in range(K)], open(f"hash_keys/{K}.hashes", "w"))
ft = mi_oov.FeatureTable({"id": torch.arange(4), "f": torch.zeros(4)})
torch.manual_seed(0)
emb = mi_oov.DeepHashEmbedder(ft, ft, 2, 2, 8, 8, D, dev, 112062759511, K)
for m in emb.modules():
    if isinstance(m, torch.nn.Linear):
        torch.nn.init.xavier_normal_(m.weight); torch.nn.init.zeros_(m.bias)
g = torch.Generator(device=dev).manual_seed(3)
ids = torch.randint(N // 2, N, (20, B), generator=g, device=dev)

def timeit(fn, n=10):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(3 + i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

with torch.no_grad():
    t_hash = timeit(lambda i: emb._hash_ids(ids[i]))
    h = emb._hash_ids(ids[0])
    t_mlp = timeit(lambda i: emb.item_hash_net(h))
    t_all = timeit(lambda i: emb.embed_item_ids(ids[i], None))

flop = 2 * (512 * K + 2 * 512 * 512 + 512 * D) * B
print(json.dumps({"siphash_ms": round(t_hash, 3), "mlp_ms": round(t_mlp, 3), "embed_ms": round(t_all, 3),
                  "hashes_per_s": B * K / t_hash * 1e3, "mlp_TFLOPs": flop / t_mlp / 1e9, "lookups_per_s": B / t_all * 1e3}))
