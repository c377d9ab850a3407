#!/usr/bin/env python3
"""Developer probe: wall time of one BPR + dhe training step (calculate_loss + backward + Adam; K = 1024 SipHash-2-4 hashes
per id -> the 1024-512-512-512-64 net for OOV users / items) at the reference's batch size, with the hash nets on the
split-bf16 GEMM (default) and on the f32 matrix instruction (MI_OOV_LINEAR_X3=0).   python tools/train_step_dhe_bpr.py"""
import hashlib, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mi_oov

dev = torch.device("cuda", 0)
n_users, n_items, n_new, D, K, B = 200_000, 300_000, 400_000, 64, 1024, 2048
PAD = 112062759511


class Cfg(dict):
    def __getitem__(self, k):
        return self.get(k, None)


class DS:
    def num(self, f):
        return {"user_id": n_users, "item_id": n_items}[f]


os.chdir(tempfile.mkdtemp())
os.makedirs("hash_keys")
json.dump([hashlib.sha256(b"mi-oov-key-%d" % j).digest()[:16].hex() for j in range(K)], open(f"hash_keys/{K}.hashes", "w"))
ft = mi_oov.FeatureTable({"id": torch.arange(4), "f": torch.zeros(4)})
torch.manual_seed(0)
emb = mi_oov.DeepHashEmbedder(ft, ft, n_users, n_items, 8, 8, D, dev, PAD, K)
for m in emb.modules():  # raw hashes (< 2^24) feed the first layer: small weights keep the untrained net finite
    if isinstance(m, torch.nn.Linear):
        torch.nn.init.normal_(m.weight, std=1e-4 if m.in_features == K else 0.05)
cfg = Cfg(USER_ID_FIELD="user_id", ITEM_ID_FIELD="item_id", NEG_PREFIX="neg_", device=dev, embedding_size=D,
          add_oov_buckets=True, user_oov_buckets=8, item_oov_buckets=8, oov_freeze_embedding=False)
model = mi_oov.BPR(cfg, DS(), None, emb).to(dev)
model.train()
emb.set_train()
opt = torch.optim.Adam(model.parameters(), lr=1e-4)
gd = torch.Generator(device=dev).manual_seed(1)


def batch():
    u = torch.randint(1, n_users, (B,), generator=gd, device=dev)
    p = torch.randint(1, n_items, (B,), generator=gd, device=dev)
    n = torch.randint(1, n_items, (B,), generator=gd, device=dev)
    pad = torch.rand((B,), generator=gd, device=dev) < 0.2  # oov_train_ratio: simulate OOV by prime-padding
    return {"user_id": torch.where(pad, u + PAD, u), "item_id": torch.where(pad, p + PAD, p), "neg_item_id": n}


def step():
    loss = model.calculate_loss(batch())
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss


for mode in ("1", "0", "1"):
    os.environ["MI_OOV_LINEAR_X3"] = mode
    for _ in range(15):
        step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            loss = step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 10)
    print(f"MI_OOV_LINEAR_X3={mode}: {best * 1e3:.2f} ms per step (B = {B}, K = {K}), loss {loss.item():.4f}")
