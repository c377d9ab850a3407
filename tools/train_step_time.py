"""Developer probe: wall time of one BPR + lsh OOV training step (calculate_loss + backward + Adam) at the
reference's batch size, with the sync-free lookup (default) and with the reference-shaped mask / gather / splice
sequence (MI_OOV_TRAIN_LOOKUP=0).   python tools/train_step_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mi_oov

dev = torch.device("cuda", 0)
n_users, n_items, n_new, D, H, B = 200_000, 300_000, 400_000, 64, 8, 2048
PAD = 112062759511


class Cfg(dict):
    def __getitem__(self, k):
        return self.get(k, None)


class DS:
    def num(self, f):
        return {"user_id": n_users, "item_id": n_items}[f]


g = torch.Generator().manual_seed(0)
# all-positive features + an all-positive first plane: bit 0 is always set, no all-zero codes (NaN rows abort a step)
ft_u = mi_oov.FeatureTable({"id": torch.arange(n_new), "f": torch.randn((n_new, 64), generator=g).abs()})
ft_i = mi_oov.FeatureTable({"id": torch.arange(n_new), "f": torch.randn((n_new, 64), generator=g).abs()})
emb = mi_oov.LSHInductiveEmbedder(ft_u, ft_i, n_users, n_items, H, H, D, dev, PAD, "none", mi_oov.InductiveFeatureCache())
cfg = Cfg(USER_ID_FIELD="user_id", ITEM_ID_FIELD="item_id", NEG_PREFIX="neg_", device=dev, embedding_size=D,
          add_oov_buckets=True, user_oov_buckets=H, item_oov_buckets=H, oov_freeze_embedding=False)
with torch.no_grad():
    emb.user_lsh.uniform_planes[0][0].abs_()
    emb.item_lsh.uniform_planes[0][0].abs_()
model = mi_oov.BPR(cfg, DS(), None, emb).to(dev)
model.train()
emb.set_train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
gd = torch.Generator(device=dev).manual_seed(1)


def batch():
    u = torch.randint(1, n_users, (B,), generator=gd, device=dev)
    p = torch.randint(1, n_items, (B,), generator=gd, device=dev)
    n = torch.randint(1, n_items, (B,), generator=gd, device=dev)
    pad = torch.rand((B,), generator=gd, device=dev) < 0.2  # oov_train_ratio: simulate OOV by prime-padding
    return {"user_id": torch.where(pad, u + PAD, u), "item_id": torch.where(pad, p + PAD, p), "neg_item_id": n}


finite = 0


def step():
    global finite
    loss = model.calculate_loss(batch())
    if torch.isfinite(loss):
        finite += 1
        opt.zero_grad()
        loss.backward()
        opt.step()


if "--graph" in sys.argv:
    # whole-step capture (the sync-free lookups make the step capturable): static batch buffers, capturable Adam, the
    # trainer's NaN check moved out of the step (it is a device -> host sync; here it runs once at the end)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)
    static = batch()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            model.calculate_loss(static).backward()
            opt.step()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(gr):
        static_loss = model.calculate_loss(static)
        static_loss.backward()
        opt.step()

    def graphed_step():
        nb = batch()
        for k_ in static:
            static[k_].copy_(nb[k_])
        gr.replay()

    for _ in range(20):
        graphed_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        graphed_step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    print(f"graphed step: {dt:.3f} ms per training step (B={B}), final loss finite: {bool(torch.isfinite(static_loss))}")
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        gr.replay()
    b.record()
    torch.cuda.synchronize()
    print(f"graph replay alone (same batch): {a.elapsed_time(b) / n:.3f} ms per step")
    sys.exit(0)

for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 200
for _ in range(n):
    step()
torch.cuda.synchronize()
print(f"finite steps {finite}/{n + 20}; MI_OOV_TRAIN_LOOKUP={os.environ.get('MI_OOV_TRAIN_LOOKUP', '1')}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per training step (B={B})")
