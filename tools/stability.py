"""Developer probe: is the per-launch time of the hot kernel stable within a process / across processes?
Prints the HIP-event time of 10 consecutive 2000-launch regions, plus where the big tensors landed."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mi_oov
from mi_oov import ops

dev = torch.device("cuda", 0)
N, B, F, H, D = 10_000_000, 65536, 64, 8, 64
g = torch.Generator(device=dev).manual_seed(1)
feat = torch.nn.functional.normalize(torch.randn((N, F), generator=g, device=dev), dim=-1)
planes = torch.randn((H, F), generator=g, device=dev)
buckets = torch.randn((H, D), generator=g, device=dev)
K = 2000
ids = torch.randint(0, N, (K, B), generator=g, device=dev)
users = torch.randn((8, B, D), generator=g, device=dev)
print("feat ptr %x (mod 2MiB = %x)  ids ptr %x  users ptr %x" % (feat.data_ptr(), feat.data_ptr() % (2 << 20), ids.data_ptr(), users.data_ptr()))
with torch.no_grad():
    t_end = time.time() + 1.5
    while time.time() < t_end:
        for i in range(256):
            ops.lsh_embed_score(ids[i], feat, planes, buckets, users[i % 8])
        torch.cuda.synchronize()
    out = []
    for rep in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(K):
            ops.lsh_embed_score(ids[i], feat, planes, buckets, users[i % 8])
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / K * 1e3)
print("us per launch:", " ".join(f"{v:.2f}" for v in out))

# the same K launches captured once into a HIP graph (stream capture sees the C-ABI launches because they
# go to torch's current stream) and replayed: no per-launch host work at all
gr = torch.cuda.CUDAGraph()
outs = []
with torch.no_grad():
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(3):
            ops.lsh_embed_score(ids[i], feat, planes, buckets, users[i % 8])
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    t0 = time.time()
    with torch.cuda.graph(gr):
        for i in range(K):
            outs.append(ops.lsh_embed_score(ids[i], feat, planes, buckets, users[i % 8]))
    print("capture of %d launches: %.2f s" % (K, time.time() - t0))
    res = []
    for rep in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        gr.replay()
        b.record()
        torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / K * 1e3)
print("graph replay, us per launch:", " ".join(f"{v:.2f}" for v in res))
ref = ops.lsh_embed_score(ids[K - 1], feat, planes, buckets, users[(K - 1) % 8])
print("graph output == eager output:", bool(torch.equal(torch.nan_to_num(outs[-1]), torch.nan_to_num(ref))))
