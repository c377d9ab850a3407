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


def loop_time(launches):
    res = []
    for rep in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for fn in launches:
            fn()
        b.record()
        torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / len(launches) * 1e3)
    return res


# the same loop through the validated-once forms: LshScorer (static operands checked once) and LshScorer.bind (every
# argument converted once: a call descriptor per preallocated buffer triple)
scorer = ops.LshScorer(feat, planes, buckets)
sbuf = torch.empty((8, B), device=dev)
with torch.no_grad():
    res = loop_time([(lambda i=i: scorer(ids[i], users[i % 8], sbuf[i % 8])) for i in range(K)])
    print("LshScorer, us per launch:", " ".join(f"{v:.2f}" for v in res))
    keep = [(ids[i], users[i % 8], sbuf[i % 8]) for i in range(K)]
    bound = [scorer.bind(*t) for t in keep]
    res = loop_time(bound)
    print("LshScorer.bind, us per launch:", " ".join(f"{v:.2f}" for v in res))
    t0 = time.perf_counter()
    for fn in bound:
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("LshScorer.bind host time per call: %.2f us" % ((t1 - t0) / K * 1e6))

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
