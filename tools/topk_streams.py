#!/usr/bin/env python3
"""Developer probe: the fused top-k of ONE user batch cut into P row chunks issued on P streams (the chains of 5 kernels
then overlap: one chunk's kernel boundaries and ramps are another's work).  python3 tools/topk_streams.py [D] [B] [N] [k]
Round 4, final kernels, 4096 x 50 000, k = 20, D = 64: 1 / 2 / 3 / 4 chunks = 78.3 / 87.4 / 91.8 / 100.6 us in a graph replay (eager: 101 / 141 /
152 / 207, host-bound); D = 128: 117.2 / 121.0 / 135.1 / 142.3 -- the smaller launches keep their fixed costs and fill the chip worse: no gain."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mi_oov  # noqa: F401
from mi_oov import ops

D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
N = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
k = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
U = torch.randn((B, D), generator=g, device=dev)
E = torch.randn((N, D), generator=g, device=dev)
cat = ops.TopkCatalogue(E)
ref_v, ref_i = ops.score_topk(U, cat, k, 1)
for P in (1, 2, 3, 4, 8):
    streams = [torch.cuda.Stream(dev) for _ in range(P)]
    rows = -(-B // P)
    rows = -(-rows // 64) * 64
    def call():
        main = torch.cuda.current_stream(dev)
        outs = []
        for p, s in enumerate(streams):
            lo, hi = p * rows, min(B, (p + 1) * rows)
            if lo >= hi:
                continue
            s.wait_stream(main)
            with torch.cuda.stream(s):
                outs.append(ops.score_topk(U[lo:hi], cat, k, 1))
        for s in streams:
            main.wait_stream(s)
        return outs
    for _ in range(5):
        outs = call()
    torch.cuda.synchronize()
    v = torch.cat([o[0] for o in outs]); i = torch.cat([o[1] for o in outs])
    same = torch.equal(i, ref_i) and torch.equal(v, ref_v)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(3):
        a.record()
        for _ in range(20):
            call()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / 20)
    # the same call captured ONCE into a HIP graph (fork / join inside the capture) and replayed: no host time between launches
    gbest = float("nan")
    try:
        cap = torch.cuda.Stream(dev)
        with torch.cuda.stream(cap):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=cap):
                for _ in range(10):
                    keep = call()
        torch.cuda.synchronize()
        gr.replay()
        torch.cuda.synchronize()
        gbest = 1e9
        for rep in range(3):
            a.record()
            gr.replay()
            b.record()
            torch.cuda.synchronize()
            gbest = min(gbest, a.elapsed_time(b) * 1e3 / 10)
    except Exception as e:  # noqa: BLE001
        print("  (graph capture failed:", type(e).__name__, str(e)[:100], ")")
    print(f"D={D} B={B} N={N} k={k}: {P} chunk(s) on {P} stream(s): {best:.1f} us per batch eager, {gbest:.1f} us in a graph replay, identical results: {same}", flush=True)
