import sys, os, json
sys.path.insert(0, "/root/repo")
import torch
import mi_oov
from mi_oov import ops
sys.path.insert(0, "/root/repo/tools")
from large_calls import timeit
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
N, D, H = 10_000_000, 64, 8
for B in (65536, 1 << 21):
    ids = torch.randint(0, N, (5, B), generator=g, device=dev)
    users = torch.randn((B, D), generator=g, device=dev)
    for F in (4, 20, 32, 48):
        feat = torch.randn((N, F), generator=g, device=dev)
        planes = torch.randn((H, F), generator=g, device=dev)
        W = torch.randn((H, D), generator=g, device=dev)
        featp = torch.nn.functional.pad(feat, (0, 64 - F)).contiguous()
        planesp = torch.nn.functional.pad(planes, (0, 64 - F)).contiguous()
        with torch.no_grad():
            a = ops.lsh_embed(ids[0], feat, planes, W)
            b = ops.lsh_embed(ids[0], featp, planesp, W)
            same = torch.equal(torch.nan_to_num(a, 7.0), torch.nan_to_num(b, 7.0))
            t0 = timeit(lambda i: ops.lsh_embed(ids[i], feat, planes, W), 5)
            t1 = timeit(lambda i: ops.lsh_embed(ids[i], featp, planesp, W), 5)
            t2 = timeit(lambda i: ops.lsh_embed_score(ids[i], feat, planes, W, users), 5)
            t3 = timeit(lambda i: ops.lsh_embed_score(ids[i], featp, planesp, W, users), 5)
        print(json.dumps({"B": B, "F": F, "same_bits": same, "rows_generic_us": round(t0, 1), "rows_padded64_us": round(t1, 1),
                          "score_generic_us": round(t2, 1), "score_padded64_us": round(t3, 1)}), flush=True)
        del feat, featp
