#!/usr/bin/env python3
"""Developer tool: producer + consumer pairs (lsh_embed rows -> rowdot; gather_rows -> rowdot) as one graph replay, to see
whether a producer's non-temporal row stores cost its consumer more than they save.  Library under test: MI_LIB."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("MI_LIB"):
    shutil.copy(os.environ["MI_LIB"], os.path.join(ROOT, "improving-inductive-oov-recsys_amd", "lib", "libmi_oov.so"))
import torch  # noqa: E402

import mi_oov  # noqa: E402,F401
from mi_oov import ops  # noqa: E402
from tune import timeit  # noqa: E402

dev = torch.device("cuda:0")
N, F, D, H, B = 10_000_000, 64, 64, 8, int(os.environ.get("B", 65536))
g = torch.Generator(device=dev).manual_seed(0)
feat = torch.nn.functional.normalize(torch.randn((N, F), generator=g, device=dev), dim=-1)
planes = torch.randn((H, F), generator=g, device=dev)
buckets = torch.randn((H, D), generator=g, device=dev)
ids = torch.randint(0, N, (45, B), generator=g, device=dev)
users = torch.randn((8, B, D), generator=g, device=dev)
with torch.no_grad():
    for name, fn in (("lsh_embed", lambda i: ops.lsh_embed(ids[i], feat, planes, buckets)),
                     ("lsh_embed + rowdot", lambda i: ops.rowdot(users[i % 8], ops.lsh_embed(ids[i], feat, planes, buckets))),
                     ("gather_rows", lambda i: ops.gather_rows(ids[i], feat)),
                     ("gather_rows + rowdot", lambda i: ops.rowdot(users[i % 8], ops.gather_rows(ids[i], feat)))):
        print(json.dumps({"lib": os.environ.get("MI_LIB", "tree"), "case": name, "B": B, "us": round(timeit(fn, 40), 2)}), flush=True)
