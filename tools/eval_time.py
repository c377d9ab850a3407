#!/usr/bin/env python3
"""Developer tool: wall time of driver.evaluate (uni-N sampled ranking evaluation, BPR + lsh, 64-wide features) with one
model.predict call per reference batch (eval_rows_per_launch = 1: the round-2 loop) and with the batches queued (default).
    python3 tools/eval_time.py [n_interactions] [negatives]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mi_oov  # noqa: E402
from mi_oov import driver  # noqa: E402


class Cfg(dict):
    def __getitem__(self, k):
        return self.get(k, None)


class DS:
    def __init__(self, nu, ni):
        self.n = {"user_id": nu, "item_id": ni}

    def num(self, f):
        return self.n[f]


def main():
    n_inter = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    nneg = int(sys.argv[2]) if len(sys.argv) > 2 else 250
    dev = torch.device("cuda:0")
    n_users, n_items, new_u, new_i = 200_000, 2_000_000, 100_000, 1_000_000
    g = torch.Generator().manual_seed(0)
    fu = torch.randn((n_users + new_u, 64), generator=g)
    fi = torch.randn((n_items + new_i, 64), generator=g)
    ft_u = mi_oov.FeatureTable({"id": torch.arange(fu.shape[0]), "f": fu})
    ft_i = mi_oov.FeatureTable({"id": torch.arange(fi.shape[0]), "f": fi})
    cfg = Cfg(USER_ID_FIELD="user_id", ITEM_ID_FIELD="item_id", NEG_PREFIX="neg_", device=dev, embedding_size=64,
              add_oov_buckets=True, user_oov_buckets=8, item_oov_buckets=8, oov_freeze_embedding=False)
    lsh = mi_oov.LSHInductiveEmbedder(ft_u, ft_i, n_users, n_items, 8, 8, 64, dev, 112062759511, "none", mi_oov.InductiveFeatureCache())
    model = mi_oov.BPR(cfg, DS(n_users, n_items), None, lsh).to(dev).eval()
    gd = torch.Generator(device=dev).manual_seed(1)
    users = torch.randint(1, fu.shape[0], (n_inter,), generator=gd, device=dev)
    items = torch.randint(1, fi.shape[0], (n_inter,), generator=gd, device=dev)
    base = dict(driver.DEFAULTS, eval_negatives=nneg, metrics=None)
    out = {}
    for name, rpl in (("one predict per 1e5-row batch", 1), ("batches queued (4M rows per launch)", None),
                      ("one predict per 1e5-row batch", 1), ("batches queued (4M rows per launch)", None)):
        c = driver.Config(dict(base, eval_rows_per_launch=rpl))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = driver.evaluate(model, users, items, fi.shape[0], c, n_users, n_items, dev, torch.Generator(device=dev).manual_seed(9))
        torch.cuda.synchronize()
        out.setdefault(name, []).append(round(time.perf_counter() - t0, 4))
        out[name + " recall@10"] = res["overall"]["recall@10"]
    print(json.dumps({"interactions": n_inter, "negatives": nneg, "rows": n_inter * (1 + nneg), "seconds": out}))


if __name__ == "__main__":
    main()
