#!/usr/bin/env python3
"""Every plugin class on odd but legal configurations (what `--embedding_size`, `--*_oov_buckets`, `--dhe_num_hashes`,
`--dhe_layer_size`, `--oov_knn_num_neighbors` and a dataset's feature columns can be): eval and train forward, a backward
pass, against a float64 torch restatement where one is cheap.  Developer probe, GPU box: prints ok / FAIL per case."""
import itertools
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import mi_oov as mi  # noqa: E402

dev = torch.device("cuda:0")
PRIME_PAD = 112062759511
os.chdir(tempfile.mkdtemp())


class M(torch.nn.Module):
    def __init__(self, n_u, n_i, nbu, nbi, D):
        super().__init__()
        self.user_embedding = torch.nn.Embedding(n_u, D)
        self.item_embedding = torch.nn.Embedding(n_i, D)
        self.user_oov_buckets = torch.nn.Embedding(nbu, D)
        self.item_oov_buckets = torch.nn.Embedding(nbi, D)


def run(name, make, D, nbu, nbi, n_new=300, n_orig=200):
    try:
        g = torch.Generator().manual_seed(1)
        emb = make()
        model = M(n_orig, n_orig, nbu, nbi, D).to(dev)
        ids_u = torch.randint(0, n_new, (257,), generator=g).to(dev)
        ids_i = torch.randint(0, n_new, (129,), generator=g).to(dev)
        emb.set_eval()
        with torch.no_grad():
            eu, ei = emb.embed_user_ids(ids_u.clone(), model), emb.embed_item_ids(ids_i.clone(), model)
        k = getattr(emb, "n_neighbors", 2)  # knn: `.split(2)` + mean of the B k gathered rows (knn_embedder.py:125-126): B k / 2 rows
        assert eu.shape == ((257 * k + 1) // 2, D) and ei.shape == ((129 * k + 1) // 2, D), (eu.shape, ei.shape)
        emb.set_train()
        tu = emb.embed_user_ids(ids_u.clone() + PRIME_PAD * (torch.arange(257, device=dev) % 2), model)
        ti = emb.embed_item_ids(ids_i.clone(), model)
        loss = torch.nan_to_num(tu).sum() + torch.nan_to_num(ti).sum()
        if loss.requires_grad:
            loss.backward()
        same_u = torch.equal(torch.nan_to_num(tu.detach(), 7.0), torch.nan_to_num(eu, 7.0))
        print(f"ok   {name}" + ("" if same_u else "   (train-mode rows differ from eval rows)"), flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"FAIL {name}: {type(e).__name__}: {str(e)[:160]}", flush=True)


def feats(n, widths, seed):
    g = torch.Generator().manual_seed(seed)
    cols = {"id": torch.arange(n)}
    for j, w in enumerate(widths):
        cols[f"f{j}"] = torch.randn((n,) if w == 0 else (n, w), generator=g)
    return mi.FeatureTable(cols)


n_new, n_orig = 300, 200
for widths, D, nb in itertools.product(((0,), (0, 3, 17), (70,), (300,)), (64, 50, 1, 300), (1, 2, 8, 300)):
    fu, fi = feats(n_new, widths, 1), feats(n_new, widths[::-1], 2)
    tag = f"F={sum(max(1, w) for w in widths)} D={D} buckets={nb}"
    for norm in ("per-feature", "global", "none"):
        run(f"lsh  {tag} {norm}", lambda: mi.LSHInductiveEmbedder(fu, fi, n_orig, n_orig, nb, nb, D, dev, PRIME_PAD, norm, mi.InductiveFeatureCache()), D, nb, nb)
    run(f"slsh {tag}", lambda: mi.SingleLSHInductiveEmbedder(fu, fi, n_orig, n_orig, nb, nb, D, dev, PRIME_PAD, "per-feature"), D, nb, nb)
for widths, D in itertools.product(((0,), (0, 3, 17), (300,)), (64, 50, 1)):
    fu, fi = feats(n_new, widths, 1), feats(n_new, widths[::-1], 2)
    tag = f"F={sum(max(1, w) for w in widths)} D={D}"
    for k in (1, 7, 1024):
        run(f"dhe  {tag} hashes={k}", lambda: mi.DeepHashEmbedder(fu, fi, n_orig, n_orig, 8, 8, D, dev, PRIME_PAD, num_hashes=k), D, 8, 8)
    for k, L in ((1, 1), (16, 33), (7, 2048)):
        run(f"fdhe {tag} hashes={k} layer={L}", lambda: mi.FeatDeepHashEmbedder(fu, fi, n_orig, n_orig, 8, 8, D, dev, PRIME_PAD, num_hashes=k, dhe_layer_size=L), D, 8, 8)
    for L in (1, 33, 2048):
        run(f"dnn  {tag} layer={L}", lambda: mi.DNNEmbedder(fu, fi, n_orig, n_orig, 8, 8, D, dev, PRIME_PAD, dhe_layer_size=L), D, 8, 8)
    for k in (1, 2, 3, 10, 199):
        run(f"knn  {tag} neighbours={k}", lambda: mi.KNNInductiveEmbedder(fu, fi, n_orig, n_orig, 8, 8, D, dev, PRIME_PAD, n_neighbors=k), D, 8, 8)
    run(f"mean {tag}", lambda: mi.MeanEmbedder(fu, fi, n_orig, n_orig, 8, 8, D, dev), D, 8, 8)
    run(f"zero {tag}", lambda: mi.ZeroEmbedder(fu, fi, n_orig, n_orig, D, dev), D, 8, 8)
