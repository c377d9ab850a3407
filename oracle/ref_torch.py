"""The reference's torch-op sequences, restated for CPU tensors -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Used (a) by tests as a second, BLAS-ordered witness next to the canonical-order C oracle and
(b) by bench.py's `cpu_baseline` leg: the reference itself cannot travel to the GPU box, so "the
reference CPU inductive_embedder" timed there is this restatement of the very ops it executes
(index gather -> sgemm -> threshold -> sgemm -> divide), with torch's own CPU kernels.
Each function cites the reference lines it follows (R/ = RecBole/recbole/).
"""
import torch


def hash_points(planes, x):
    """R/inductive/torch_hash.py:55-60."""
    result = x @ planes.T
    neg = result < 0
    result[neg] = 0
    result[~neg] = 1
    return result


def lsh_embed(ids, feat, planes, buckets):
    """R/inductive/lsh_embedder.py:127-130,176-179."""
    bits = hash_points(planes, feat[ids])
    return (bits @ buckets) / bits.sum(dim=1).view(-1, 1)


def slsh_embed(ids, feat, planes, buckets):
    """R/inductive/single_lsh_embedder.py:82-87,103-109."""
    bits = hash_points(planes, feat[ids])
    node = (2 ** bits).sum(axis=1).long() % buckets.shape[0]
    return buckets[node], node


def rowdot(u, e):
    """R/model/general_recommender/bpr.py:145-149."""
    return torch.mul(u, e).sum(dim=1)


def full_sort(u, e):
    """R/model/general_recommender/bpr.py:158-163."""
    return torch.matmul(u, e.transpose(0, 1))


def knn_aggregate(idx, weight):
    """R/inductive/knn_embedder.py:125-126."""
    sel = weight[idx.ravel()]
    return torch.vstack([x.mean(dim=0) for x in sel.split(2)])
