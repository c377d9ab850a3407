"""`torch.ops.mi_oov.*`: the hot entry points registered with torch's dispatcher ("host
Python calls hand-written HIP kernels through a thin C-ABI custom-op layer").

What a reference maintainer gains over calling `mi_oov.ops` directly: the ops show up in `torch.ops`, carry a schema,
dispatch on the device of their arguments -- the CUDA (ROCm) key runs the HIP kernel, every other key, CPU included,
raises (there is no CPU implementation: nothing here computes off the GPU) -- and have a fake-tensor rule (shapes /
dtypes without running), so graphs that contain them can be traced.  Gradients: registered for the ops whose outputs the
reference differentiates (lsh / slsh bucket tables), by the same backward kernels `mi_oov.ops` uses.

    import mi_oov.torch_ops            # registers
    emb = torch.ops.mi_oov.lsh_embed(ids, feature_mat, planes, bucket_weight)        # lsh_embedder.py:161-179
    s   = torch.ops.mi_oov.lsh_embed_score(ids, feature_mat, planes, bucket_weight, user_rows)
"""
import torch

from . import ops

_lib = torch.library.Library("mi_oov", "DEF")

_SCHEMAS = {
    "lsh_embed": "(Tensor ids, Tensor feat, Tensor planes, Tensor buckets) -> Tensor",
    "lsh_bits": "(Tensor ids, Tensor feat, Tensor planes) -> Tensor",
    "lsh_embed_score": "(Tensor ids, Tensor feat, Tensor planes, Tensor buckets, Tensor other) -> Tensor",
    "lsh_embed_score_multi": "(Tensor[] ids, Tensor feat, Tensor planes, Tensor buckets, Tensor[] other) -> Tensor[]",
    "slsh_embed": "(Tensor ids, Tensor feat, Tensor planes, Tensor buckets) -> Tensor",
    "slsh_index": "(Tensor ids, Tensor feat, Tensor planes, int n_buckets) -> Tensor",
    "mapper_map": "(Tensor ids, str kind, int n_original, int n_buckets) -> Tensor",
    "siphash24_mod": "(Tensor ids, Tensor keys) -> Tensor",
    "gather_mean": "(Tensor idx, Tensor weight, int group) -> Tensor",
    "rowdot": "(Tensor a, Tensor b) -> Tensor",
    "score_topk": "(Tensor users, Tensor items, int k, int n_skip_low) -> (Tensor, Tensor)",
}


def _no_cpu(name):
    def raise_(*args, **kwargs):
        raise RuntimeError(f"mi_oov::{name}: tensors are not on a ROCm device -- mi_oov kernels run on an MI355X only; "
                           "there is no CPU fallback")
    return raise_


_IMPL = {
    "lsh_embed": lambda ids, feat, planes, buckets: ops._lsh_forward(ids, feat, planes, buckets)[0],
    "lsh_bits": ops.lsh_bits,
    "lsh_embed_score": lambda ids, feat, planes, buckets, other: ops.lsh_embed_score(ids, feat, planes, buckets, other),
    "lsh_embed_score_multi": lambda ids, feat, planes, buckets, other: ops.lsh_embed_score_multi(ids, feat, planes, buckets, other),
    "slsh_embed": lambda ids, feat, planes, buckets: ops._slsh_forward(ids, feat, planes, buckets, buckets.shape[0])[0],
    "slsh_index": ops.slsh_index,
    "mapper_map": ops.mapper_map,
    "siphash24_mod": ops.siphash24_mod,
    "gather_mean": lambda idx, weight, group: ops.gather_mean(idx, weight, group),
    "rowdot": ops.rowdot,
    "score_topk": lambda users, items, k, n_skip_low: ops.score_topk(users, items, k, n_skip_low),
}


def _f32(rows, cols, like):
    return torch.empty((rows, cols), dtype=torch.float32, device=like.device)


_FAKE = {
    "lsh_embed": lambda ids, feat, planes, buckets: _f32(ids.numel(), buckets.shape[1], ids),
    "lsh_bits": lambda ids, feat, planes: torch.empty((ids.numel(), planes.shape[0]), dtype=torch.uint8, device=ids.device),
    "lsh_embed_score": lambda ids, feat, planes, buckets, other: torch.empty((ids.numel(),), dtype=torch.float32, device=ids.device),
    "lsh_embed_score_multi": lambda ids, feat, planes, buckets, other: [torch.empty((i.numel(),), dtype=torch.float32, device=i.device) for i in ids],
    "slsh_embed": lambda ids, feat, planes, buckets: _f32(ids.numel(), buckets.shape[1], ids),
    "slsh_index": lambda ids, feat, planes, n_buckets: torch.empty((ids.numel(),), dtype=torch.int64, device=ids.device),
    "mapper_map": lambda ids, kind, n_original, n_buckets: torch.empty_like(ids),
    "siphash24_mod": lambda ids, keys: _f32(ids.numel(), keys.shape[0], ids),
    "gather_mean": lambda idx, weight, group: _f32(idx.numel() // group, weight.shape[1], idx),
    "rowdot": lambda a, b: torch.empty((a.shape[0],), dtype=torch.float32, device=a.device),
    "score_topk": lambda users, items, k, n_skip_low: (_f32(users.shape[0], k, users),
                                                       torch.empty((users.shape[0], k), dtype=torch.int64, device=users.device)),
}

for _name, _schema in _SCHEMAS.items():
    _lib.define(_name + _schema)
    _lib.impl(_name, _IMPL[_name], "CUDA")
    _lib.impl(_name, _no_cpu(_name), "CPU")
    torch.library.register_fake("mi_oov::" + _name, _FAKE[_name], lib=_lib)


# gradients of the two ops whose outputs the reference differentiates with respect to a bucket table
def _lsh_setup(ctx, inputs, output):
    ids, feat, planes, buckets = inputs
    ctx.save_for_backward(ops.lsh_bits(ids, feat, planes))


def _lsh_backward(ctx, grad):
    (bits,) = ctx.saved_tensors
    return None, None, None, ops.lsh_embed_backward(bits, grad.contiguous())


def _slsh_setup(ctx, inputs, output):
    ids, feat, planes, buckets = inputs
    ctx.save_for_backward(ops.slsh_index(ids, feat, planes, buckets.shape[0]))
    ctx.n_buckets = buckets.shape[0]


def _slsh_backward(ctx, grad):
    (idx,) = ctx.saved_tensors
    return None, None, None, ops.slsh_embed_backward(idx, grad.contiguous(), ctx.n_buckets)


torch.library.register_autograd("mi_oov::lsh_embed", _lsh_backward, setup_context=_lsh_setup, lib=_lib)
torch.library.register_autograd("mi_oov::slsh_embed", _slsh_backward, setup_context=_slsh_setup, lib=_lib)

OPS = tuple(_SCHEMAS)
