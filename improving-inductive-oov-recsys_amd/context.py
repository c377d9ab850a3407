"""OOV splice for the context-aware models (DCNV2 / WideDeep / xDeepFM), SURVEY.md section 8f rank 2.

The reference overrides two methods so that out-of-vocabulary user / item ids get plugin (or OOV
bucket) rows instead of fused-table rows:

    InductiveContextRecommender.embed_token_fields   R/model/abstract_recommender.py:794-842
    InductiveFMFirstOrderLinear.embed_token_fields   R/model/layers.py:1634-1693

The dense towers of those models (cross net, MLP, CIN) are ordinary torch modules and out of scope;
what is mirrored here is exactly the two splices, as functions that take the model-side tensors the
reference methods read (`token_embedding_table.embedding.weight`, `.offsets`, `n_users`, `n_items`,
the OOV bucket tables, the mapper / embedder) and return what they return.  One HIP launch gathers
all token fields and splices the OOV rows (mi_oov_token_fields_embed); the OOV rows themselves come
from the plugin (`embed_user_ids` / `embed_item_ids`) or from the bucket tables through the mapper,
as in the reference.  INTEGRATION.md shows the two-line change in the reference's methods.
"""
import torch

from . import _cabi as C
from . import ops


def _oov_rows(ids, n_vocab, side, model, mapper, embedder, buckets_weight):
    """Rows for the out-of-vocabulary ids of one side, in order of appearance (attached to the autograd graph when
    gradients are enabled: bucket tables for lsh / slsh / the mapper, MLP weights for dhe, the token table for knn)."""
    oov = ids >= n_vocab
    oov_ids = ids[oov]  # fresh copy: embedders may strip prime_pad in place
    if oov_ids.numel() == 0:
        return oov, None
    if mapper is not None:
        mapped = mapper.map_user_ids(oov_ids) if side == "user" else mapper.map_item_ids(oov_ids)
        return oov, ops.gather_rows(mapped - n_vocab, buckets_weight)
    if embedder is not None:
        rows = embedder.embed_user_ids(oov_ids, model) if side == "user" else embedder.embed_item_ids(oov_ids, model)
        return oov, rows
    raise RuntimeError("Must provide either self.inductive_mapper or self.inductive_embedder")


def _launch(tokens, off, table, n_users, n_items, rows_u, rank_u, rows_i, rank_i, sum_fields):
    B, nf = tokens.shape
    T, D = table.shape
    out = torch.empty((B, D) if sum_fields else (B, nf, D), dtype=torch.float32, device=tokens.device)
    with C.on_device(tokens):
        rc = C.lib().mi_oov_token_fields_embed(
            C.ptr(tokens), B, nf, C.ptr(off), C.ptr(table), T, D, int(n_users), int(n_items),
            C.ptr(rows_u), C.ptr(rank_u), 0 if rows_u is None else rows_u.shape[0],
            C.ptr(rows_i), C.ptr(rank_i), 0 if rows_i is None else rows_i.shape[0],
            1 if sum_fields else 0, C.ptr(out), C.stream_of(tokens))
    C.check(rc, "mi_oov_token_fields_embed")
    return out


class _TokenFieldsEmbed(torch.autograd.Function):
    """The splice under autograd.  The reference builds it from nn.Embedding + in-place index_put
    (abstract_recommender.py:815-839): a table row receives the gradient of every (b, field) that gathered it EXCEPT the
    positions an OOV row overwrote; an OOV row receives the gradient of the position it was written to; the first-order
    form sums over the fields, so every field sees the same gradient.  Both scatters are mi_oov_scatter_add_rows
    (indices of overwritten / in-vocabulary positions are -1, which the kernel skips: no boolean indexing, no sync)."""

    @staticmethod
    def forward(ctx, tokens, off, table, rows_u, rank_u, oov_u, rows_i, rank_i, oov_i, n_users, n_items, sum_fields):
        D = table.shape[1]
        ru = None if rows_u is None else C.dev_tensor(rows_u.detach().reshape(-1, D), torch.float32, "oov user rows")
        ri = None if rows_i is None else C.dev_tensor(rows_i.detach().reshape(-1, D), torch.float32, "oov item rows")
        out = _launch(tokens, off, C.dev_tensor(table.detach(), torch.float32, "table_weight"), n_users, n_items, ru, rank_u,
                      ri, rank_i, sum_fields)
        ctx.save_for_backward(tokens, off, rank_u, oov_u, rank_i, oov_i)
        ctx.meta = (table.shape, None if rows_u is None else rows_u.shape, None if rows_i is None else rows_i.shape,
                    sum_fields, table.requires_grad, rows_u is not None and rows_u.requires_grad,
                    rows_i is not None and rows_i.requires_grad)
        return out

    @staticmethod
    def backward(ctx, g):
        tokens, off, rank_u, oov_u, rank_i, oov_i = ctx.saved_tensors
        (T, D), shape_u, shape_i, sum_fields, need_t, need_u, need_i = ctx.meta
        B, nf = tokens.shape
        g = g.contiguous().view(B, 1 if sum_fields else nf, D)
        gt = gu = gi = None
        if need_t:
            idx = tokens + off[None, :]
            idx[:, 0] = torch.where(oov_u, torch.full_like(rank_u, -1), idx[:, 0])
            idx[:, 1] = torch.where(oov_i, torch.full_like(rank_i, -1), idx[:, 1])
            gt = ops.scatter_add_rows(idx.reshape(-1), g.expand(B, nf, D).reshape(-1, D), T)
        if need_u:
            gu = ops.scatter_add_rows(torch.where(oov_u, rank_u, torch.full_like(rank_u, -1)), g[:, 0].contiguous(),
                                      shape_u[0] if len(shape_u) == 2 else shape_u.numel() // D).view(shape_u)
        if need_i:
            col = 0 if sum_fields else 1
            gi = ops.scatter_add_rows(torch.where(oov_i, rank_i, torch.full_like(rank_i, -1)), g[:, col].contiguous(),
                                      shape_i[0] if len(shape_i) == 2 else shape_i.numel() // D).view(shape_i)
        return None, None, gt, gu, None, None, gi, None, None, None, None, None


def embed_token_fields(token_fields, table_weight, offsets, n_users, n_items, model, mapper=None, embedder=None,
                       user_buckets=None, item_buckets=None, sum_fields=False):
    """token_fields int64[B, nf] (column 0 = user id, column 1 = item id) -> float32[B, nf, D], or the
    first-order form float32[B, 1, D] summed over the fields when sum_fields=True.

    `model` is handed to the embedder exactly like the reference hands `self`: lsh/slsh read
    `model.user_oov_buckets` / `model.item_oov_buckets` from it.  With gradients enabled the result carries them to
    `table_weight`, to the OOV bucket tables (mapper) and to whatever the embedder's rows depend on, as the reference's
    in-place splice does (tests/test_context_splice.py: gradients pinned on the reference's autograd)."""
    if token_fields is None:
        return None
    tokens = C.dev_tensor(token_fields, torch.int64, "token_fields")
    if not isinstance(table_weight, torch.Tensor) or table_weight.dtype != torch.float32:
        raise TypeError("table_weight must be a float32 tensor")
    B, nf = tokens.shape
    D = table_weight.shape[1]
    off = torch.as_tensor(offsets, dtype=torch.int64, device=tokens.device).contiguous()
    users, items = tokens[:, 0].contiguous(), tokens[:, 1].contiguous()
    oov_u, rows_u = _oov_rows(users, n_users, "user", model, mapper, embedder, user_buckets)
    oov_i, rows_i = _oov_rows(items, n_items, "item", model, mapper, embedder, item_buckets)
    rank_u = (torch.cumsum(oov_u, 0) - oov_u.to(torch.int64)).contiguous()
    rank_i = (torch.cumsum(oov_i, 0) - oov_i.to(torch.int64)).contiguous()
    needs_grad = torch.is_grad_enabled() and (table_weight.requires_grad or (rows_u is not None and rows_u.requires_grad)
                                              or (rows_i is not None and rows_i.requires_grad))
    if needs_grad:
        out = _TokenFieldsEmbed.apply(tokens, off, table_weight, rows_u, rank_u, oov_u, rows_i, rank_i, oov_i, n_users,
                                      n_items, sum_fields)
    else:
        ru = None if rows_u is None else C.dev_tensor(rows_u.detach().reshape(-1, D), torch.float32, "oov user rows")
        ri = None if rows_i is None else C.dev_tensor(rows_i.detach().reshape(-1, D), torch.float32, "oov item rows")
        out = _launch(tokens, off, C.dev_tensor(table_weight.detach(), torch.float32, "table_weight"), n_users, n_items,
                      ru, rank_u, ri, rank_i, sum_fields)
    return out.view(B, 1, D) if sum_fields else out
