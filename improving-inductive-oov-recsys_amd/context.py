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
    """Rows for the out-of-vocabulary ids of one side, in order of appearance."""
    oov = ids >= n_vocab
    oov_ids = ids[oov]  # fresh copy: embedders may strip prime_pad in place
    width = buckets_weight.shape[1] if buckets_weight is not None else None
    if oov_ids.numel() == 0:
        return oov, None
    if mapper is not None:
        mapped = mapper.map_user_ids(oov_ids) if side == "user" else mapper.map_item_ids(oov_ids)
        return oov, ops.gather_rows(mapped - n_vocab, buckets_weight)
    if embedder is not None:
        rows = embedder.embed_user_ids(oov_ids, model) if side == "user" else embedder.embed_item_ids(oov_ids, model)
        return oov, rows
    raise RuntimeError("Must provide either self.inductive_mapper or self.inductive_embedder")


def embed_token_fields(token_fields, table_weight, offsets, n_users, n_items, model, mapper=None, embedder=None,
                       user_buckets=None, item_buckets=None, sum_fields=False):
    """token_fields int64[B, nf] (column 0 = user id, column 1 = item id) -> float32[B, nf, D], or the
    first-order form float32[B, 1, D] summed over the fields when sum_fields=True.

    `model` is handed to the embedder exactly like the reference hands `self`: lsh/slsh read
    `model.user_oov_buckets` / `model.item_oov_buckets` from it."""
    if token_fields is None:
        return None
    tokens = C.dev_tensor(token_fields, torch.int64, "token_fields")
    table = C.dev_tensor(table_weight.detach(), torch.float32, "table_weight")
    B, nf = tokens.shape
    T, D = table.shape
    off = torch.as_tensor(offsets, dtype=torch.int64, device=tokens.device).contiguous()
    users, items = tokens[:, 0].contiguous(), tokens[:, 1].contiguous()
    oov_u, rows_u = _oov_rows(users, n_users, "user", model, mapper, embedder,
                              None if user_buckets is None else user_buckets.detach())
    oov_i, rows_i = _oov_rows(items, n_items, "item", model, mapper, embedder,
                              None if item_buckets is None else item_buckets.detach())
    rank_u = (torch.cumsum(oov_u, 0) - oov_u.to(torch.int64)).contiguous()
    rank_i = (torch.cumsum(oov_i, 0) - oov_i.to(torch.int64)).contiguous()
    if rows_u is not None:
        rows_u = C.dev_tensor(rows_u.detach().reshape(-1, D), torch.float32, "oov user rows")
    if rows_i is not None:
        rows_i = C.dev_tensor(rows_i.detach().reshape(-1, D), torch.float32, "oov item rows")
    out = torch.empty((B, D) if sum_fields else (B, nf, D), dtype=torch.float32, device=tokens.device)
    with C.on_device(tokens):
        rc = C.lib().mi_oov_token_fields_embed(
            C.ptr(tokens), B, nf, C.ptr(off), C.ptr(table), T, D, int(n_users), int(n_items),
            C.ptr(rows_u), C.ptr(rank_u), 0 if rows_u is None else rows_u.shape[0],
            C.ptr(rows_i), C.ptr(rank_i), 0 if rows_i is None else rows_i.shape[0],
            1 if sum_fields else 0, C.ptr(out), C.stream_of(tokens))
    C.check(rc, "mi_oov_token_fields_embed")
    return out.view(B, 1, D) if sum_fields else out
