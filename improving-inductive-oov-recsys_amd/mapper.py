"""Host-side mirror of the reference's id -> OOV-bucket-id mappers, backed by libmi_oov.so.

    AbstractInductiveMapper     R/inductive/abstract_mapper.py:5-68
    RandomOOVInductiveMapper    R/inductive/random_mapper.py:37-130      'random'
"""
import torch
from torch import nn

from . import ops


class AbstractInductiveMapper(nn.Module):
    def __init__(self, user_features, item_features) -> None:
        super().__init__()
        self.user_features = user_features
        self.item_features = item_features
        self.n_new_users = len(user_features)
        self.n_new_items = len(item_features)
        self.training = False

    def set_train(self):
        self.training = True

    def set_eval(self):
        self.training = False

    def map_user_ids(self, user_ids: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError()

    def map_item_ids(self, item_ids: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError()

    def map_all_item_embeddings(self, item_embeddings):
        raise NotImplementedError()


class RandomOOVInductiveMapper(AbstractInductiveMapper):
    """ids < n_original pass through; the rest become hash(id - n_original) % n_buckets +
    n_original with hash in {'mod','fast','3round','64bit'} (random_mapper.py:104-130).  The whole
    map is one elementwise HIP launch (the reference needs ~10 torch kernels, and a NumPy round
    trip for '64bit')."""

    def __init__(self, user_features, item_features, n_original_users, n_original_items, n_user_oov_buckets,
                 n_item_oov_buckets, embedding_size, device, prime_pad, hash_function) -> None:
        super().__init__(user_features, item_features)
        self.n_original_users = n_original_users
        self.n_original_items = n_original_items
        self.n_user_oov_buckets = n_user_oov_buckets
        self.n_item_oov_buckets = n_item_oov_buckets
        self.embedding_size = embedding_size
        self.prime_pad = prime_pad
        self.hash_function = hash_function

    def set_train(self):
        # random_mapper.py:60-63
        super().set_train()
        self.n_new_users = self.n_original_users * 2
        self.n_new_items = self.n_original_items * 2

    def set_eval(self):
        # random_mapper.py:65-68
        super().set_eval()
        self.n_new_users = len(self.user_features)
        self.n_new_items = len(self.item_features)

    def _check(self):
        if self.hash_function not in ops.HASH_KINDS:
            raise ValueError(f"Unknown hash function {self.hash_function}")

    def _fast_int_hash(self, x):
        return ops.mapper_hash(x, "fast")

    def _three_round_int_hash(self, x):
        return ops.mapper_hash(x, "3round")

    def _big_64bit_hash(self, x, n_buckets):
        return ops.mapper_map(x, "64bit", 0, n_buckets)

    def _hash_ids(self, oov_ids, n_buckets):
        self._check()
        return ops.mapper_map(oov_ids, self.hash_function, 0, n_buckets)

    def map_user_ids(self, user_ids):
        self._check()
        return ops.mapper_map(user_ids, self.hash_function, self.n_original_users, self.n_user_oov_buckets)

    def map_item_ids(self, item_ids):
        self._check()
        return ops.mapper_map(item_ids, self.hash_function, self.n_original_items, self.n_item_oov_buckets)
