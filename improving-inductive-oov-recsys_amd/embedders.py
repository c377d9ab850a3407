"""Host-side mirror of the reference's inductive-embedder plugin surface, backed by libmi_oov.so.

Same class names, constructor keywords, attribute names and error behaviour as the reference
(R/ = RecBole/recbole/ in snap-research/improving-inductive-oov-recsys):

    AbstractInductiveEmbedder   R/inductive/abstract_embedder.py:5-70
    TorchLSHash                 R/inductive/torch_hash.py:10-66      (parameter `uniform_planes.0`)
    InductiveFeatureCache       R/inductive/feature_cache.py:1-22
    LSHInductiveEmbedder        R/inductive/lsh_embedder.py:11-192         'lsh'
    SingleLSHInductiveEmbedder  R/inductive/single_lsh_embedder.py:9-115   'slsh'
    DeepHashEmbedder            R/inductive/dh_embedder.py:18-259          'dhe'
    FeatDeepHashEmbedder        R/inductive/feat_dh_embedder.py:86-210     'fdhe'
    DNNEmbedder                 R/inductive/dnn_embedder.py:8-112          'dnn'
    KNNInductiveEmbedder        R/inductive/knn_embedder.py:18-150         'knn'
    MeanEmbedder                R/inductive/mean_embedder.py:12-87         'mean'
    ZeroEmbedder                R/inductive/zero_embedder.py:6-60          'zero'

The per-batch work of every embed_*_ids goes through the HIP kernels (ops.py); torch is used to
hold parameters/feature matrices in HBM and for the one-off constructor-time feature build.
`state_dict()` keys are those of the reference (a reference checkpoint re-exported as tensors only,
`{'state_dict': ...}`, loads as it is; the reference's own file also pickles its Config and optimizer, which the
safe loader used here refuses: driver.py).
"""
import json
import os
import secrets

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import ops


class FeatureTable:
    """Minimal stand-in for the RecBole `Interaction` the reference passes as user/item features
    (R/data/interaction.py): ordered named columns of equal length, column 0 being the id.
    Anything with `.columns`, `[name] -> Tensor` and `len() -> rows` works; this class lets the
    package be driven without RecBole's data layer (which is out of scope)."""

    def __init__(self, columns):
        self._cols = dict(columns)
        lens = {int(v.shape[0]) for v in self._cols.values()}
        if len(lens) > 1:
            raise ValueError(f"columns have different lengths: {sorted(lens)}")
        self._len = lens.pop() if lens else 0

    @property
    def columns(self):
        return list(self._cols.keys())

    def __getitem__(self, name):
        return self._cols[name]

    def __len__(self):
        return self._len


class AbstractInductiveEmbedder(nn.Module):
    """abstract_embedder.py:5-70: ctor bookkeeping, train/eval toggles, abstract embed methods."""

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

    def embed_user_ids(self, user_ids, model):
        raise NotImplementedError()

    def embed_item_ids(self, item_ids, model):
        raise NotImplementedError()

    def map_all_item_embeddings(self, item_embeddings):
        raise NotImplementedError()

    def embed_all_items(self, item_embeddings, model):
        raise NotImplementedError()


class InductiveFeatureCache:
    """feature_cache.py:1-22: lets the main and the first-order lsh embedder share matrices."""

    def __init__(self, mode="transductive"):
        self._user_feats = None
        self._item_feats = None
        self.mode = mode

    def get_mode(self):
        return self.mode

    def has_cached(self):
        return self._user_feats is not None and self._item_feats is not None

    def get_cached(self):
        return self._user_feats, self._item_feats

    def add_to_cache(self, user_feats, item_feats):
        self._user_feats = user_feats
        self._item_feats = item_feats


class TorchLSHash(nn.Module):
    """Random hyperplanes as an nn.ParameterList named `uniform_planes` (torch_hash.py:40-42), so
    the checkpoint key `<prefix>.uniform_planes.0` matches.  hash_points runs on the HIP kernel via
    the owning embedder; here it is offered for already-gathered rows."""

    def __init__(self, hash_size, input_dim, num_hashtables=1, storage_instance=None, device="cpu"):
        super().__init__()
        self.hash_size = hash_size
        self.input_dim = input_dim
        self.num_hashtables = num_hashtables
        self.storage_instance = storage_instance
        self.device = device
        self.uniform_planes = nn.ParameterList([
            nn.Parameter(torch.randn(self.hash_size, self.input_dim, device=self.device))
            for _ in range(self.num_hashtables)])

    def hash_points(self, planes, input_points):
        """f32 0/1 codes of rows that are already gathered (torch_hash.py:55-60)."""
        ids = torch.arange(input_points.shape[0], device=input_points.device)
        return ops.lsh_bits(ids, input_points, planes.data).to(torch.float32)


# ----------------------------------------------------------------------------------------------
# constructor-time feature matrices (lsh_embedder.py:77-106 and the copies in the other classes)
# ----------------------------------------------------------------------------------------------
def _columns(features):
    cols = features.columns if hasattr(features, "columns") else list(features.keys())
    return list(cols)[1:]  # column 0 is the id (lsh_embedder.py:77-78)


def build_feature_matrix(features, n_rows, per_feature, device):
    """hstack of every non-id column as float [n,-1]; per_feature=True L2-normalises each column
    block first, so a scalar column becomes +-1/0 (lsh_embedder.py:83-90)."""
    blocks = []
    for c in _columns(features):
        col = features[c].float().view(n_rows, -1)
        blocks.append(F.normalize(col, dim=-1) if per_feature else col)
    return torch.hstack(blocks).to(device)


def _strip_prime_pad_(ids, prime_pad):
    """In-place `ids[ids >= pad] -= pad` exactly as lsh_embedder.py:153-155 (the caller's tensor is
    a fresh masked copy, and the reference mutates it)."""
    mask = ids >= prime_pad
    ids[mask] = ids[mask] - prime_pad
    return ids


def _general_tables(model):
    """Which weight matrices an embedder reads from the model (knn_embedder.py:117-123,135-144;
    mean_embedder.py:53-60,75-86).  The reference switches on isinstance of its own model classes;
    the mirror keys on the attributes those classes define so that both the reference's models
    and this package's BPR are accepted.  Anything else raises ValueError like the reference."""
    if hasattr(model, "user_embedding") and hasattr(model, "item_embedding"):
        return model.user_embedding.weight, model.item_embedding.weight
    if hasattr(model, "token_embedding_table") and hasattr(model, "token_field_offsets"):
        w = model.token_embedding_table.embedding.weight
        off = model.token_field_offsets
        user_w = w[off[0]:off[1]]
        item_w = w[off[1]:] if len(off) == 2 else w[off[1]:off[2]]
        return user_w, item_w
    raise ValueError("Unknown model type")


_HOT_F = 64  # feature-row width of the hot kernels (csrc/lsh64.hip, lsh64p.hip): a 256-byte row per 16-lane group
_PAD_FEATURES = os.environ.get("MI_OOV_PAD_FEATURES", "1") != "0"
_PAD_MAX_BYTES = int(float(os.environ.get("MI_OOV_PAD_FEATURES_MAX_GIB", "32")) * (1 << 30))


class _FeatureEmbedder(AbstractInductiveEmbedder):
    """Shared ctor fields of the feature-driven embedders."""

    def hot_operands(self, side):
        """(feature matrix, hyperplanes) of `side` ("user" / "item") as the lsh / slsh kernels are to be given them.

        The hot kernels are laid out for 64-float feature rows; a narrower matrix -- the hstack of a real dataset's
        non-id columns (lsh_embedder.py:77-106) is a few dozen floats wide -- takes the generic kernels, which run 1.5-2x
        slower although they move fewer bytes (2 M lookups on a 10 M-row table, F = 20: 334 us against 188 us padded;
        tools/fpad_probe.py).  So a matrix narrower than 64 columns is kept ZERO-PADDED to 64 beside the original, and
        the hyperplanes are padded to match.  The results are the same bits: in the canonical summation order a lane
        owns elements e with (e / 4) % 16 == lane, so the padding lives in lanes and chain positions that add +0 -- a
        projection can at most turn from -0 to +0, which `x < 0` does not see (tests/test_gpu_plugin.py).  The public
        attributes user_feature_mat / item_feature_mat and uniform_planes keep the reference's shapes.
        MI_OOV_PAD_FEATURES=0 turns this off; matrices whose padded copy would exceed MI_OOV_PAD_FEATURES_MAX_GIB (32)
        stay as they are."""
        user = side == "user"
        feat = self.user_feature_mat if user else self.item_feature_mat
        planes = (self.user_lsh if user else self.item_lsh).uniform_planes[0].data
        width = feat.size(1)
        if (not _PAD_FEATURES or width >= _HOT_F or planes.size(0) > self._hot_planes() or not self._hot_dims()
                or feat.size(0) * _HOT_F * 4 > _PAD_MAX_BYTES or not feat.is_cuda):
            return feat, planes
        cache = self.__dict__.setdefault("_hot", {})
        ent = cache.get(side)
        if ent is None or ent[0] is not feat or ent[1] != feat._version:
            ent = cache[side] = [feat, feat._version, F.pad(feat, (0, _HOT_F - width)).contiguous(), None, -1, None]
        if ent[3] is not planes or ent[4] != planes._version:
            ent[3], ent[4], ent[5] = planes, planes._version, F.pad(planes, (0, _HOT_F - width)).contiguous()
        return ent[2], ent[5]

    def _hot_dims(self):
        return False  # which embedding widths have a hot tile: per embedder

    def _hot_planes(self):
        return 32  # most hyperplanes the 64-wide kernels take (slsh64_kernel: 32; lsh64g_kernel: 64)

    def _operands_of(self, lsh, feature_mat):
        """hot_operands for the reference's private `_hash_node(nodes, lsh, feature_mat)` signature."""
        if lsh is self.user_lsh and feature_mat is self.user_feature_mat:
            return self.hot_operands("user")
        if lsh is self.item_lsh and feature_mat is self.item_feature_mat:
            return self.hot_operands("item")
        return feature_mat, lsh.uniform_planes[0].data

    def _common(self, n_original_users, n_original_items, n_user_oov_buckets, n_item_oov_buckets, embedding_size,
                device, prime_pad):
        self.n_original_users = n_original_users
        self.n_original_items = n_original_items
        self.n_user_oov_buckets = n_user_oov_buckets
        self.n_item_oov_buckets = n_item_oov_buckets
        self.embedding_size = embedding_size
        self.device = device
        self.prime_pad = prime_pad


class LSHInductiveEmbedder(_FeatureEmbedder):
    """'lsh': mean of the OOV-bucket rows selected by n_buckets sign-random-projections."""

    def __init__(self, user_features, item_features, n_original_users, n_original_items, n_user_oov_buckets,
                 n_item_oov_buckets, embedding_size, device, prime_pad, normalization_type, feature_cache) -> None:
        super().__init__(user_features, item_features)
        self._common(n_original_users, n_original_items, n_user_oov_buckets, n_item_oov_buckets, embedding_size,
                     device, prime_pad)
        if feature_cache.has_cached():
            self.user_feature_mat, self.item_feature_mat = feature_cache.get_cached()
        else:
            if normalization_type == "per-feature":
                per = True
            elif normalization_type in ("global", "none"):
                per = False
            else:
                raise ValueError(f"Invalid normalization type: {normalization_type}")
            self.user_feature_mat = build_feature_matrix(user_features, self.n_new_users, per, device)
            self.item_feature_mat = build_feature_matrix(item_features, self.n_new_items, per, device)
            if normalization_type == "global":
                self.user_feature_mat = F.normalize(self.user_feature_mat, dim=-1)
                self.item_feature_mat = F.normalize(self.item_feature_mat, dim=-1)
            feature_cache.add_to_cache(self.user_feature_mat, self.item_feature_mat)
        # one hyperplane per OOV bucket (lsh_embedder.py:108-114)
        self.user_lsh = TorchLSHash(hash_size=n_user_oov_buckets, input_dim=self.user_feature_mat.size(1),
                                    device=device)
        self.item_lsh = TorchLSHash(hash_size=n_item_oov_buckets, input_dim=self.item_feature_mat.size(1),
                                    device=device)

    def _hot_dims(self):
        return self.embedding_size == 64  # lsh64 / lsh64g / the persistent kernel: F = D = 64

    def _hot_planes(self):
        return 64

    def _hash_node(self, nodes, lsh, feature_mat):
        assert lsh.uniform_planes is not None
        feat, planes = self._operands_of(lsh, feature_mat)
        return ops.lsh_bits(nodes, feat, planes).to(torch.float32)

    def _hash_users(self, users):
        return self._hash_node(users, self.user_lsh, self.user_feature_mat)

    def _hash_items(self, items):
        return self._hash_node(items, self.item_lsh, self.item_feature_mat)

    def embed_user_ids(self, user_ids, model):
        if self.training:
            _strip_prime_pad_(user_ids, self.prime_pad)
        feat, planes = self.hot_operands("user")
        return ops.lsh_embed(user_ids, feat, planes, model.user_oov_buckets.weight)

    def embed_item_ids(self, item_ids, model):
        if self.training:
            _strip_prime_pad_(item_ids, self.prime_pad)
        feat, planes = self.hot_operands("item")
        return ops.lsh_embed(item_ids, feat, planes, model.item_oov_buckets.weight)

    # ---- K queued batches in one persistent launch (csrc/lsh64p.hip; not part of the reference surface) ----------
    def lsh_table(self, side, model):
        """The prepared 2^H-row table of aggregates of one side's bucket table (ops.LshTable: made once, re-made when the
        bucket Parameter has been written to), or None for shapes the persistent kernel does not take."""
        buckets = (model.user_oov_buckets if side == "user" else model.item_oov_buckets).weight
        cache = self.__dict__.setdefault("_lsh_tables", {})
        ent = cache.get(side)
        if ent is None or ent.buckets is not buckets:
            ent = cache[side] = ops.LshTable(buckets)
        return ent

    def _embed_ids_multi(self, side, ids_list, model, out=None):
        user = side == "user"
        if self.training:
            for ids in ids_list:
                _strip_prime_pad_(ids, self.prime_pad)
        feat, planes = self.hot_operands(side)
        buckets = (model.user_oov_buckets if user else model.item_oov_buckets).weight
        if torch.is_grad_enabled() and buckets.requires_grad:  # training: the per-batch autograd path
            return [ops.lsh_embed(ids, feat, planes, buckets) for ids in ids_list]
        return ops.lsh_embed_multi(ids_list, feat, planes, buckets, out=out, table=self.lsh_table(side, model))

    def embed_user_ids_multi(self, user_ids_list, model, out=None):
        """[embed_user_ids(ids, model) for ids in user_ids_list] (lsh_embedder.py:141-159), equal-sized batches, one launch."""
        return self._embed_ids_multi("user", user_ids_list, model, out)

    def embed_item_ids_multi(self, item_ids_list, model, out=None):
        """[embed_item_ids(ids, model) for ids in item_ids_list] (lsh_embedder.py:161-179), equal-sized batches, one launch."""
        return self._embed_ids_multi("item", item_ids_list, model, out)

    def __deepcopy__(self, memo):
        # prepared tables are derived state (get_flops deep-copies the model): re-made on first use by the copy
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in ("_lsh_tables", "_hot"):
                continue
            setattr(new, k, copy.deepcopy(v, memo))
        return new

    # fused entry points used by this package's BPR (not part of the reference surface)
    def score_item_ids(self, item_ids, model, user_e):
        if self.training:
            _strip_prime_pad_(item_ids, self.prime_pad)
        feat, planes = self.hot_operands("item")
        return ops.lsh_embed_score(item_ids, feat, planes, model.item_oov_buckets.weight, user_e)


class SingleLSHInductiveEmbedder(_FeatureEmbedder):
    """'slsh': one bucket row, index (bits_req + popcount) % n_buckets (single_lsh_embedder.py:86)."""

    def __init__(self, user_features, item_features, n_original_users, n_original_items, n_user_oov_buckets,
                 n_item_oov_buckets, embedding_size, device, prime_pad, normalization_type) -> None:
        super().__init__(user_features, item_features)
        self._common(n_original_users, n_original_items, n_user_oov_buckets, n_item_oov_buckets, embedding_size,
                     device, prime_pad)
        if normalization_type == "per-feature":
            per = True
        elif normalization_type in ("global", "none"):
            per = False  # 'global' is NOT post-normalised here, unlike lsh (single_lsh_embedder.py:66-75)
        else:
            raise ValueError(f"Invalid normalization type: {normalization_type}")
        self.user_feature_mat = build_feature_matrix(user_features, self.n_new_users, per, device)
        self.item_feature_mat = build_feature_matrix(item_features, self.n_new_items, per, device)
        self.user_bits_req = int(np.ceil(np.log2(self.n_user_oov_buckets)))
        self.item_bits_req = int(np.ceil(np.log2(self.n_item_oov_buckets)))
        self.user_lsh = TorchLSHash(hash_size=self.user_bits_req, input_dim=self.user_feature_mat.size(1),
                                    device=device)
        self.item_lsh = TorchLSHash(hash_size=self.item_bits_req, input_dim=self.item_feature_mat.size(1),
                                    device=device)

    def _hot_dims(self):
        return self.embedding_size in (64, 128)  # slsh64_kernel: F = 64, D = 64 / 128

    def _hash_node(self, nodes, lsh, feature_mat, n_buckets):
        assert lsh.uniform_planes is not None
        feat, planes = self._operands_of(lsh, feature_mat)
        return ops.slsh_index(nodes, feat, planes, n_buckets)

    def _hash_users(self, users):
        return self._hash_node(users, self.user_lsh, self.user_feature_mat, self.n_user_oov_buckets)

    def _hash_items(self, items):
        return self._hash_node(items, self.item_lsh, self.item_feature_mat, self.n_item_oov_buckets)

    def embed_user_ids(self, user_ids, model):
        if self.training:
            _strip_prime_pad_(user_ids, self.prime_pad)
        feat, planes = self.hot_operands("user")
        return ops.slsh_embed(user_ids, feat, planes, model.user_oov_buckets.weight)

    def embed_item_ids(self, item_ids, model):
        if self.training:
            _strip_prime_pad_(item_ids, self.prime_pad)
        feat, planes = self.hot_operands("item")
        return ops.slsh_embed(item_ids, feat, planes, model.item_oov_buckets.weight)


def _hash_mlp(in_features, hidden, out_features, device):
    """Linear-GELU x3, Linear, Sigmoid (dh_embedder.py:70-89); key names *.0/2/4/6.{weight,bias}."""
    return nn.Sequential(nn.Linear(in_features, hidden), nn.GELU(), nn.Linear(hidden, hidden), nn.GELU(),
                         nn.Linear(hidden, hidden), nn.GELU(), nn.Linear(hidden, out_features),
                         nn.Sigmoid()).to(device)


def _needs_grad(net, x):
    return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in net.parameters()))


def _run_hash_net(net, x):
    """The reference's `*_hash_net(x)`.  Without autograd the Linear+GELU / Linear+Sigmoid pairs run on
    mi_oov_linear_x3 (bf16 matrix cores at f32 accuracy, activation fused in the epilogue); when a gradient is required
    the same GEMM runs the forward unfused (pre-activations kept) and the backward (`ops.hash_net_train`): torch
    autograd only links the pieces, it computes nothing."""
    if _needs_grad(net, x):
        return ops.hash_net_train(net, x)
    return ops.hash_net_forward(net, x)


class _HashKeyMixin:
    HASH_KEY_PATH = "./hash_keys"
    MAX_HASH = 16777216

    def get_hash_keys(self):
        """./hash_keys/{K}.hashes relative to the CWD: hex JSON, created with secrets.token_bytes
        when absent (dh_embedder.py:95-120)."""
        os.makedirs(self.HASH_KEY_PATH, exist_ok=True)
        file_path = os.path.join(self.HASH_KEY_PATH, f"{self.num_hashes}.hashes")
        if os.path.exists(file_path):
            with open(file_path) as f:
                keys = json.load(f)
                assert len(keys) == self.num_hashes
                return [bytes.fromhex(x) for x in keys]
        keys = [secrets.token_bytes(16) for _ in range(self.num_hashes)]
        with open(file_path, "w") as f:
            json.dump([x.hex() for x in keys], f)
        return keys

    def _key_tensor(self, device):
        kt = getattr(self, "_keys_dev", None)
        if kt is None or kt.device != torch.device(device) or getattr(self, "_keys_src", None) is not self.hash_keys:
            flat = np.frombuffer(b"".join(self.hash_keys), dtype=np.uint8).reshape(-1, 16).copy()
            kt = torch.from_numpy(flat).to(device)
            self._keys_dev, self._keys_src = kt, self.hash_keys
        return kt

    def _hash_ids(self, ids):
        """[B, K] float hashes; one HIP launch instead of B*K csiphash calls (dh_embedder.py:154-170)."""
        return ops.siphash24_mod(ids, self._key_tensor(ids.device), self.MAX_HASH)

    def _hash_then_net(self, ids, net, extra=None):
        """net(hstack(hashes(ids), extra rows)) -- the dhe / fdhe forward (dh_embedder.py:191-217, feat_dh_embedder.py:164-172).
        One SipHash launch, then the layers.  (Cutting the batch in chunks and hashing chunk c + 1 on a side stream under
        the layers of chunk c was built and measured in round 3: 1.71 ms serial against 1.94 / 2.07 / 2.16 / 3.04 ms with
        2 / 3 / 4 / 8 chunks at K = 1024, 65536 lookups -- the f32 matrix instruction issues at the vector rate, so the
        integer hash and the GEMM compete for the same issue slots, and the smaller GEMMs fill the chip worse.  Again with
        the split-bf16 layers: 1.18 ms serial, 1.26 / 1.42 with 2 / 4 chunks -- the pipelined layer kernel's two waves per
        SIMD hold 496 of its 512 registers, so a hash wave only runs where a layer workgroup has left.)"""
        if extra is not None and not _needs_grad(net, extra) and ops._x3_wanted():
            # fdhe, inference: the hashes go straight into the net's input (rows of a multiple of 16 floats: what the
            # pipelined layer kernel takes), the feature columns beside them, zeros behind -- no concatenation and no
            # padding copy of the [B, K + F] matrix afterwards (two 268 MB copies at K = 1024, 65536 lookups)
            K, F = self._key_tensor(ids.device).shape[0], extra.shape[1]
            x = torch.empty((ids.numel(), -(-(K + F) // 16) * 16), dtype=torch.float32, device=ids.device)
            ops.siphash24_mod(ids, self._key_tensor(ids.device), self.MAX_HASH, out=x)
            x[:, K:K + F] = extra
            x[:, K + F:] = 0.0
            return ops.hash_net_forward(net, x)
        h = self._hash_ids(ids)
        return _run_hash_net(net, h if extra is None else torch.hstack((h, extra)))

    def __deepcopy__(self, memo):
        # device key cache is derived state: drop it so copies (get_flops deep-copies the model) stay light
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in ("_keys_dev", "_keys_src"):
                continue
            setattr(new, k, copy.deepcopy(v, memo))
        return new


class DeepHashEmbedder(_HashKeyMixin, _FeatureEmbedder):
    """'dhe': K SipHash-2-4 values of the raw id -> MLP(K,512,512,512,D) -> sigmoid.
    Hidden width is fixed at 512 and prime_pad is NOT stripped (dh_embedder.py:70-89,219-245)."""

    def __init__(self, user_features, item_features, n_original_users, n_original_items, n_user_oov_buckets,
                 n_item_oov_buckets, embedding_size, device, prime_pad, num_hashes) -> None:
        super().__init__(user_features, item_features)
        self._common(n_original_users, n_original_items, n_user_oov_buckets, n_item_oov_buckets, embedding_size,
                     device, prime_pad)
        self.num_hashes = num_hashes
        self.user_hash_net = _hash_mlp(num_hashes, 512, embedding_size, device)
        self.item_hash_net = _hash_mlp(num_hashes, 512, embedding_size, device)
        self.user_feature_mat = build_feature_matrix(user_features, self.n_new_users, True, device)
        self.item_feature_mat = build_feature_matrix(item_features, self.n_new_items, True, device)
        self.hash_keys = self.get_hash_keys()

    def _hash_users(self, users):
        return self._hash_then_net(users, self.user_hash_net)

    def _hash_items(self, items):
        return self._hash_then_net(items, self.item_hash_net)

    def embed_user_ids(self, user_ids, model):
        return self._hash_users(user_ids)

    def embed_item_ids(self, item_ids, model):
        return self._hash_items(item_ids)


class FeatDeepHashEmbedder(_HashKeyMixin, _FeatureEmbedder):
    """'fdhe': hashes of the UN-stripped id concatenated with the feature row of the stripped id
    (feat_dh_embedder.py:180-206)."""

    def __init__(self, user_features, item_features, n_original_users, n_original_items, n_user_oov_buckets,
                 n_item_oov_buckets, embedding_size, device, prime_pad, num_hashes, dhe_layer_size) -> None:
        super().__init__(user_features, item_features)
        self._common(n_original_users, n_original_items, n_user_oov_buckets, n_item_oov_buckets, embedding_size,
                     device, prime_pad)
        self.num_hashes = num_hashes
        self.user_feature_mat = build_feature_matrix(user_features, self.n_new_users, True, device)
        self.item_feature_mat = build_feature_matrix(item_features, self.n_new_items, True, device)
        self.user_hash_net = _hash_mlp(num_hashes + self.user_feature_mat.size(1), dhe_layer_size, embedding_size,
                                       device)
        self.item_hash_net = _hash_mlp(num_hashes + self.item_feature_mat.size(1), dhe_layer_size, embedding_size,
                                       device)
        self.hash_keys = self.get_hash_keys()

    def _lookup(self, old_ids):
        if self.training:
            return _strip_prime_pad_(old_ids.clone(), self.prime_pad)
        return old_ids

    def _hash_users(self, users, feat_lookup_users):
        return self._hash_then_net(users, self.user_hash_net, ops.gather_rows(feat_lookup_users, self.user_feature_mat))

    def _hash_items(self, items, feat_lookup_items):
        return self._hash_then_net(items, self.item_hash_net, ops.gather_rows(feat_lookup_items, self.item_feature_mat))

    def embed_user_ids(self, old_user_ids, model):
        return self._hash_users(old_user_ids, self._lookup(old_user_ids))

    def embed_item_ids(self, old_item_ids, model):
        return self._hash_items(old_item_ids, self._lookup(old_item_ids))


class DNNEmbedder(_FeatureEmbedder):
    """'dnn': feature row -> MLP (dnn_embedder.py:65-109)."""

    def __init__(self, user_features, item_features, n_original_users, n_original_items, n_user_oov_buckets,
                 n_item_oov_buckets, embedding_size, device, prime_pad, dhe_layer_size) -> None:
        super().__init__(user_features, item_features)
        self._common(n_original_users, n_original_items, n_user_oov_buckets, n_item_oov_buckets, embedding_size,
                     device, prime_pad)
        self.user_feature_mat = build_feature_matrix(user_features, self.n_new_users, True, device)
        self.item_feature_mat = build_feature_matrix(item_features, self.n_new_items, True, device)
        self.user_hash_net = _hash_mlp(self.user_feature_mat.size(1), dhe_layer_size, embedding_size, device)
        self.item_hash_net = _hash_mlp(self.item_feature_mat.size(1), dhe_layer_size, embedding_size, device)

    def _lookup(self, old_ids):
        if self.training:
            return _strip_prime_pad_(old_ids.clone(), self.prime_pad)
        return old_ids

    def _hash_users(self, users, feat_lookup_users):
        return _run_hash_net(self.user_hash_net, ops.gather_rows(feat_lookup_users, self.user_feature_mat))

    def _hash_items(self, items, feat_lookup_items):
        return _run_hash_net(self.item_hash_net, ops.gather_rows(feat_lookup_items, self.item_feature_mat))

    def embed_user_ids(self, old_user_ids, model):
        return self._hash_users(old_user_ids, self._lookup(old_user_ids))

    def embed_item_ids(self, old_item_ids, model):
        return self._hash_items(old_item_ids, self._lookup(old_item_ids))


class KNNInductiveEmbedder(_FeatureEmbedder):
    """'knn': mean of the embeddings of the n_neighbors most similar in-vocabulary rows.

    Deliberate deviation ("parity unpinned"): the reference searches with
    ScaNN (approximate, absent from this image, knn_embedder.py:84-93,100-102); here the search is
    EXACT max-inner-product top-k on the f32 matrix cores (ops.score_topk), ties to the lower row.
    The aggregate keeps the reference's hard-coded `.split(2)` (knn_embedder.py:126,147)."""

    def __init__(self, user_features, item_features, n_original_users, n_original_items, n_user_oov_buckets,
                 n_item_oov_buckets, embedding_size, device, prime_pad, n_neighbors=2) -> None:
        super().__init__(user_features, item_features)
        self._common(n_original_users, n_original_items, n_user_oov_buckets, n_item_oov_buckets, embedding_size,
                     device, prime_pad)
        self.n_neighbors = n_neighbors
        # per-feature normalise, stack, then normalise rows again (knn_embedder.py:73-80); kept in HBM
        self.user_feature_mat = F.normalize(build_feature_matrix(user_features, self.n_new_users, True, device))
        self.item_feature_mat = F.normalize(build_feature_matrix(item_features, self.n_new_items, True, device))

    def __deepcopy__(self, memo):
        # like the reference (knn_embedder.py:95-98) the copy is rebuilt and loses n_neighbors
        from copy import deepcopy
        return KNNInductiveEmbedder(deepcopy(self.user_features, memo), deepcopy(self.item_features, memo),
                                    self.n_original_users, self.n_original_items, self.n_user_oov_buckets,
                                    self.n_item_oov_buckets, self.embedding_size, self.device, self.prime_pad)

    def _hash_node(self, nodes, n_original, feature_mat):
        q = ops.gather_rows(nodes, feature_mat)
        # the searched table is fixed for the embedder's lifetime: its fused-search form is made once, where the
        # reference builds its ScaNN searcher (knn_embedder.py:84-93); None for widths the fused path does not take
        cache = self.__dict__.setdefault("_catalogues", {})
        key = (feature_mat.data_ptr(), n_original)
        cat = cache.get(key)
        if key not in cache or (cat is not None and not cat.fresh()):
            cat = cache[key] = ops.TopkCatalogue.of(feature_mat[:n_original])
        return ops.score_topk(q, cat if cat is not None else feature_mat[:n_original], self.n_neighbors)[1]

    def _hash_users(self, users):
        return self._hash_node(users, self.n_original_users, self.user_feature_mat)

    def _hash_items(self, items):
        return self._hash_node(items, self.n_original_items, self.item_feature_mat)

    def embed_user_ids(self, user_ids, model):
        if self.training:
            _strip_prime_pad_(user_ids, self.prime_pad)
        hashed = self._hash_users(user_ids)
        weight_mat = _general_tables(model)[0]
        return ops.gather_mean(hashed, weight_mat, 2)

    def embed_item_ids(self, item_ids, model):
        if self.training:
            _strip_prime_pad_(item_ids, self.prime_pad)
        hashed = self._hash_items(item_ids)
        weight_mat = _general_tables(model)[1]
        return ops.gather_mean(hashed, weight_mat, 2)


class MeanEmbedder(AbstractInductiveEmbedder):
    """'mean': column mean of the whole table (padding row included), cached forever
    (mean_embedder.py:54-56,76-78), broadcast to [B,D]; no grad."""

    def __init__(self, user_features, item_features, n_original_users, n_original_items, n_user_oov_buckets,
                 n_item_oov_buckets, embedding_size, device) -> None:
        super().__init__(user_features, item_features)
        self.user_feat_mean = None
        self.item_feat_mean = None
        self.n_original_users = n_original_users
        self.n_original_items = n_original_items

    @torch.no_grad()
    def embed_user_ids(self, user_ids, model):
        try:  # the model type is checked on EVERY call, cached mean or not (mean_embedder.py:53-61)
            w = _general_tables(model)[0]
        except ValueError:
            raise ValueError("Invalid model type for mean embedder")
        if self.user_feat_mean is None:
            self.user_feat_mean = ops.col_mean(w)
        return ops.broadcast_rows(self.user_feat_mean, len(user_ids))

    @torch.no_grad()
    def embed_item_ids(self, item_ids, model):
        try:  # the model type is checked on EVERY call, cached mean or not (mean_embedder.py:75-87)
            w = _general_tables(model)[1]
        except ValueError:
            raise ValueError("Invalid model type for mean embedder")
        if self.item_feat_mean is None:
            self.item_feat_mean = ops.col_mean(w)
        return ops.broadcast_rows(self.item_feat_mean, len(item_ids))


class ZeroEmbedder(AbstractInductiveEmbedder):
    """'zero' (zero_embedder.py:30-60)."""

    def __init__(self, user_features, item_features, n_original_users, n_original_items, embedding_size,
                 device) -> None:
        super().__init__(user_features, item_features)
        self.zero_vec = torch.zeros(embedding_size, device=device)
        self.n_original_users = n_original_users
        self.n_original_items = n_original_items

    def embed_user_ids(self, user_ids, model):
        return ops.broadcast_rows(None, len(user_ids), self.zero_vec.numel(), self.zero_vec.device)

    def embed_item_ids(self, item_ids, model):
        return ops.broadcast_rows(None, len(item_ids), self.zero_vec.numel(), self.zero_vec.device)
