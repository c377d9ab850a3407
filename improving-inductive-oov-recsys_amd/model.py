"""BPR with the inductive lookup path, the caller of the plugin.

Mirrors, name for name, the parts of the reference model that sit on the hot path:
    InductiveGeneralRecommender   R/model/abstract_recommender.py:117-163 (OOV bucket tables :134-139,
                                  set_oov_train/set_oov_eval :147-163)
    BPR                           R/model/general_recommender/bpr.py:31-163
Every per-batch tensor op of the reference's get_*_embedding / predict / full_sort_predict is a
libmi_oov kernel here (gather, splice, lsh, row dot, f32-MFMA scoring).  Training orchestration
(Trainer, optimisers, samplers) is out of scope and stays whatever drives this module.
"""
import os

import torch
from torch import nn

from . import ops
from .embedders import LSHInductiveEmbedder, SingleLSHInductiveEmbedder


def xavier_normal_initialization(module):
    """R/model/init.py: Embedding and Linear weights xavier-normal, Linear bias 0.  Applied with
    self.apply(), it re-initialises bucket tables and embedder MLPs but never the LSH planes
    (a ParameterList), exactly as in the reference (bpr.py:46)."""
    # (the reference writes through `.data`; here on the Parameter itself -- nn.init runs under no_grad -- so that torch's
    # version counter moves and what is derived from a weight, ops.LshTable / ops.LinearX3Weights, sees a re-initialisation
    # after the first forward call.  Same generator draws, same values.)
    if isinstance(module, nn.Embedding):
        nn.init.xavier_normal_(module.weight)
    elif isinstance(module, nn.Linear):
        nn.init.xavier_normal_(module.weight)
        if module.bias is not None:
            nn.init.constant_(module.bias, 0)


_SYNC_FREE_TRAIN = os.environ.get("MI_OOV_TRAIN_LOOKUP", "1") != "0"  # developer A/B knob (tools/train_step_time.py)


class BPRLoss(nn.Module):
    """R/model/loss.py:21-47."""

    def __init__(self, gamma=1e-10):
        super().__init__()
        self.gamma = gamma

    def forward(self, pos_score, neg_score):
        return -torch.log(self.gamma + torch.sigmoid(pos_score - neg_score)).mean()


class InductiveGeneralRecommender(nn.Module):
    def __init__(self, config, dataset, inductive_mapper=None, inductive_embedder=None):
        super().__init__()
        self.USER_ID = config["USER_ID_FIELD"]
        self.ITEM_ID = config["ITEM_ID_FIELD"]
        self.NEG_ITEM_ID = config["NEG_PREFIX"] + self.ITEM_ID
        self.n_users = dataset.num(self.USER_ID)
        self.n_items = dataset.num(self.ITEM_ID)
        self.device = config["device"]

        self.n_user_oov_buckets = 0
        self.n_item_oov_buckets = 0
        self.embedding_size = config["embedding_size"]
        self.inductive_mapper = inductive_mapper
        self.inductive_embedder = inductive_embedder
        self.oov_freeze_embedding = config["oov_freeze_embedding"]
        self.oov_training = False
        if self.inductive_mapper is None and self.inductive_embedder is None:
            raise NotImplementedError("Must provide either self.inductive_mapper or self.inductive_embedder")
        self.n_new_items = (self.inductive_mapper.n_new_items if self.inductive_mapper
                            else self.inductive_embedder.n_new_items)
        if config["add_oov_buckets"]:
            self.n_user_oov_buckets = config["user_oov_buckets"]
            self.user_oov_buckets = nn.Embedding(self.n_user_oov_buckets, self.embedding_size)
            self.n_item_oov_buckets = config["item_oov_buckets"]
            self.item_oov_buckets = nn.Embedding(self.n_item_oov_buckets, self.embedding_size)

    def set_oov_train(self, no_freeze=False):
        self.oov_training = True
        if self.inductive_mapper is not None:
            self.inductive_mapper.set_train()
        if self.inductive_embedder is not None:
            self.inductive_embedder.set_train()
        if self.oov_freeze_embedding and not no_freeze:
            self.freeze_non_oov_layers()

    def set_oov_eval(self, no_freeze=False):
        self.oov_training = False
        if self.inductive_mapper is not None:
            self.inductive_mapper.set_eval()
        if self.inductive_embedder is not None:
            self.inductive_embedder.set_eval()
        if self.oov_freeze_embedding and not no_freeze:
            self.unfreeze_non_oov_layers()

    def freeze_non_oov_layers(self):
        raise NotImplementedError()

    def unfreeze_non_oov_layers(self):
        raise NotImplementedError()


class BPR(InductiveGeneralRecommender):
    def __init__(self, config, dataset, inductive_mapper=None, inductive_embedder=None):
        super().__init__(config, dataset, inductive_mapper, inductive_embedder)
        self.embedding_size = config["embedding_size"]
        self.user_embedding = nn.Embedding(self.n_users, self.embedding_size)
        self.item_embedding = nn.Embedding(self.n_items, self.embedding_size)
        self.loss = BPRLoss()
        self.apply(xavier_normal_initialization)

    # ---- lookups (bpr.py:48-125) --------------------------------------------------------------
    def _lookup(self, ids, side):
        user = side == "user"
        table = (self.user_embedding if user else self.item_embedding).weight
        n_vocab = self.n_users if user else self.n_items
        if self.inductive_mapper is not None:
            ids = self.inductive_mapper.map_user_ids(ids) if user else self.inductive_mapper.map_item_ids(ids)
        emb = self.inductive_embedder
        if self._fused_lsh_inference():
            # one launch: in-vocabulary rows and lsh rows spliced inside the kernel
            feat, planes = emb.hot_operands(side)
            buckets = (self.user_oov_buckets if user else self.item_oov_buckets).weight
            return ops.lsh_lookup(ids, table, feat, planes, buckets)
        if isinstance(emb, LSHInductiveEmbedder) and torch.is_grad_enabled() and _SYNC_FREE_TRAIN:
            # training with the lsh plugin: one sync-free path (no boolean-mask indexing), same values and gradients
            feat, planes = emb.hot_operands(side)
            buckets = (self.user_oov_buckets if user else self.item_oov_buckets).weight
            feat_ids = torch.where(ids >= emb.prime_pad, ids - emb.prime_pad, ids) if emb.training else ids
            return ops.lsh_train_lookup(ids, feat_ids, table, feat, planes, buckets)
        if torch.is_grad_enabled() and _SYNC_FREE_TRAIN and (emb is None or isinstance(emb, SingleLSHInductiveEmbedder)):
            buckets = (self.user_oov_buckets if user else self.item_oov_buckets).weight
            if emb is None:  # mapper only: the mapped id itself addresses the bucket table (bpr.py:75,122)
                idx = ids - n_vocab
            else:
                feat, planes = emb.hot_operands(side)
                feat_ids = torch.where(ids >= emb.prime_pad, ids - emb.prime_pad, ids) if emb.training else ids
                idx = ops.slsh_index(feat_ids, feat, planes, buckets.shape[0])
            idx = torch.where(ids >= n_vocab, idx, torch.full_like(idx, -1))
            return ops.bucket_train_lookup(ids, idx, table, buckets)
        oov_mask = ids >= n_vocab
        oov_ids = ids[oov_mask]  # fresh copy: the embedder may strip prime_pad in place
        if oov_ids.numel() == 0:
            return ops.gather_rows(ids, table)
        if emb is not None:
            oov_rows = emb.embed_user_ids(oov_ids, self) if user else emb.embed_item_ids(oov_ids, self)
        else:
            buckets = (self.user_oov_buckets if user else self.item_oov_buckets).weight
            oov_rows = ops.gather_rows(oov_ids - n_vocab, buckets)
        return ops.splice_rows(ids, table, oov_rows)

    def get_user_embedding(self, new_user_ids):
        return self._lookup(new_user_ids, "user")

    def get_item_embedding(self, item):
        return self._lookup(item, "item")

    def _user_id_lookup(self, user_ids):
        return ops.gather_rows(user_ids, self.user_embedding.weight)

    def _item_id_lookup(self, item_ids):
        return ops.gather_rows(item_ids, self.item_embedding.weight)

    def freeze_non_oov_layers(self):
        self.user_embedding.weight.requires_grad = False
        self.item_embedding.weight.requires_grad = False

    def unfreeze_non_oov_layers(self):
        self.user_embedding.weight.requires_grad = True
        self.item_embedding.weight.requires_grad = True

    def forward(self, user, item):
        return self.get_user_embedding(user), self.get_item_embedding(item)

    # ---- loss / scoring (bpr.py:132-163) ----------------------------------------------------------
    def calculate_loss(self, interaction):
        user = interaction[self.USER_ID]
        pos_item = interaction[self.ITEM_ID]
        neg_item = interaction[self.NEG_ITEM_ID]
        user_e, pos_e = self.forward(user, pos_item)
        neg_e = self.get_item_embedding(neg_item)
        return self.loss(ops.rowdot(user_e, pos_e), ops.rowdot(user_e, neg_e))

    def _fused_lsh_inference(self):
        emb = self.inductive_embedder
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        return isinstance(emb, LSHInductiveEmbedder) and not needs_grad and not emb.training

    def predict(self, interaction):
        user, item = interaction[self.USER_ID], interaction[self.ITEM_ID]
        if self._fused_lsh_inference():
            # two launches: user lookup, then item lookup fused with the row dot (item rows never stored)
            user_e = self.get_user_embedding(user)
            emb = self.inductive_embedder
            if self.inductive_mapper is not None:
                item = self.inductive_mapper.map_item_ids(item)
            feat, planes = emb.hot_operands("item")
            return ops.lsh_lookup_score(item, self.item_embedding.weight, feat, planes, self.item_oov_buckets.weight, user_e)
        user_e, item_e = self.forward(user, item)
        return ops.rowdot(user_e, item_e)

    def predict_multi(self, interactions):
        """[predict(i) for i in interactions] for K queued batches OF THE SAME SIZE (a serving / evaluation loop with
        its batches queued).  With the lsh plugin in inference: TWO persistent launches for all K batches -- the user
        rows (in-vocabulary row or lsh row per id), then the item lookups fused with the row dot -- instead of 2 K
        launches (bpr.py:145-149 over :48-125, K times).  Anything else: the per-batch path, K times."""
        if not interactions:
            return []
        sizes = {int(i[self.USER_ID].numel()) for i in interactions}
        if not self._fused_lsh_inference() or len(sizes) != 1 or len(interactions) == 1:
            return [self.predict(i) for i in interactions]
        emb = self.inductive_embedder
        users = [i[self.USER_ID] for i in interactions]
        items = [i[self.ITEM_ID] for i in interactions]
        if self.inductive_mapper is not None:
            users = [self.inductive_mapper.map_user_ids(u) for u in users]
            items = [self.inductive_mapper.map_item_ids(t) for t in items]
        (ufeat, uplanes), (ifeat, iplanes) = emb.hot_operands("user"), emb.hot_operands("item")
        user_rows = ops.lsh_lookup_multi(users, self.user_embedding.weight, ufeat, uplanes, self.user_oov_buckets.weight,
                                         lsh_table=emb.lsh_table("user", self))
        return ops.lsh_lookup_multi(items, self.item_embedding.weight, ifeat, iplanes, self.item_oov_buckets.weight,
                                    other_list=user_rows, lsh_table=emb.lsh_table("item", self))

    def ind_full_sort_predict(self, interaction, item_ids):
        user_e = self.get_user_embedding(interaction[self.USER_ID])
        all_item_e = self.get_item_embedding(item_ids)
        return ops.full_sort_scores(user_e, all_item_e).view(-1)

    def full_sort_predict(self, interaction):
        user_e = self.get_user_embedding(interaction[self.USER_ID])
        return ops.full_sort_scores(user_e, self.item_embedding.weight).view(-1)

    def full_sort_topk(self, interaction, k, skip_padding=True):
        """Fused scores + per-user top-k over the in-vocabulary catalogue: what the evaluator does
        with full_sort_predict's output (trainer.py:541-544 masks item 0, collector.py:158-167 topk)
        without ever materialising [B, n_items]."""
        user_e = self.get_user_embedding(interaction[self.USER_ID])
        return ops.score_topk(user_e, self.item_embedding.weight, k, 1 if skip_padding else 0)


class DirectAU(BPR):
    """DirectAU with the OOV splice (R/model/general_recommender/directau.py:25-186): the second general recommender
    that calls the plugin (get_user_embedding / get_item_embedding at :132,167 are BPR's lookups verbatim, so the
    kernels behind them are the same); rows are L2-normalised before the alignment / uniformity loss and before
    the row dot of `predict`.  The loss arithmetic (pdist, norms, logs) is ordinary torch on [B, D] rows."""

    def __init__(self, config, dataset, inductive_mapper=None, inductive_embedder=None):
        super().__init__(config, dataset, inductive_mapper, inductive_embedder)
        self.gamma = config["gamma"]
        self.detach = config["detach"] if "detach" in config else False
        self.restore_user_e = None
        self.restore_item_e = None
        self.other_parameter_name = ["restore_user_e", "restore_item_e"]

    def forward(self, user, item):
        user_e, item_e = self.get_user_embedding(user), self.get_item_embedding(item)
        return torch.nn.functional.normalize(user_e, dim=-1), torch.nn.functional.normalize(item_e, dim=-1)

    @staticmethod
    def alignment(x, y, alpha=2):
        return (x - y).norm(p=2, dim=1).pow(alpha).mean()

    @staticmethod
    def uniformity(x, t=2):
        return torch.pdist(x, p=2).pow(2).mul(-t).exp().mean().log()

    def calculate_loss(self, interaction):
        if self.restore_user_e is not None or self.restore_item_e is not None:
            self.restore_user_e, self.restore_item_e = None, None
        user_e, item_e = self.forward(interaction[self.USER_ID], interaction[self.ITEM_ID])
        align = self.alignment(user_e, item_e)
        uniform = self.gamma * (self.uniformity(user_e) + self.uniformity(item_e)) / 2
        return align + uniform

    def predict(self, interaction):
        user_e, item_e = self.forward(interaction[self.USER_ID], interaction[self.ITEM_ID])
        return ops.rowdot(user_e, item_e)

    def full_sort_predict(self, interaction):
        raise NotImplementedError()  # as the reference (directau.py:172)

    def ind_full_sort_predict(self, interaction, item_ids):  # un-normalised, as the reference (directau.py:181-186)
        user_e = self.get_user_embedding(interaction[self.USER_ID])
        return ops.full_sort_scores(user_e, self.get_item_embedding(item_ids)).view(-1)

    def full_sort_topk(self, interaction, k, skip_padding=True):
        raise NotImplementedError()

