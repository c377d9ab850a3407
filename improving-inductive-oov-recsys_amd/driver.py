"""Minimal end-to-end driver behind the reference's `run_recbole.py` command line (S/run_recbole.py:202-266).

SCOPE.  RecBole's config / data / trainer / evaluator layers are out of scope of this build
(SURVEY.md section 2.1); what the north star asks to keep is the ENTRY: the same `--key=value` flags
driving the inductive path end to end.  This module therefore implements only the minimum around the
kernels, deliberately simplified and documented as such:

  * atomic-file loader (`<name>.inter/.user/.item`, `field:type` headers, token ids remapped with 0 =
    padding like RecBole) -> feature tables the embedders consume;
  * an inductive split: the last `--oov_fraction` of the remapped user / item ids are out of
    vocabulary (no embedding rows), 10 % of the interactions are held out for testing;
  * BPR training with Adam and uniform negatives; `--train_oov` adds the reference's OOV augmentation
    (R/trainer/trainer.py:1654-1667,1748-1759: a share of each batch is re-issued with `oov_prime_pad`
    added to the user and/or item id, and entries are zeroed at `oov_feature_mask_rate`);
  * evaluation in the reference driver's mode `uni250` (S/run_recbole.py:214-221): every held-out
    positive is ranked against 250 uniformly drawn items through `model.predict`, i.e. through the fused
    lookup + score kernels; Recall/MRR/NDCG/Hit@k reported overall and for old/new users and items.

Everything numeric on the path (lookups, plugin, scoring) runs on libmi_oov; torch provides autograd,
the optimiser and the host-side sampling.
"""
import json
import math
import os
import random
import sys
import time

import numpy as np
import torch

from .embedders import FeatureTable
from .factory import get_inductive_embedder, get_inductive_mapper
from .model import BPR, DirectAU


# ---- command line: `--k=v` typed, bare `--k` -> True (S/utils/parse.py:44-61) --------------------------
def _convert(s):
    if s.isnumeric():
        return int(s)
    try:
        return float(s)
    except ValueError:
        pass
    if s.lower() in ("true", "false"):
        return s.lower() == "true"
    if s.startswith("[") and s.endswith("]"):
        inner = s[1:-1]
        return [_convert(x) for x in inner.split(",")] if inner else []
    return s


def custom_parse_args(argv=None):
    out = {}
    for arg in (sys.argv if argv is None else argv):
        if not arg.startswith("--"):
            continue
        pieces = arg[2:].split("=", 1)
        out[pieces[0]] = True if len(pieces) == 1 else _convert(pieces[1])
    return out


class Config(dict):
    """dict whose missing keys read as None, like recbole's Config (R/config/configurator.py:583-584)."""

    def __getitem__(self, k):
        return self.get(k, None)


DEFAULTS = dict(  # R/properties/overall.yaml:59-119 and the driver's own settings (S/run_recbole.py:205-229)
    embedding_size=64, train_batch_size=2048, eval_batch_size=100000, learning_rate=1e-3, epochs=5, seed=2020,
    oov_train_ratio=0.2, oov_feature_mask_rate=0.2, oov_prime_pad=112062759511, oov_hash_function="3round",
    oov_only_epoch=True, oov_freeze_embedding=False, dhe_num_hashes=128, dhe_layer_size=512,
    oov_knn_num_neighbors=2, oov_normalization_type="per-feature", topk=[10, 20], oov_fraction=0.2,
    data_path="dataset", USER_ID_FIELD="user_id", ITEM_ID_FIELD="item_id", NEG_PREFIX="neg_", eval_negatives=250,
    nan_policy="skip")  # 'raise' = the reference's _check_nan; 'skip' counts the batch and fails on a no-op phase


# ---- atomic files ---------------------------------------------------------------------------------------
def _read_atomic(path):
    with open(path) as f:
        header = f.readline().rstrip("\n").split("\t")
        names, types = zip(*(h.split(":") for h in header))
        cols = [[] for _ in names]
        for line in f:
            for c, v in zip(cols, line.rstrip("\n").split("\t")):
                c.append(v)
    return list(names), list(types), cols


def _remap(values):
    """token -> id with 0 reserved for padding (RecBole's [PAD]); ids follow the order of first appearance, as
    `pandas.factorize` numbers them in R/data/dataset/dataset.py:1219-1241 (`_remap`)."""
    vocab = {}
    for t in values:
        if t not in vocab:
            vocab[t] = len(vocab) + 1
    return vocab


def _remap_in_order(*token_lists):
    """token -> id in order of first appearance over the lists (0 = [PAD]): RecBole's remap over the concatenated
    benchmark files (R/data/dataset/dataset.py `_remap_ID_all`), which is what puts every training-time id below
    every inductive-only id in an `X_ind` dataset (S/perform_hashing.py:101-138)."""
    vocab = {}
    for tokens in token_lists:
        for t in tokens:
            if t not in vocab:
                vocab[t] = len(vocab) + 1
    return vocab


class AtomicDataset:
    """`<name>.inter` (+ `.user`, `.item`), or -- with `benchmark_filename=[train, ..., test]` as in the reference's
    inductive datasets (`benchmark_filename: ['train', 'empty', 'test_filt']`) -- the pre-split files
    `<name>.<part>.inter`: ids are then numbered by first appearance, train first, so `n_train_users/items` is the
    size of the transductive vocabulary and everything above it is out-of-vocabulary; `split` holds the part of
    every interaction."""

    def __init__(self, name, data_path, user_field="user_id", item_field="item_id", benchmark_filename=None):
        root = os.path.join(data_path, name)
        self.split = None
        if benchmark_filename:
            parts = [p for p in benchmark_filename]
            users_raw, items_raw, split, is_new = [], [], [], []
            per_part = []
            for k, part in enumerate(parts):
                inames, _, icols = _read_atomic(os.path.join(root, f"{name}.{part}.inter"))
                u, i = icols[inames.index(user_field)], icols[inames.index(item_field)]
                per_part.append((u, i))
                users_raw += u
                items_raw += i
                split += [k] * len(u)
                if "is_new" in inames:  # `is_new:token`, -1 old / 1 new (R/data/dataset/dataset.py:174-190)
                    is_new += [v == "1" for v in icols[inames.index("is_new")]]
                else:
                    is_new += [False] * len(u)
            self.is_new = np.array(is_new, dtype=bool)
            self.uvocab = _remap_in_order(*(u for u, _ in per_part))
            self.ivocab = _remap_in_order(*(i for _, i in per_part))
            self.split = np.array(split, dtype=np.int64)
            self.n_train_users = len(set(per_part[0][0])) + 1
            self.n_train_items = len(set(per_part[0][1])) + 1
        else:
            inames, _, icols = _read_atomic(os.path.join(root, f"{name}.inter"))
            users_raw, items_raw = icols[inames.index(user_field)], icols[inames.index(item_field)]
            self.uvocab, self.ivocab = _remap(users_raw), _remap(items_raw)
            self.is_new = None
        for path, field, vocab in ((os.path.join(root, f"{name}.user"), user_field, self.uvocab),
                                   (os.path.join(root, f"{name}.item"), item_field, self.ivocab)):
            if os.path.exists(path):  # entities that only occur in the feature files come last (dataset.py:1162-1192:
                names, _, raw = _read_atomic(path)  # the id field is remapped over inter_feat first, then the feature file)
                for t in raw[names.index(field)]:
                    if t not in vocab:
                        vocab[t] = len(vocab) + 1
        self.user_num, self.item_num = len(self.uvocab) + 1, len(self.ivocab) + 1
        self.inter_user = np.array([self.uvocab[u] for u in users_raw], dtype=np.int64)
        self.inter_item = np.array([self.ivocab[i] for i in items_raw], dtype=np.int64)
        self.field2token_id, self.field2type = {}, {}
        self.user_feat = self._features(os.path.join(root, f"{name}.user"), user_field, self.uvocab, self.user_num)
        self.item_feat = self._features(os.path.join(root, f"{name}.item"), item_field, self.ivocab, self.item_num)

    def _features(self, path, id_field, vocab, n):
        """Feature file -> columns indexed by the remapped entity id (row 0 = padding).  Token fields are numbered by
        first appearance in file order, sequences flattened (dataset.py:1198-1241); a missing float is the mean of the
        present ones, a missing token is [PAD] (dataset.py:655-680 `_fill_nan`); entities the file does not list keep
        the same fill values (dataset.py `_user_item_feat_preparation`: reindex, then `_fill_nan`)."""
        cols = {id_field: torch.arange(n)}
        if not os.path.exists(path):
            return FeatureTable(cols)
        names, types, raw = _read_atomic(path)
        rows = [vocab.get(t, 0) for t in raw[names.index(id_field)]]
        for name, typ, vals in zip(names, types, raw):
            if name == id_field:
                continue
            if typ == "float":
                present = [float(v) for v in vals if v != ""]
                fill = float(np.mean(np.array(present, dtype=np.float64))) if present else 0.0
                t = torch.full((n,), fill, dtype=torch.float32)
                t[0] = 0.0
                t[rows] = torch.tensor([float(v) if v != "" else fill for v in vals], dtype=torch.float32)
            elif typ == "token":
                tv = _remap([v for v in vals if v != ""])
                t = torch.zeros(n, dtype=torch.int64)
                t[rows] = torch.tensor([tv.get(v, 0) for v in vals])
                self.field2token_id[name] = tv
            elif typ in ("token_seq", "float_seq"):
                seqs = [v.split(" ") if v else [] for v in vals]
                width = max(1, max(len(s) for s in seqs))
                if typ == "token_seq":
                    tv = _remap([x for s in seqs for x in s])
                    t = torch.zeros((n, width), dtype=torch.int64)
                    for r, s in zip(rows, seqs):
                        t[r, :len(s)] = torch.tensor([tv[x] for x in s], dtype=torch.int64)
                    self.field2token_id[name] = tv
                else:
                    t = torch.zeros((n, width))
                    for r, s in zip(rows, seqs):
                        t[r, :len(s)] = torch.tensor([float(x) for x in s])
            else:
                continue
            cols[name] = t
            self.field2type[name] = typ
        return FeatureTable(cols)

    def remap_features(self, orig):
        """Bring the feature columns of an inductive dataset onto the numbering of its transductive twin `orig` (the
        dataset the model was trained on), as InductiveDataset.remap_features does before evaluation
        (R/data/dataset/inductive_dataset.py:73-168): every token id becomes the id the same token has in `orig`, a
        token `orig` never saw becomes 0 ([PAD]); token sequences are cut to `orig`'s width; the value a float column
        was mean-filled with becomes `orig`'s fill value.  After it, rows 1..orig.n-1 of every column equal `orig`'s
        (the check S/perform_hashing.py:112-138 prints), so the plugin hashes old entities exactly as in training.
        Returns {field: [tokens orig never saw]}."""
        missing = {}
        for feat, ofeat in ((self.user_feat, orig.user_feat), (self.item_feat, orig.item_feat)):
            for name in feat.columns[1:]:
                if name not in ofeat.columns:
                    continue
                col, ocol = feat[name], ofeat[name]
                typ = self.field2type.get(name)
                if typ in ("token", "token_seq"):
                    mine, theirs = self.field2token_id[name], orig.field2token_id[name]
                    lut = torch.zeros(len(mine) + 1, dtype=torch.int64)
                    for tok, i in mine.items():
                        lut[i] = theirs.get(tok, 0)
                    missing[name] = [tok for tok in mine if tok not in theirs]
                    col = lut[col]
                    if col.ndim > 1 and col.shape[1] != ocol.shape[1]:
                        col = col[:, :ocol.shape[1]]
                    self.field2token_id[name] = dict(theirs, **{tok: 0 for tok in missing[name]})
                elif typ == "float":
                    k = ocol.shape[0]
                    old = col[1:k]
                    diff = ocol[1:] != old
                    if bool(diff.any()):
                        if not (bool((old[diff] == old[diff][0]).all()) and bool((ocol[1:][diff] == ocol[1:][diff][0]).all())):
                            raise AssertionError(f"float feature {name}: old rows differ by more than the fill value")
                        col = col.clone()
                        col[1:k][diff] = ocol[1:][diff][0]
                feat._cols[name] = col
        return missing

    def get_user_feature(self):
        return self.user_feat

    def get_item_feature(self):
        return self.item_feat


class _VocabView:
    """What the model sees as `dataset`: the in-vocabulary sizes."""

    def __init__(self, n_users, n_items, user_field, item_field):
        self._n = {user_field: n_users, item_field: n_items}

    def num(self, field):
        return self._n[field]


# ---- OOV augmentation (R/trainer/trainer.py:1654-1667, 1748-1759) ----------------------------------------
def transform_interaction_oov(batch, cfg, user_key, item_key):
    option = random.choice([0, 1, 2])
    if option in (0, 2):
        batch[item_key] = batch[item_key] + cfg["oov_prime_pad"]
    if option in (1, 2):
        batch[user_key] = batch[user_key] + cfg["oov_prime_pad"]
    rate = cfg["oov_feature_mask_rate"]
    if rate and rate > 0:
        for k in batch:
            mask = torch.rand(batch[k].shape, device=batch[k].device) < rate
            batch[k] = batch[k].masked_fill(mask, 0)
    return batch


def augment_with_oov(batch, cfg, user_key, item_key):
    n = len(batch[user_key])
    sel = torch.rand(n, device=batch[user_key].device) < cfg["oov_train_ratio"]
    extra = transform_interaction_oov({k: v[sel] for k, v in batch.items()}, cfg, user_key, item_key)
    perm = torch.randperm(n + int(sel.sum()), device=sel.device)
    return {k: torch.cat((batch[k], extra[k]))[perm] for k in batch}


# ---- uni250 evaluation: per user, its positives followed by 250 sampled negatives per positive ------------------
def evaluate(model, users, items, tot_items, cfg, n_users, n_items, device, gen):
    """Batches shaped like NegSampleEvalDataLoader's (general_dataloader.py:157-190: per user the positives first,
    then `eval_negatives` sampled items per positive; users packed until eval_batch_size rows), scored through
    model.predict and ranked by the sampled-ranking evaluator (evaluator.py: segment top-k + hits kernels, the
    reference's metric arithmetic).  NaN scores (all-zero lsh codes) are ranked as torch.topk ranks them: first.

    The reference scores one such batch per model.predict call.  Scores and rankings of a user do not depend on what
    else is in its batch, so here the batches are QUEUED: the plan (which users form which batch) is made on the host
    from ONE device -> host copy of the per-user counts -- the reference's dataloader plans on the host as well --,
    the negatives are drawn batch by batch, in the reference's order (so the sampled items do not depend on the
    queueing), and up to `eval_rows_per_launch` rows of consecutive batches are built, scored and ranked as ONE group:
    `mi_oov_eval_rows_build` writes the group's (user, item) rows in one pass from the per-user counts (rows are
    user-contiguous by construction: nothing is sorted), model.predict scores them -- with BPR + lsh one user-row launch
    and one fused lookup + score launch (the persistent kernel of csrc/lsh64p.hip from 524288 rows on) --, and
    `SampledRankingEvaluator.eval_group` ranks them (`mi_oov_segment_dedup`, three `mi_oov_segment_topk` +
    `mi_oov_topk_hits_range` for the nine collectors).  The count vector is the run's only device -> host copy before the
    rec.topk blocks come back in `ev.evaluate()`."""
    from . import ops
    from .evaluator import SampledRankingEvaluator
    ukey, ikey, nneg = cfg["USER_ID_FIELD"], cfg["ITEM_ID_FIELD"], int(cfg["eval_negatives"])
    ev = SampledRankingEvaluator(cfg["topk"], cfg["metrics"] or ("recall", "mrr", "ndcg", "hit", "precision"),
                                 n_old_users=n_users, n_old_items=n_items)
    order = torch.argsort(users, stable=True)
    su, si = users[order], items[order]
    uniq, counts = torch.unique_consecutive(su, return_counts=True)
    cum_h = np.cumsum(counts.cpu().numpy())  # the run's one device -> host copy: the batch plan is made on the host
    budget = int(cfg["eval_batch_size"])
    group_rows = max(budget, int(cfg["eval_rows_per_launch"] or (1 << 22)))
    # plan: batches of whole users with at most `budget` rows (a user with more rows is a batch of its own): one binary
    # search per BATCH over the cumulative counts (a Python loop over the users was 15 ms of a 10 M-row evaluation) ...
    batches, lo_u, pos_lo, per_batch = [], 0, 0, budget // (1 + nneg)
    while lo_u < len(cum_h):
        hi_u = max(lo_u + 1, int(np.searchsorted(cum_h, pos_lo + per_batch, side="right")))
        npos = int(cum_h[hi_u - 1]) - pos_lo
        batches.append((lo_u, hi_u, pos_lo, npos))
        lo_u, pos_lo = hi_u, pos_lo + npos
    # ... and groups of consecutive batches of at most `group_rows` rows, each ONE model.predict / evaluator call
    groups, cur, cur_rows = [], [], 0
    for bt in batches:
        rows = bt[3] * (1 + nneg)
        if cur and cur_rows + rows > group_rows:
            groups.append(cur)
            cur, cur_rows = [], 0
        cur.append(bt)
        cur_rows += rows
    if cur:
        groups.append(cur)
    # positives CSR over the sorted users (device); a group's slice of it, rebased, is the group's CSR
    pos_ptr_all = torch.cat((torch.zeros(1, dtype=torch.int64, device=device), torch.cumsum(counts, 0)))
    with torch.no_grad():
        for grp in groups:
            g_lo, g_hi, p_lo = grp[0][0], grp[-1][1], grp[0][2]
            n_pos = sum(bt[3] for bt in grp)
            # negatives: ONE draw per reference batch, in the reference's order (so the sampled items do not depend on the
            # queueing), each straight into its place of the group's array -- a batch's negatives lie in user order
            # (general_dataloader.py:157-190), so consecutive batches concatenate to the layout mi_oov_eval_rows_build reads
            neg = torch.empty((n_pos * nneg,), dtype=torch.int64, device=device)
            off = 0
            for _, _, _, npos in grp:
                torch.randint(1, tot_items, (npos * nneg,), generator=gen, device=device, out=neg[off:off + npos * nneg])
                off += npos * nneg
            g_users = uniq[g_lo:g_hi]
            pos_i = si[p_lo:p_lo + n_pos]
            pos_ptr = pos_ptr_all[g_lo:g_hi + 1] - p_lo
            # the group's rows in one pass: per user its positives, then its negatives; rows are user-contiguous by
            # construction, so nothing is sorted and the evaluator is told so (eval_group)
            row_user, row_item, seg_ptr = ops.eval_rows_build(pos_ptr, g_users, pos_i, neg, nneg)
            scores = model.predict({ukey: row_user, ikey: row_item.clone()})
            ev.eval_group(scores, g_users, row_item, seg_ptr, pos_ptr, pos_i)
    return ev.evaluate()


def run(args):
    cfg = Config({**DEFAULTS, **args})
    if cfg["model"] not in (None, "BPR", "DirectAU"):
        raise NotImplementedError(f"model {cfg['model']}: only the general recommenders that call the plugin (BPR, "
                                  "DirectAU) are built (SURVEY.md section 2.1)")
    if not torch.cuda.is_available():
        raise RuntimeError("run_recbole needs an MI355X (ROCm device): the path has no CPU fallback")
    device = torch.device("cuda", int(cfg["gpu_id"] or 0))
    cfg["device"] = device
    seed = int(cfg["seed"])
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    ukey, ikey = cfg["USER_ID_FIELD"], cfg["ITEM_ID_FIELD"]
    bench_files = cfg["benchmark_filename"]
    if isinstance(bench_files, str):
        bench_files = [p.strip(" '\"[]") for p in bench_files.split(",") if p.strip(" '\"[]")]
    ds = AtomicDataset(cfg["dataset"], cfg["data_path"], ukey, ikey, benchmark_filename=bench_files)
    if cfg["orig_dataset"]:  # the transductive twin the checkpoint was trained on (S/perform_hashing.py:101-109:
        orig = AtomicDataset(cfg["orig_dataset"], cfg["data_path"], ukey, ikey)  # `ind_dataset.set_orig_dataset`)
        missing = ds.remap_features(orig)
        print("feature tokens the transductive dataset never saw (-> [PAD]):",
              {k: len(v) for k, v in missing.items() if v})
    if ds.split is not None:  # pre-split inductive dataset: vocabulary = what the train part contains
        n_users, n_items = ds.n_train_users, ds.n_train_items
        is_test = ds.split == ds.split.max()
        keep = (ds.split == 0) | is_test
        ds.inter_user, ds.inter_item, is_test = ds.inter_user[keep], ds.inter_item[keep], is_test[keep]
    else:
        n_users = max(2, int(ds.user_num * (1 - cfg["oov_fraction"])))
        n_items = max(2, int(ds.item_num * (1 - cfg["oov_fraction"])))
        rng = np.random.default_rng(seed)
        is_test = rng.random(len(ds.inter_user)) < 0.1
    tu = torch.from_numpy(ds.inter_user[~is_test]).to(device)
    ti = torch.from_numpy(ds.inter_item[~is_test]).to(device)
    eu = torch.from_numpy(ds.inter_user[is_test]).to(device)
    ei = torch.from_numpy(ds.inter_item[is_test]).to(device)

    embedder = get_inductive_embedder(cfg, ds, user_num=n_users, item_num=n_items)
    mapper = get_inductive_mapper(cfg, ds, user_num=n_users, item_num=n_items)
    model_cls = DirectAU if cfg["model"] == "DirectAU" else BPR
    if cfg["gamma"] is None:
        cfg["gamma"] = 1.0  # R/properties/model/DirectAU.yaml
    model = model_cls(cfg, _VocabView(n_users, n_items, ukey, ikey), mapper, embedder).to(device)
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=cfg["learning_rate"])
    gen = torch.Generator(device=device).manual_seed(seed)
    print(f"dataset {cfg['dataset']}: {ds.user_num} users ({n_users} in vocabulary), {ds.item_num} items "
          f"({n_items} in vocabulary), {len(tu)} train / {len(eu)} test interactions; embedder "
          f"{type(embedder).__name__ if embedder else None}, mapper {type(mapper).__name__ if mapper else None}")

    bs = int(cfg["train_batch_size"])
    neg_hi = n_items if ds.split is not None else ds.item_num  # pre-split: negatives from the training catalogue
    if cfg["load_checkpoint"]:
        # Loaded with weights_only=True: nothing in the file is executed.  This package's own checkpoints are
        # {'state_dict': tensors}.  A checkpoint written by the REFERENCE (trainer.py:304-313) pickles its Config object and
        # the optimizer state next to 'state_dict'; the safe loader refuses such a file (it does not skip the foreign
        # objects), so it has to be re-exported as tensors only first -- the state_dict KEYS are the reference's.
        import pickle
        try:
            blob = torch.load(cfg["load_checkpoint"], map_location=device, weights_only=True)
        except pickle.UnpicklingError as e:
            raise RuntimeError(
                f"{cfg['load_checkpoint']} holds pickled non-tensor objects (a reference checkpoint stores its Config and "
                "optimizer next to 'state_dict') and is not loaded: re-export it where it was written with "
                "torch.save({'state_dict': model.state_dict()}, path)") from e
        model.load_state_dict(blob["state_dict"] if isinstance(blob, dict) and "state_dict" in blob else blob)
        print(f"loaded {cfg['load_checkpoint']}")
    for epoch in range(0 if cfg["eval_only"] else int(cfg["epochs"])):
        t0, total, nb = time.time(), 0.0, 0
        skipped = {"iv": [0, 0], "oov": [0, 0]}  # phase -> [batches with a non-finite loss, batches tried]
        perm = torch.randperm(len(tu), generator=gen, device=device)
        oov_pass = bool(cfg["train_oov"]) and bool(cfg["oov_only_epoch"])
        for phase in (("iv", "oov") if oov_pass else ("iv",)):
            if phase == "oov":
                model.set_oov_train()
            for lo in range(0, len(perm), bs):
                idx = perm[lo:lo + bs]
                batch = {ukey: tu[idx], ikey: ti[idx],
                         cfg["NEG_PREFIX"] + ikey: torch.randint(1, neg_hi, (len(idx),), generator=gen, device=device)}
                if phase == "oov":
                    if random.random() > cfg["oov_train_ratio"]:
                        continue
                    batch = transform_interaction_oov(batch, cfg, ukey, ikey)
                elif cfg["train_oov"] and not cfg["oov_only_epoch"]:
                    model.set_oov_train(no_freeze=True)
                    batch = augment_with_oov(batch, cfg, ukey, ikey)
                loss = model.calculate_loss(batch)
                skipped[phase][1] += 1
                if not torch.isfinite(loss):
                    # an lsh row whose code is all zeros is 0/0 = NaN (lsh_embedder.py:178) and poisons the batch loss.
                    # The reference aborts here (trainer.py _check_nan: ValueError); nan_policy='skip' drops the batch,
                    # counts it, and refuses to let a whole phase turn into a silent no-op.
                    if cfg["nan_policy"] == "raise":
                        raise ValueError("Training loss is nan")
                    skipped[phase][0] += 1
                    continue
                opt.zero_grad()
                loss.backward()
                opt.step()
                total, nb = total + loss.item(), nb + 1
            if phase == "oov":
                model.set_oov_eval()
            bad, tried = skipped[phase]
            if tried and bad == tried:
                raise ValueError(f"epoch {epoch}: every one of the {tried} batches of the '{phase}' phase had a non-finite "
                                 "loss (all-zero lsh codes give NaN rows): the phase trained nothing")
        n_bad = skipped["iv"][0] + skipped["oov"][0]
        print(f"epoch {epoch}: loss {total / max(1, nb):.4f} over {nb} batches, {time.time() - t0:.2f}s"
              + (f"; {n_bad} batches skipped for a non-finite loss (iv {skipped['iv'][0]}/{skipped['iv'][1]}, "
                 f"oov {skipped['oov'][0]}/{skipped['oov'][1]})" if n_bad else ""))
    if cfg["save_checkpoint"]:
        torch.save({"state_dict": model.state_dict()}, cfg["save_checkpoint"])  # tensors only: weights_only-loadable
        print(f"saved {cfg['save_checkpoint']}")
    model.eval()
    model.set_oov_eval()
    results = evaluate(model, eu, ei, ds.item_num, cfg, n_users, n_items, device, gen)
    print(json.dumps(results))
    return results, model
