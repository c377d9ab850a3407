#!/usr/bin/env python3
"""Golden fixture for the 'fdhe' plugin, made by RUNNING THE REAL REFERENCE (build container only):

    python tests/golden/make_golden_fdhe.py      ->  tests/golden/fdhe.npz

Drives recbole.inductive.feat_dh_embedder.FeatDeepHashEmbedder (feat_dh_embedder.py:86-210) through the leaf shims of
ref_shims.py (csiphash -> an independent pure-Python SipHash-2-4): hidden width dhe_layer_size = 96 (not the 512 of
dhe), K = 8 hashes, two feature columns per side; TRAIN mode with prime-padded ids (the hashes are taken of the
un-stripped id, the feature row of the stripped one, :180-206) and EVAL mode.  Data only: inputs, keys, MLP weights,
hash matrices, the concatenated MLP input, pre-sigmoid activations and outputs.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()

import torch  # noqa: E402
from recbole.data.interaction import Interaction  # noqa: E402
from recbole.inductive.feat_dh_embedder import FeatDeepHashEmbedder  # noqa: E402

PRIME_PAD = 112062759511


def np_(t):
    return t.detach().cpu().numpy()


def features(n, cols, seed, id_name):
    g = torch.Generator().manual_seed(seed)
    d = {id_name: torch.arange(n)}
    for name, width in cols:
        t = torch.randn((n,) if width == 1 else (n, width), generator=g)
        t[0] = 0  # padding row
        d[name] = t
    return Interaction(d)


def main():
    K, D, L, n = 8, 16, 96, 80
    keys = [hashlib.sha256(b"mi-oov-fdhe-key-%d" % j).digest()[:16] for j in range(K)]
    os.makedirs("/tmp/mi_oov_golden_fdhe/hash_keys", exist_ok=True)
    cwd = os.getcwd()
    os.chdir("/tmp/mi_oov_golden_fdhe")  # the class reads ./hash_keys/{K}.hashes relative to the CWD
    json.dump([k.hex() for k in keys], open(f"hash_keys/{K}.hashes", "w"))
    uf = features(n, [("age", 1), ("vec", 5)], 21, "user_id")
    itf = features(n, [("year", 1), ("genre", 7)], 22, "item_id")
    torch.manual_seed(23)
    emb = FeatDeepHashEmbedder(uf, itf, 40, 40, 8, 8, D, "cpu", PRIME_PAD, K, L)
    os.chdir(cwd)
    assert [k.hex() for k in emb.hash_keys] == [k.hex() for k in keys]
    base = torch.tensor([0, 1, 2, 39, 40, 41, 79, 5, 63, 17], dtype=torch.int64)
    padded = base.clone()
    padded[[1, 4, 6, 8]] += PRIME_PAD  # OOV-training augmentation: ids shifted by the prime pad
    out = {"keys": np.frombuffer(b"".join(keys), dtype=np.uint8).reshape(K, 16), "ids_eval": np_(base), "ids_train": np_(padded),
           "user_feature_mat": np_(emb.user_feature_mat), "item_feature_mat": np_(emb.item_feature_mat),
           "dims": np.array([K, D, L, n]),
           "user_age": np_(uf["age"]), "user_vec": np_(uf["vec"]), "item_year": np_(itf["year"]), "item_genre": np_(itf["genre"])}
    with torch.no_grad():
        for mode, ids in (("eval", base), ("train", padded)):
            emb.set_train() if mode == "train" else emb.set_eval()
            for side, net, fm in (("user", emb.user_hash_net, emb.user_feature_mat), ("item", emb.item_hash_net, emb.item_feature_mat)):
                keep = ids.clone()
                res = emb.embed_user_ids(ids, None) if side == "user" else emb.embed_item_ids(ids, None)
                assert torch.equal(ids, keep), "fdhe must not modify the caller's ids"
                hm = emb._hash_ids(ids).float()
                stripped = torch.where(ids >= PRIME_PAD, ids - PRIME_PAD, ids) if mode == "train" else ids
                nn_in = torch.hstack((hm, fm[stripped]))
                assert torch.equal(net(nn_in), res)
                out[f"{mode}_{side}_hashes"] = np_(hm)
                out[f"{mode}_{side}_input"] = np_(nn_in)
                out[f"{mode}_{side}_pre_sigmoid"] = np_(net[:-1](nn_in))
                out[f"{mode}_{side}_out"] = np_(res)
    for k, v in emb.state_dict().items():
        out["sd__" + k.replace(".", "__")] = np_(v)
    np.savez_compressed(os.path.join(HERE, "fdhe.npz"), **out)
    print("fdhe.npz:", {k: v.shape for k, v in out.items() if not k.startswith("sd__")})


if __name__ == "__main__":
    main()
