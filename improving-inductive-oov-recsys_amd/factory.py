"""String -> class registration, identical to the reference's factory
(R/inductive/get_inductive.py:16-36 mapper, :39-138 embedder): same strings, same config keys,
unknown string -> None, module-global feature cache reset when `mode` changes (:46-50)."""
from .embedders import (DeepHashEmbedder, DNNEmbedder, FeatDeepHashEmbedder, InductiveFeatureCache,
                        KNNInductiveEmbedder, LSHInductiveEmbedder, MeanEmbedder, SingleLSHInductiveEmbedder,
                        ZeroEmbedder)
from .mapper import RandomOOVInductiveMapper

feat_cache = InductiveFeatureCache()

EMBEDDERS = ("knn", "lsh", "slsh", "dhe", "fdhe", "dnn", "mean", "zero")
MAPPERS = ("random",)


def get_inductive_mapper(config, dataset, user_num=None, item_num=None, embedding_size=None, first_order=False):
    if embedding_size is None:
        embedding_size = config["embedding_size"]
    if config["inductive_mapper"] == "random":
        return RandomOOVInductiveMapper(
            user_features=dataset.get_user_feature(), item_features=dataset.get_item_feature(),
            n_original_users=user_num or dataset.user_num, n_original_items=item_num or dataset.item_num,
            n_user_oov_buckets=config["user_oov_buckets"], n_item_oov_buckets=config["item_oov_buckets"],
            embedding_size=embedding_size, device=config["device"], prime_pad=config["oov_prime_pad"],
            hash_function=config["oov_hash_function"])
    return None


def get_inductive_embedder(config, dataset, mode="transductive", user_num=None, item_num=None, embedding_size=None,
                           first_order=False):
    global feat_cache
    if feat_cache.get_mode() != mode:
        feat_cache = InductiveFeatureCache(mode=mode)
    if embedding_size is None:
        embedding_size = config["embedding_size"]
    name = config["inductive_embedder"]
    common = dict(user_features=dataset.get_user_feature(), item_features=dataset.get_item_feature(),
                  n_original_users=user_num or dataset.user_num, n_original_items=item_num or dataset.item_num)
    sized = dict(common, n_user_oov_buckets=config["user_oov_buckets"], n_item_oov_buckets=config["item_oov_buckets"],
                 embedding_size=embedding_size, device=config["device"])
    padded = dict(sized, prime_pad=config["oov_prime_pad"])
    if name == "knn":
        return KNNInductiveEmbedder(**padded, n_neighbors=config["oov_knn_num_neighbors"])
    if name == "lsh":
        return LSHInductiveEmbedder(**padded, normalization_type=config["oov_normalization_type"],
                                    feature_cache=feat_cache)
    if name == "slsh":
        return SingleLSHInductiveEmbedder(**padded, normalization_type=config["oov_normalization_type"])
    if name == "dhe":
        return DeepHashEmbedder(**padded, num_hashes=config["dhe_num_hashes"])
    if name == "fdhe":
        return FeatDeepHashEmbedder(**padded, num_hashes=config["dhe_num_hashes"],
                                    dhe_layer_size=config["dhe_layer_size"])
    if name == "dnn":
        return DNNEmbedder(**padded, dhe_layer_size=config["dhe_layer_size"])
    if name == "mean":
        return MeanEmbedder(**sized)
    if name == "zero":
        return ZeroEmbedder(**common, embedding_size=embedding_size, device=config["device"])
    return None
