"""mi_oov -- MI355X-native inductive OOV embedding path (drop-in for the reference's
`recbole.inductive` plugin surface + the BPR lookups/scoring that call it).

The directory name `improving-inductive-oov-recsys_amd` is not a Python identifier; import the
package as `mi_oov` (the alias module at the repository root) or through importlib.  All
submodules are imported eagerly and registered under the alias as well, so
`from mi_oov.embedders import LSHInductiveEmbedder` and attribute access on the package return
the same module objects.
"""
import sys as _sys

from . import _cabi, ops, embedders, mapper, factory, model, sharded, context, evaluator, driver, torch_ops  # noqa: F401
from ._cabi import LIB_PATH, MiOovError, available  # noqa: F401
from .embedders import (AbstractInductiveEmbedder, DeepHashEmbedder, DNNEmbedder, FeatDeepHashEmbedder,  # noqa: F401
                        FeatureTable, InductiveFeatureCache, KNNInductiveEmbedder, LSHInductiveEmbedder, MeanEmbedder,
                        SingleLSHInductiveEmbedder, TorchLSHash, ZeroEmbedder)
from .factory import get_inductive_embedder, get_inductive_mapper  # noqa: F401
from .mapper import AbstractInductiveMapper, RandomOOVInductiveMapper  # noqa: F401
from .model import BPR, DirectAU, InductiveGeneralRecommender  # noqa: F401

__version__ = "0.1.0"

for _name in ("_cabi", "ops", "embedders", "mapper", "factory", "model", "sharded", "context", "evaluator", "driver",
              "torch_ops"):
    _sys.modules.setdefault("mi_oov." + _name, _sys.modules[__name__ + "." + _name])
_sys.modules.setdefault("mi_oov", _sys.modules[__name__])
