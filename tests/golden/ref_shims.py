"""Test-harness-only shims that make the Python reference importable in the build container.

Used ONLY by tests/golden/make_golden.py (run by hand in the container that has
/root/reference mounted).  Nothing here travels into the product path and nothing here is
needed on the GPU box: the committed fixtures under tests/golden/*.npz|json are what the
tests read.

What is stubbed: leaf third-party wheels that are absent from this
image and carry no arithmetic of the path -- colorlog, colorama, texttable, thop,
tensorboard, pyLSHash (only a default-argument type), wandb -- plus the two that DO carry
arithmetic and are not vendored in the reference:

* csiphash.siphash24  -> bound to the pure-Python SipHash-2-4 below (public algorithm,
  Aumasson & Bernstein 2012; checked against the paper's test vector in make_golden.py).
* scann               -> bound to an EXACT brute-force max-inner-product searcher (numpy).
  ScaNN itself is approximate and absent, so the neighbour search is "parity unpinned";
  the aggregate that consumes the neighbour indices is pinned.
"""
import logging
import sys
import types

import numpy as np

REF_ROOT = "/root/reference/RecBole"

_MASK = (1 << 64) - 1


def _rotl(x, b):
    return ((x << b) | (x >> (64 - b))) & _MASK


def siphash24_py(key: bytes, msg: bytes) -> bytes:
    """SipHash-2-4, 64-bit output returned as 8 little-endian bytes (csiphash convention)."""
    assert len(key) == 16
    k0 = int.from_bytes(key[:8], "little")
    k1 = int.from_bytes(key[8:], "little")
    v0 = k0 ^ 0x736F6D6570736575
    v1 = k1 ^ 0x646F72616E646F6D
    v2 = k0 ^ 0x6C7967656E657261
    v3 = k1 ^ 0x7465646279746573

    def rounds(n, v0, v1, v2, v3):
        for _ in range(n):
            v0 = (v0 + v1) & _MASK
            v1 = _rotl(v1, 13) ^ v0
            v0 = _rotl(v0, 32)
            v2 = (v2 + v3) & _MASK
            v3 = _rotl(v3, 16) ^ v2
            v0 = (v0 + v3) & _MASK
            v3 = _rotl(v3, 21) ^ v0
            v2 = (v2 + v1) & _MASK
            v1 = _rotl(v1, 17) ^ v2
            v2 = _rotl(v2, 32)
        return v0, v1, v2, v3

    n = len(msg)
    full = n - (n % 8)
    for off in range(0, full, 8):
        m = int.from_bytes(msg[off:off + 8], "little")
        v3 ^= m
        v0, v1, v2, v3 = rounds(2, v0, v1, v2, v3)
        v0 ^= m
    last = (n & 0xFF) << 56
    tail = msg[full:]
    for i, b in enumerate(tail):
        last |= b << (8 * i)
    v3 ^= last
    v0, v1, v2, v3 = rounds(2, v0, v1, v2, v3)
    v0 ^= last
    v2 ^= 0xFF
    v0, v1, v2, v3 = rounds(4, v0, v1, v2, v3)
    return ((v0 ^ v1 ^ v2 ^ v3) & _MASK).to_bytes(8, "little")


class _ExactSearcher:
    """Stand-in for a built ScaNN searcher: exact top-k by dot product, ties -> lowest index."""

    def __init__(self, db):
        self.db = np.asarray(db, dtype=np.float32)

    def search_batched(self, q, final_num_neighbors=None):
        q = np.asarray(q, dtype=np.float32)
        scores = q.astype(np.float64) @ self.db.astype(np.float64).T
        k = final_num_neighbors
        idx = np.argsort(-scores, axis=1, kind="stable")[:, :k]
        return idx.astype(np.int32), np.take_along_axis(scores, idx, axis=1).astype(np.float32)


class _Builder:
    def __init__(self, db, k, metric):
        self.db = db

    def tree(self, **kw):
        return self

    def score_ah(self, *a, **kw):
        return self

    def reorder(self, *a, **kw):
        return self

    def build(self):
        return _ExactSearcher(self.db)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__file__ = f"<shim {name}>"
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    """Install the stubs and put the reference on sys.path.  Idempotent."""
    if getattr(install, "_done", False):
        return
    # NumPy-2 removed aliases that recbole/config/configurator.py assigns into yaml loaders.
    for alias, real in (("float_", np.float64), ("complex_", np.complex128), ("unicode_", np.str_)):
        if not hasattr(np, alias):
            setattr(np, alias, real)

    class ColoredFormatter(logging.Formatter):
        def __init__(self, fmt=None, datefmt=None, log_colors=None, **kw):
            super().__init__((fmt or "").replace("%(log_color)s", ""), datefmt)

    _mod("colorlog", ColoredFormatter=ColoredFormatter)
    _mod("colorama", init=lambda *a, **k: None)

    class Texttable:
        def __init__(self, *a, **k):
            self.rows = []

        def set_cols_align(self, *a):
            pass

        def set_cols_valign(self, *a):
            pass

        def set_cols_dtype(self, *a):
            pass

        def add_rows(self, rows):
            self.rows = rows

        def draw(self):
            return "\n".join(str(r) for r in self.rows)

    _mod("texttable", Texttable=Texttable)
    _mod("thop", profile=lambda *a, **k: (0, 0))

    class SummaryWriter:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return lambda *a, **k: None

    import torch.utils  # noqa: F401
    tb = _mod("torch.utils.tensorboard", SummaryWriter=SummaryWriter)
    import torch
    torch.utils.tensorboard = tb

    class StorageBase:
        pass

    class InMemoryStorage(StorageBase):
        def __init__(self, *a, **k):
            pass

    storage = _mod("pyLSHash.storage", StorageBase=StorageBase, InMemoryStorage=InMemoryStorage)
    _mod("pyLSHash", storage=storage)
    _mod("csiphash", siphash24=siphash24_py)
    pybind = _mod("scann.scann_ops_pybind", builder=lambda db, k, metric: _Builder(db, k, metric))
    _mod("scann", scann_ops_pybind=pybind)
    _mod("wandb")
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    install._done = True
