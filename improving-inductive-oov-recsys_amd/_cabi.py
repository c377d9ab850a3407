"""ctypes binding of libmi_oov.so (include/mi_oov.h) -- the ONLY compute backend of this package.

There is no CPU or eager-PyTorch fallback: if the shared library is missing, or a tensor is not
on a ROCm device, the call raises.  PyTorch is used for device memory and streams only; every
argument crosses the boundary as a raw device pointer / size (no torch types in the C ABI).
"""
import ctypes
import os

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "lib", "libmi_oov.so")

_i64 = ctypes.c_int64
_vp = ctypes.c_void_p


class MiOovError(RuntimeError):
    """A libmi_oov entry point returned a negative status."""


_lib = None

_SIGNATURES = {
    "mi_oov_version": (ctypes.c_int, []),
    "mi_oov_init": (ctypes.c_int, []),
    "mi_oov_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "mi_oov_last_hip_error": (ctypes.c_int, []),
    "mi_oov_lsh_embed": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    "mi_oov_lsh_backward_workspace": (_i64, [_i64, _i64, _i64]),
    "mi_oov_lsh_embed_backward": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "mi_oov_lsh_backward_fused_workspace": (_i64, [_i64, _i64, _i64]),
    "mi_oov_lsh_backward_fused_counters": (_i64, []),
    "mi_oov_lsh_embed_backward_fused": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "mi_oov_slsh_embed_backward_fused": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "mi_oov_score_topk_excl_workspace": (_i64, [_i64, _i64, _i64, _i64]),
    "mi_oov_score_topk_excl": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "mi_oov_score_topk_masked_workspace": (_i64, [_i64, _i64, _i64, _i64]),
    "mi_oov_score_topk_masked": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mi_oov_topk_catalogue_bytes": (_i64, [_i64, _i64]),
    "mi_oov_topk_catalogue_prepare": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp]),
    "mi_oov_score_topk_prepared": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mi_oov_segment_topk": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp]),
    "mi_oov_topk_hits": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "mi_oov_topk_hits_range": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp, _i64, _i64, _vp, _vp]),
    "mi_oov_eval_rows_build": (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "mi_oov_segment_dedup": (ctypes.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "mi_oov_topk_metric_sums_workspace": (_i64, [_i64, _i64]),
    "mi_oov_topk_metric_sums": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _i64, ctypes.c_int, _vp, _vp, _vp, _vp]),
    "mi_oov_score_topk_excl_dense_workspace": (_i64, [_i64, _i64]),
    "mi_oov_score_topk_excl_dense": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mi_oov_slsh_embed_backward": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "mi_oov_scatter_add_rows": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i64, _vp, _vp]),
    "mi_oov_lsh_embed_score": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    "mi_oov_lsh_embed_score_multi": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp]),
    "mi_oov_lsh_table_bytes": (_i64, [_i64, _i64]),
    "mi_oov_lsh_table_prepare": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp]),
    "mi_oov_lsh_multi": (ctypes.c_int, [ctypes.c_int, _vp, _vp, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _i64,
                                        _vp, _vp]),
    "mi_oov_gather_rows_multi": (ctypes.c_int, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _vp]),
    "mi_oov_gather_mean_multi": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _vp]),
    "mi_oov_slsh_embed_multi": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _i64, _vp]),
    "mi_oov_bucket_by_owner": (ctypes.c_int, [_vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "mi_oov_bucket_by_owner_scratch": (_i64, []),
    "mi_oov_bucket_by_owner_fused": (ctypes.c_int, [_vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mi_oov_lsh_codes_embed": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    "mi_oov_lsh_lookup": (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _vp]),
    "mi_oov_lsh_lookup_score": (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _vp,
                                               _vp, _vp]),
    "mi_oov_slsh_embed": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _i64, _vp, _vp, _vp]),
    "mi_oov_siphash24_mod": (ctypes.c_int, [_vp, _i64, _vp, _i64, ctypes.c_uint32, _vp, _vp]),
    "mi_oov_siphash24_mod_ld": (ctypes.c_int, [_vp, _i64, _vp, _i64, ctypes.c_uint32, _vp, _i64, _vp]),
    "mi_oov_mapper_hash": (ctypes.c_int, [_vp, _i64, ctypes.c_int, _vp, _vp]),
    "mi_oov_mapper_map": (ctypes.c_int, [_vp, _i64, ctypes.c_int, _i64, _i64, _vp, _vp]),
    "mi_oov_gather_mean": (ctypes.c_int, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp]),
    "mi_oov_gather_rows": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i64, _vp, _vp]),
    "mi_oov_splice_rows": (ctypes.c_int, [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _vp, _vp]),
    "mi_oov_token_fields_embed": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _i64, _vp,
                                                 _vp, _i64, ctypes.c_int, _vp, _vp]),
    "mi_oov_col_mean_workspace": (_i64, [_i64, _i64]),
    "mi_oov_col_mean": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp, _vp]),
    "mi_oov_broadcast_rows": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp]),
    "mi_oov_rowdot": (ctypes.c_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "mi_oov_full_sort_scores": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i64, _vp, _vp]),
    "mi_oov_linear_act": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp, _i64, ctypes.c_int, _vp, _vp]),
    "mi_oov_linear_x3_weights_bytes": (_i64, [_i64, _i64]),
    "mi_oov_linear_x3_prepare": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp]),
    "mi_oov_linear_x3_prepare_t": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp]),
    "mi_oov_linear_x3": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp, _i64, ctypes.c_int, _vp, _vp]),
    "mi_oov_linear_x3_splitk_workspace": (_i64, [_i64, _i64, _i64]),
    "mi_oov_linear_x3_splitk": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp, _i64, ctypes.c_int, _vp, _i64, _vp, _vp]),
    "mi_oov_act_forward": (ctypes.c_int, [_vp, _i64, ctypes.c_int, _vp, _vp]),
    "mi_oov_act_backward": (ctypes.c_int, [_vp, _vp, _i64, ctypes.c_int, _vp, _vp]),
    "mi_oov_transpose": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp]),
    "mi_oov_score_topk_workspace": (_i64, [_i64, _i64, _i64]),
    "mi_oov_score_topk_workspace_d": (_i64, [_i64, _i64, _i64, _i64]),
    "mi_oov_score_topk_prepared_workspace": (_i64, [_i64, _i64, _i64, _i64, ctypes.c_int]),
    "mi_oov_score_topk": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
}

EXPORTS = tuple(_SIGNATURES)


def available() -> bool:
    return os.path.exists(LIB_PATH)


def lib():
    """Load libmi_oov.so once.  torch is imported first so that the HIP runtime torch already
    mapped (same SONAME libamdhip64.so.7) is the one our library binds to: streams and device
    pointers are then shared between torch and the kernels."""
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
                "improving-inductive-oov-recsys_amd/csrc`). There is no CPU fallback.")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        l = lib()
        msg = l.mi_oov_strerror(rc).decode()
        raise MiOovError(f"{what}: {msg} (code {rc}, hipError {l.mi_oov_last_hip_error()})")


def ptr(t):
    return None if t is None else t.data_ptr()


def dev_tensor(t, dtype, name):
    """Validate a tensor that is handed to a kernel: ROCm device, dtype, dense row-major."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: mi_oov kernels run on an MI355X (ROCm) device only; "
                           "there is no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        t = t.contiguous()
    return t


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_of(t) -> int:
    """hipStream_t of torch's current stream on the tensor's device, as an int (raw handle)."""
    idx = t.device.index
    if _raw_stream is not None and idx is not None:
        return _raw_stream(idx)
    return torch.cuda.current_stream(t.device).cuda_stream


def current_device() -> int:
    return _get_device() if _get_device is not None else torch.cuda.current_device()


def raw_stream(idx: int) -> int:
    """hipStream_t (raw handle) of torch's current stream on device `idx`."""
    if _raw_stream is not None:
        return _raw_stream(idx)
    return torch.cuda.current_stream(idx).cuda_stream


class BoundCall:
    """A C-ABI call whose arguments were validated and converted to ctypes values ONCE (a "call descriptor"): invoking
    it costs the foreign call itself (~1 us of Python + the ~3.5 us hipLaunchKernel inside the library) instead of
    the ~8 us of a fully checked wrapper -- what a serving loop that rotates over preallocated buffers uses when it
    cannot capture a HIP graph.  The stream is the one current on `device_index` at CALL time (last argument).

        call = C.BoundCall("mi_oov_lsh_embed_score", idx, ids.data_ptr(), B, ...)   # without the stream argument
        call()                    # launch; raises MiOovError on a negative status
        call.rebind(0, other_ids.data_ptr())     # swap one pointer / size in place

    The caller keeps the tensors alive and on `device_index`; nothing is re-checked per call."""

    __slots__ = ("name", "_fn", "_args", "_types", "_idx")

    def __init__(self, name, device_index, *args):
        l = lib()
        self.name, self._fn, self._idx = name, getattr(l, name), device_index
        self._types = _SIGNATURES[name][1][:-1]
        if len(args) != len(self._types):
            raise TypeError(f"{name} takes {len(self._types)} arguments before the stream, got {len(args)}")
        self._args = [t(a) for t, a in zip(self._types, args)]

    def rebind(self, i, value):
        self._args[i] = self._types[i](value)

    def __call__(self):
        rc = self._fn(*self._args, _raw_stream(self._idx))
        if rc:
            check(rc, self.name)


class on_device:
    """Make the tensor's device current for the duration of a launch (no-op when it already is:
    one process per GPU is the deployment model, so the fast path is a single integer compare)."""

    __slots__ = ("idx", "prev")

    def __init__(self, t):
        self.idx = t.device.index
        self.prev = None

    def __enter__(self):
        cur = _get_device() if _get_device is not None else torch.cuda.current_device()
        if self.idx is not None and cur != self.idx:
            self.prev = cur
            torch.cuda.set_device(self.idx)

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)
        return False
