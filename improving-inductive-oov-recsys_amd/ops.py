"""Torch-facing operators over the C ABI (include/mi_oov.h).  One function per entry point.

Forward passes are the hand-written HIP kernels.  Where the reference's outputs carry gradients
(SURVEY.md section 8b "Autograd": bucket tables for lsh/slsh, the item table for knn, the MLP weights of
dhe/fdhe/dnn, both sides for the scores) a torch.autograd.Function supplies the backward, and every backward
is itself made of this library's kernels (deterministic bucket reductions, mi_oov_scatter_add_rows, the f32-MFMA
GEMM + mi_oov_transpose + mi_oov_act_backward for the hash nets and the full-sort scores): torch autograd links
the pieces and computes nothing.  Nothing here ever computes on the CPU.
"""
import os
import weakref

import torch

from . import _cabi as C

HASH_KINDS = {"mod": 0, "fast": 1, "3round": 2, "64bit": 3}


def _ids(t, name="ids"):
    return C.dev_tensor(t, torch.int64, name)


def _f32(t, name):
    return C.dev_tensor(t.detach() if t.requires_grad else t, torch.float32, name)


# ------------------------------------------------------------------------------------------
# raw forwards
# ------------------------------------------------------------------------------------------
def _lsh_forward(ids, feat, planes, buckets, want_out=True, want_bits=False, bits_out=None):
    ids, feat, planes = _ids(ids), _f32(feat, "feat"), _f32(planes, "planes")
    B, (N, F), H = ids.numel(), feat.shape, planes.shape[0]
    if planes.shape[1] != F:
        raise ValueError(f"planes have {planes.shape[1]} columns, features have {F}")
    out = bits = None
    D = 0
    if want_out:
        buckets = _f32(buckets, "buckets")
        if buckets.shape[0] != H:
            raise ValueError(f"lsh needs one bucket row per plane: {buckets.shape[0]} vs {H}")
        D = buckets.shape[1]
        out = torch.empty((B, D), dtype=torch.float32, device=ids.device)
    if want_bits and bits_out is not None:
        if bits_out.dtype != torch.uint8 or not bits_out.is_contiguous() or bits_out.numel() != B * H or bits_out.device != ids.device:
            raise ValueError(f"out must be a contiguous u8[{B}, {H}] tensor on {ids.device}")
        bits = bits_out
    elif want_bits:
        bits = torch.empty((B, H), dtype=torch.uint8, device=ids.device)
    with C.on_device(ids):
        rc = C.lib().mi_oov_lsh_embed(C.ptr(ids), B, C.ptr(feat), N, F, C.ptr(planes), H,
                                      C.ptr(buckets) if want_out else None, D, C.ptr(out), C.ptr(bits),
                                      C.stream_of(ids))
    C.check(rc, "mi_oov_lsh_embed")
    return out, bits


def lsh_bits(ids, feat, planes, out=None):
    """TorchLSHash.hash_points(planes, feat[ids]) as u8[B,H] (R/inductive/torch_hash.py:55-60); `out`: a contiguous u8[B,H]
    buffer (a view into a larger one: the sharded exchange appends its local share's codes to the exchanged ones)."""
    return _lsh_forward(ids, feat, planes, None, want_out=False, want_bits=True, bits_out=out)[1]


class _LshEmbed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, feat, planes, buckets):
        need_grad = buckets.requires_grad
        out, bits = _lsh_forward(ids, feat, planes, buckets, want_bits=need_grad)
        if need_grad:
            ctx.save_for_backward(bits)
        return out

    @staticmethod
    def backward(ctx, g):
        (bits,) = ctx.saved_tensors
        return None, None, None, lsh_embed_backward(bits, g)


_BWD_COUNTERS = {}  # (device index, raw stream) -> zeroed arrival counters: the fused backward kernels leave them at zero


def _bwd_counters(t):
    """The arrival counters of mi_oov_*_embed_backward_fused for the current stream of t's device: zero at first use, left
    zero by every launch, one pair per stream (launches on one stream are ordered; on two streams they may overlap).
    While a HIP graph is being captured a fresh zeroed pair is made inside the capture instead (the cached pair must not
    end up in a graph's private pool)."""
    n = int(C.lib().mi_oov_lsh_backward_fused_counters())
    if torch.cuda.is_current_stream_capturing():
        return torch.zeros((n,), dtype=torch.int32, device=t.device)
    key = (t.device.index, C.stream_of(t))
    c = _BWD_COUNTERS.get(key)
    if c is None:
        c = _BWD_COUNTERS[key] = torch.zeros((n,), dtype=torch.int32, device=t.device)
    return c


def _bwd_mode():
    """MI_OOV_BWD_FUSED (developer A/B knob): 2 (default) = every plane's partials in one launch + the coalesced final launch;
    1 = ONE launch (the last workgroups to arrive finish; needs the zeroed counters); 0 = round 3's launches (eight planes
    per pass over the batch, one wave per column in the final one).  Bit-identical results."""
    v = os.environ.get("MI_OOV_BWD_FUSED", "2")
    return int(v) if v in ("0", "1", "2") else 2


def lsh_embed_backward(bits, grad_out):
    """d/dW of (bits @ W) / bits.sum(1): grad_W = bits^T @ (g / popcount); a popcount-0 row gives NaN like the
    reference's autograd.  Deterministic; ONE launch (mi_oov_lsh_embed_backward_fused: partial sums per row partition,
    the last workgroups to arrive add them up) -- bit-identical to the two-launch mi_oov_lsh_embed_backward."""
    bits = C.dev_tensor(bits, torch.uint8, "bits")
    g = _f32(grad_out, "grad_out")
    (B, H), D = bits.shape, g.shape[1]
    if g.shape[0] != B:
        raise ValueError(f"grad_out has {g.shape[0]} rows, bits {B}")
    lib = C.lib()
    if D > 256:
        # embedding rows wider than the 256 columns the reduction keeps per workgroup: one window of columns at a time (a
        # column's sum does not depend on the other columns, so the bits are those of a single pass)
        return torch.cat([lsh_embed_backward(bits, g[:, d0:d0 + 256].contiguous()) for d0 in range(0, D, 256)], dim=1)
    out = torch.empty((H, D), dtype=torch.float32, device=g.device)
    if _bwd_mode():
        ws = torch.empty((max(int(lib.mi_oov_lsh_backward_fused_workspace(B, H, D)), 1),), dtype=torch.float32, device=g.device)
        cnt = _bwd_counters(g) if _bwd_mode() == 1 else None
        with C.on_device(g):
            rc = lib.mi_oov_lsh_embed_backward_fused(C.ptr(bits), C.ptr(g), B, H, D, C.ptr(out), C.ptr(ws), C.ptr(cnt), C.stream_of(g))
        C.check(rc, "mi_oov_lsh_embed_backward_fused")
        return out
    ws = torch.empty((max(int(lib.mi_oov_lsh_backward_workspace(B, H, D)), 1),), dtype=torch.float32, device=g.device)
    with C.on_device(g):
        rc = lib.mi_oov_lsh_embed_backward(C.ptr(bits), C.ptr(g), B, H, D, C.ptr(out), C.ptr(ws), C.stream_of(g))
    C.check(rc, "mi_oov_lsh_embed_backward")
    return out


def lsh_embed(ids, feat, planes, buckets):
    """(bits @ buckets) / popcount for feat[ids] (R/inductive/lsh_embedder.py:116-179)."""
    if torch.is_grad_enabled() and buckets.requires_grad:
        return _LshEmbed.apply(ids, feat, planes, buckets)
    return _lsh_forward(ids, feat, planes, buckets)[0]


def lsh_embed_score(ids, feat, planes, buckets, other, want_emb=False, score_out=None):
    """Fused lsh embedding + BPR.predict row dot (bpr.py:145-149).  Inference only.
    `score_out` (f32[B], optional) receives the scores instead of a fresh tensor (serving loops, graph capture)."""
    ids, feat, planes, buckets, other = (_ids(ids), _f32(feat, "feat"), _f32(planes, "planes"),
                                         _f32(buckets, "buckets"), _f32(other, "other"))
    B, (N, F), H, D = ids.numel(), feat.shape, planes.shape[0], buckets.shape[1]
    if other.shape != (B, D):
        raise ValueError(f"other must be [{B},{D}], got {tuple(other.shape)}")
    if score_out is None:
        score = torch.empty((B,), dtype=torch.float32, device=ids.device)
    else:
        score = C.dev_tensor(score_out, torch.float32, "score_out")
        if score.shape != (B,) or score.data_ptr() != score_out.data_ptr():
            raise ValueError(f"score_out must be a contiguous f32[{B}] tensor on the device")
    if D > 256:
        # rows wider than the 256 floats one launch keeps per lookup: the library writes them a window of columns at a
        # time (csrc/lsh.hip, dispatch_lsh) and the row dot is a launch of its own -- the same values in the same order
        out = lsh_embed(ids, feat, planes, buckets)
        score.copy_(_rowdot_forward(other, out))
        return (score, out) if want_emb else score
    out = torch.empty((B, D), dtype=torch.float32, device=ids.device) if want_emb else None
    with C.on_device(ids):
        rc = C.lib().mi_oov_lsh_embed_score(C.ptr(ids), B, C.ptr(feat), N, F, C.ptr(planes), H, C.ptr(buckets), D,
                                            C.ptr(other), C.ptr(score), C.ptr(out), C.stream_of(ids))
    C.check(rc, "mi_oov_lsh_embed_score")
    return (score, out) if want_emb else score


class LshScorer:
    """Serving-loop form of lsh_embed_score: the static operands (feature table, planes, bucket table) are
    validated ONCE and their device pointers kept; a call then costs one ctypes launch (~3 us of host time
    instead of ~8.6 us for the fully checked wrapper, tools/hostov.py).  Same kernel, same results.

        scorer = ops.LshScorer(feat, planes, buckets)
        scores = scorer(ids, user_rows)                       # f32[B]
        scorer(ids, user_rows, score_out=buf)                 # into a caller-owned buffer (graph capture)

    The tensors are held by the object; if a bucket table is updated IN PLACE (an optimizer step) the next
    call sees the new values, if it is re-allocated build a new scorer."""

    __slots__ = ("feat", "planes", "buckets", "_fn", "_static", "N", "F", "H", "D", "device", "_idx")

    def __init__(self, feat, planes, buckets):
        self.feat, self.planes, self.buckets = _f32(feat, "feat"), _f32(planes, "planes"), _f32(buckets, "buckets")
        (self.N, self.F), self.H, self.D = self.feat.shape, self.planes.shape[0], self.buckets.shape[1]
        if self.planes.shape[1] != self.F:
            raise ValueError(f"planes have {self.planes.shape[1]} columns, features have {self.F}")
        if self.buckets.shape[0] != self.H:
            raise ValueError(f"lsh needs one bucket row per plane: {self.buckets.shape[0]} vs {self.H}")
        self.device = self.feat.device
        self._idx = self.device.index
        self._fn = C.lib().mi_oov_lsh_embed_score

    def __call__(self, ids, other, score_out=None):
        B = ids.numel()
        if (ids.dtype is not torch.int64 or other.dtype is not torch.float32 or ids.device != self.device
                or other.device != self.device or not ids.is_contiguous() or not other.is_contiguous()
                or other.shape != (B, self.D)):
            raise ValueError(f"LshScorer needs contiguous int64[B] ids and f32[B,{self.D}] rows on {self.device}")
        if score_out is None:
            score_out = torch.empty((B,), dtype=torch.float32, device=self.device)
        elif (score_out.dtype is not torch.float32 or score_out.device != self.device or score_out.shape != (B,)
              or not score_out.is_contiguous()):
            raise ValueError(f"score_out must be a contiguous f32[{B}] tensor on {self.device}")
        if C.current_device() != self._idx:
            with C.on_device(ids):
                rc = self._call(ids, B, other, score_out)
        else:
            rc = self._call(ids, B, other, score_out)
        if rc:
            C.check(rc, "mi_oov_lsh_embed_score")
        return score_out

    def _call(self, ids, B, other, score_out):
        return self._fn(ids.data_ptr(), B, self.feat.data_ptr(), self.N, self.F, self.planes.data_ptr(), self.H,
                        self.buckets.data_ptr(), self.D, other.data_ptr(), score_out.data_ptr(), None,
                        C.raw_stream(self._idx))

    def bind(self, ids, other, score_out):
        """A prevalidated launch for ONE (ids, other, score_out) triple of preallocated buffers: `launch = scorer.bind(...)`
        checks them now; `launch()` then costs ~1 us of Python on top of the hipLaunchKernel inside the library, so a
        loop that cannot be captured in a HIP graph stays bound by the 8.5 us kernel instead of by the host.  The caller
        keeps the three tensors alive and unmoved; their CONTENTS may change between launches."""
        B = ids.numel()
        if (ids.dtype is not torch.int64 or other.dtype is not torch.float32 or ids.device != self.device
                or other.device != self.device or not ids.is_contiguous() or not other.is_contiguous()
                or other.shape != (B, self.D)):
            raise ValueError(f"LshScorer needs contiguous int64[B] ids and f32[B,{self.D}] rows on {self.device}")
        if (score_out.dtype is not torch.float32 or score_out.device != self.device or score_out.shape != (B,)
                or not score_out.is_contiguous()):
            raise ValueError(f"score_out must be a contiguous f32[{B}] tensor on {self.device}")
        return C.BoundCall("mi_oov_lsh_embed_score", self._idx, ids.data_ptr(), B, self.feat.data_ptr(), self.N, self.F,
                           self.planes.data_ptr(), self.H, self.buckets.data_ptr(), self.D, other.data_ptr(),
                           score_out.data_ptr(), None)


class LshTable:
    """The 2^H-row table of aggregates `(bits @ buckets) / popcount` for ONE bucket tensor, prepared once
    (`mi_oov_lsh_table_prepare`) and handed to the persistent launches, which then load it instead of rebuilding it at
    their head.  `get()` returns the device table, re-prepared whenever the bucket tensor was written to since (torch's
    version counter: an optimizer step, `load_state_dict` and every other in-place operation on the tensor bump it) or
    points at other memory (`p.data = new`); an in-place write THROUGH `.data` (`w.data.normal_()`) moves neither --
    call `invalidate()` after one (this package's own initialiser writes under `torch.no_grad()`, which does bump the
    counter).  None for shapes the persistent kernel does not take (D != 64 or more than 8 buckets): callers pass that on
    and the launch builds the table itself."""

    __slots__ = ("buckets", "table", "version", "addr")

    def __init__(self, buckets):
        self.buckets = buckets  # the caller's tensor (a Parameter keeps its identity across optimizer steps)
        self.table, self.version, self.addr = None, None, None

    def invalidate(self):
        self.version = None

    def get(self):
        b = self.buckets
        H, D = b.shape
        nbytes = int(C.lib().mi_oov_lsh_table_bytes(H, D))
        if nbytes <= 0 or not b.is_cuda or b.data_ptr() % 16 or not b.is_contiguous():
            return None
        # stale when the tensor was written in place (version counter), re-pointed (`p.data = other`: same Parameter, same
        # version, another address) or moved to another device
        if self.table is None or self.version != b._version or self.addr != b.data_ptr() or self.table.device != b.device:
            if self.table is None or self.table.device != b.device:
                self.table = torch.empty((1 << H, D), dtype=torch.float32, device=b.device)
            src = _f32(b, "buckets")
            with C.on_device(src):
                rc = C.lib().mi_oov_lsh_table_prepare(C.ptr(src), H, D, C.ptr(self.table), C.stream_of(src))
            C.check(rc, "mi_oov_lsh_table_prepare")
            self.version, self.addr = b._version, b.data_ptr()
        return self.table


class LshBatchQueue:
    """K queued batches for the persistent multi-batch launch (mi_oov_lsh_multi, csrc/lsh64p.hip).

    The kernel reads device arrays of K pointers (ids, rows of the other side, outputs).  This object validates the K
    tensors of each kind ONCE, builds those arrays, and keeps every tensor alive until it is dropped:

        q = ops.LshBatchQueue(ids_list, other_list)              # SCORES: f32[B] per batch, allocated here ...
        q = ops.LshBatchQueue(ids_list, other_list, score_list)  # ... or caller-owned buffers
        q = ops.LshBatchQueue(ids_list, rows=True)               # ROWS: f32[B,D] per batch (what embed_*_ids returns)
        q = ops.LshBatchQueue(ids_list, out_list=bufs, rows=True)
        scorer.run(q)              # all K batches, one launch
        scorer.run(q, 5, 20)       # batches 5 .. 24 of the queue (a slice costs nothing: pointer arithmetic)

    Every batch of a queue has the same B; D is the embedder's width."""

    __slots__ = ("ids", "other", "scores", "K", "B", "D", "device", "tab", "rows")

    def __init__(self, ids_list, other_list=None, score_list=None, rows=False, out_list=None, D=64):
        if out_list is not None:
            score_list = out_list
        K = len(ids_list)
        if K == 0 or (not rows and (other_list is None or len(other_list) != K)) or \
                (score_list is not None and len(score_list) != K):
            raise ValueError("LshBatchQueue needs K >= 1 id tensors and as many row (and output) tensors")
        dev, B = ids_list[0].device, ids_list[0].numel()
        if not rows:
            D = other_list[0].shape[-1]
        if score_list is None:
            block = torch.empty((K, B, D) if rows else (K, B), dtype=torch.float32, device=dev)
            score_list = [block[k] for k in range(K)]
        for k in range(K):
            i, s = ids_list[k], score_list[k]
            o = None if rows else other_list[k]
            for t, dt, nm in ((i, torch.int64, "ids"), (o, torch.float32, "other"), (s, torch.float32, "out")):
                if t is None:
                    continue
                C.dev_tensor(t, dt, f"{nm}[{k}]")
                if not t.is_contiguous() or t.device != dev:
                    raise ValueError(f"{nm}[{k}] must be contiguous and on {dev}")
            if i.numel() != B or (o is not None and o.shape != (B, D)) or s.shape != ((B, D) if rows else (B,)):
                raise ValueError(f"batch {k}: need int64[{B}] ids" + ("" if rows else f", f32[{B},{D}] rows") +
                                 (f", f32[{B},{D}] output rows" if rows else f", f32[{B}] scores"))
            if D % 4 == 0 and ((o is not None and o.data_ptr() % 16) or (rows and s.data_ptr() % 16)):
                raise ValueError(f"rows of batch {k} are not 16-byte aligned")  # (D % 4 != 0: the scalar kernels, any address)
        self.ids, self.other, self.scores = list(ids_list), (None if rows else list(other_list)), list(score_list)
        self.K, self.B, self.D, self.device, self.rows = K, B, D, dev, bool(rows)
        self.tab = _ptr_table(self.ids, self.other, self.scores)  # [3, K] device pointers

    @property
    def outputs(self):
        return self.scores


class LshMultiScorer:
    """The four lsh per-batch calls for K batches in ONE persistent launch (mi_oov_lsh_multi): scores or rows, with or
    without the in-vocabulary table of BPR's lookups.  Static operands validated once (as LshScorer); the table of
    aggregates is prepared once per bucket-table version (LshTable) unless `prepared=False`.
    Shapes the persistent kernel does not serve (F or D != 64, H > 8) run as K single launches: same results.

        ops.LshMultiScorer(feat, planes, buckets).run(q)                  # K x lsh_embed_score  (q: scores queue)
        ops.LshMultiScorer(feat, planes, buckets).run(q_rows)             # K x lsh_embed        (q: rows queue)
        ops.LshMultiScorer(feat, planes, buckets, vtable=item_table).run(q)    # K x lsh_lookup_score / lsh_lookup"""

    __slots__ = ("feat", "planes", "buckets", "vtable", "N", "F", "H", "D", "device", "_idx", "_fn", "persistent", "_table")

    def __init__(self, feat, planes, buckets, vtable=None, prepared=True):
        self.feat, self.planes, self.buckets = _f32(feat, "feat"), _f32(planes, "planes"), _f32(buckets, "buckets")
        (self.N, self.F), self.H, self.D = self.feat.shape, self.planes.shape[0], self.buckets.shape[1]
        if self.planes.shape[1] != self.F:
            raise ValueError(f"planes have {self.planes.shape[1]} columns, features have {self.F}")
        if self.buckets.shape[0] != self.H:
            raise ValueError(f"lsh needs one bucket row per plane: {self.buckets.shape[0]} vs {self.H}")
        self.vtable = None if vtable is None else _f32(vtable, "vtable")
        if self.vtable is not None and self.vtable.shape[1] != self.D:
            raise ValueError(f"the in-vocabulary table has {self.vtable.shape[1]} columns, the bucket table {self.D}")
        self.device = self.feat.device
        self._idx = self.device.index
        self._fn = C.lib().mi_oov_lsh_multi
        self.persistent = self.F == 64 and self.D == 64 and 1 <= self.H <= 8
        if self.persistent:
            with C.on_device(self.feat):
                C.lib().mi_oov_init()  # the library's one allocation (128 pinned bytes per device), made here, not by a launch
        self._table = LshTable(buckets) if (prepared and self.persistent) else None

    def run(self, q, k0=0, k=None):
        """Batches [k0, k0 + k) of the queue (default: all).  Returns the list of output tensors (scores or rows)."""
        k = q.K - k0 if k is None else k
        if k0 < 0 or k < 0 or k0 + k > q.K:
            raise ValueError(f"batches [{k0}, {k0 + k}) are not inside a queue of {q.K}")
        if q.device != self.device or q.D != self.D:
            raise ValueError(f"queue is for {q.device}, D = {q.D}; scorer for {self.device}, D = {self.D}")
        if k == 0:
            return []
        if not self.persistent or q.B > (1 << 23):
            for j in range(k0, k0 + k):
                self._single(q, j)
            return q.scores[k0:k0 + k]
        table = self._table.get() if self._table is not None else None
        base = q.tab.data_ptr()
        step = q.K * 8
        if C.current_device() != self._idx:
            with C.on_device(self.feat):
                rc = self._launch(q.rows, base, step, k0, k, q.B, table)
        else:
            rc = self._launch(q.rows, base, step, k0, k, q.B, table)
        if rc:
            C.check(rc, "mi_oov_lsh_multi")
        return q.scores[k0:k0 + k]

    def bind(self, q, k0=0, k=None):
        """A prevalidated launch of batches [k0, k0 + k) of the queue (`_cabi.BoundCall`: every argument converted once):
        `launch = scorer.bind(q, 0, 20)`, then `launch()` costs ~1 us of Python on top of the hipLaunchKernel inside the
        library instead of ~6 us -- for a loop that issues ONE persistent launch from an idle stream that difference is
        time the GPU waits.  The prepared table is taken NOW: re-bind after the bucket table has been updated.  Queues
        the persistent kernel does not serve cannot be bound (use run)."""
        k = q.K - k0 if k is None else k
        if k0 < 0 or k <= 0 or k0 + k > q.K:
            raise ValueError(f"batches [{k0}, {k0 + k}) are not inside a queue of {q.K}")
        if q.device != self.device or q.D != self.D:
            raise ValueError(f"queue is for {q.device}, D = {q.D}; scorer for {self.device}, D = {self.D}")
        if not self.persistent or q.B > (1 << 23):
            raise ValueError("only queues of the persistent kernel's shape (F = D = 64, H <= 8, B <= 2^23) can be bound")
        table = self._table.get() if self._table is not None else None
        base, step, off = q.tab.data_ptr(), q.K * 8, k0 * 8
        vt, nv = (self.vtable.data_ptr(), self.vtable.shape[0]) if self.vtable is not None else (None, 0)
        return C.BoundCall("mi_oov_lsh_multi", self._idx, 1 if q.rows else 0, base + off, None if q.rows else base + step + off,
                           base + 2 * step + off, k, q.B, vt, nv, self.feat.data_ptr(), self.N, self.F, self.planes.data_ptr(), self.H,
                           self.buckets.data_ptr(), self.D, None if table is None else table.data_ptr())

    def _single(self, q, j):
        lib, ids, out = C.lib(), q.ids[j], q.scores[j]
        if not q.rows and self.D > 256:
            # the fused score entries take rows of up to 256 floats (one launch holds the whole row): wider rows are written
            # a window of columns at a time and the row dot is a launch of its own -- what lsh_embed_score does
            emb = (lsh_embed(ids, self.feat, self.planes, self.buckets) if self.vtable is None
                   else lsh_lookup(ids, self.vtable, self.feat, self.planes, self.buckets))
            out.copy_(_rowdot_forward(q.other[j], emb))
            return
        vt, nv = (self.vtable.data_ptr(), self.vtable.shape[0]) if self.vtable is not None else (None, 0)
        with C.on_device(ids):
            st = C.stream_of(ids)
            if q.rows and vt is None:
                rc = lib.mi_oov_lsh_embed(ids.data_ptr(), q.B, self.feat.data_ptr(), self.N, self.F, self.planes.data_ptr(), self.H,
                                          self.buckets.data_ptr(), self.D, out.data_ptr(), None, st)
            elif q.rows:
                rc = lib.mi_oov_lsh_lookup(ids.data_ptr(), q.B, vt, nv, self.feat.data_ptr(), self.N, self.F, self.planes.data_ptr(),
                                           self.H, self.buckets.data_ptr(), self.D, out.data_ptr(), st)
            elif vt is None:
                rc = lib.mi_oov_lsh_embed_score(ids.data_ptr(), q.B, self.feat.data_ptr(), self.N, self.F, self.planes.data_ptr(),
                                                self.H, self.buckets.data_ptr(), self.D, q.other[j].data_ptr(), out.data_ptr(), None, st)
            else:
                rc = lib.mi_oov_lsh_lookup_score(ids.data_ptr(), q.B, vt, nv, self.feat.data_ptr(), self.N, self.F,
                                                 self.planes.data_ptr(), self.H, self.buckets.data_ptr(), self.D,
                                                 q.other[j].data_ptr(), out.data_ptr(), None, st)
        C.check(rc, "single-batch launch of a queue the persistent kernel does not serve")

    def _launch(self, rows, base, step, k0, k, B, table):
        off = k0 * 8
        vt, nv = (self.vtable.data_ptr(), self.vtable.shape[0]) if self.vtable is not None else (None, 0)
        return self._fn(1 if rows else 0, base + off, None if rows else base + step + off, base + 2 * step + off, k, B, vt, nv,
                        self.feat.data_ptr(), self.N, self.F, self.planes.data_ptr(), self.H, self.buckets.data_ptr(), self.D,
                        None if table is None else table.data_ptr(), C.raw_stream(self._idx))


def lsh_embed_score_multi(ids_list, feat, planes, buckets, other_list, score_out=None):
    """[lsh_embed_score(ids, feat, planes, buckets, other) for ids, other in zip(ids_list, other_list)] in one
    persistent launch (every batch the same size).  Returns the list of f32[B] score tensors.  Inference only."""
    q = LshBatchQueue(ids_list, other_list, score_out)
    # (the pointer table may be released as soon as the launch is enqueued: torch's caching allocator hands a freed
    # block only to work that is ordered behind the launch on the same stream)
    return LshMultiScorer(feat, planes, buckets, prepared=False).run(q)


def lsh_embed_multi(ids_list, feat, planes, buckets, out=None, table=None):
    """[lsh_embed(ids, feat, planes, buckets) for ids in ids_list] in one persistent launch: the [B, D] rows
    LSHInductiveEmbedder.embed_*_ids returns (lsh_embedder.py:141-179), K batches of the same size.  Inference only."""
    q = LshBatchQueue(ids_list, rows=True, out_list=out, D=buckets.shape[1])
    sc = LshMultiScorer(feat, planes, buckets, prepared=False)
    sc._table = table
    return sc.run(q)


def lsh_lookup_multi(ids_list, table, feat, planes, buckets, other_list=None, out=None, lsh_table=None):
    """K batches of lsh_lookup (other_list None: the [B, D] rows of BPR.get_*_embedding, bpr.py:48-125) or of
    lsh_lookup_score (other_list given: BPR.predict's scores, bpr.py:145-149) in one persistent launch."""
    rows = other_list is None
    q = LshBatchQueue(ids_list, other_list, out, rows=rows, D=buckets.shape[1])
    sc = LshMultiScorer(feat, planes, buckets, vtable=table, prepared=False)
    sc._table = lsh_table
    return sc.run(q)


def _batch_list(tensors, dtype, name, shape_tail=None):
    """K equally shaped, contiguous device tensors -> (list, rows per batch)."""
    if len(tensors) == 0:
        raise ValueError(f"{name}: need at least one batch")
    out = [C.dev_tensor(t, dtype, f"{name}[{k}]") for k, t in enumerate(tensors)]
    n0 = out[0].shape
    for k, t in enumerate(out):
        if t.shape != n0 or t.device != out[0].device:
            raise ValueError(f"{name}[{k}] has shape {tuple(t.shape)} on {t.device}; every queued batch must be {tuple(n0)} on {out[0].device}")
    return out


_PTR_TABLES = {}  # (device, addresses) -> device int64 table; a table holds nothing but the addresses that key it
_PTR_TABLES_MAX = 256


def _ptr_table(*lists):
    """device int64[len(lists), K] of data pointers.  Cached by the addresses themselves (a table is valid for whatever
    lives at those addresses), so a loop that rotates over preallocated buffers uploads each of its tables once: the
    host -> device copy of a fresh table costs more than the launch it feeds."""
    dev = next(t for lst in lists if lst is not None for t in lst).device
    K = len(next(lst for lst in lists if lst is not None))
    rows = tuple(tuple(t.data_ptr() for t in lst) if lst is not None else (0,) * K for lst in lists)
    key = (dev, rows)
    tab = _PTR_TABLES.get(key)
    if tab is None:
        if len(_PTR_TABLES) >= _PTR_TABLES_MAX:
            _PTR_TABLES.pop(next(iter(_PTR_TABLES)))
        tab = _PTR_TABLES[key] = torch.tensor(rows, dtype=torch.int64).to(dev)
    return tab


def _out_rows(out, K, rows, D, dev):
    """Caller-owned f32[rows, D] output buffers (validated), or K slices of one fresh block."""
    if out is None:
        block = torch.empty((K, rows, D), dtype=torch.float32, device=dev)
        return [block[k] for k in range(K)]
    outs = _batch_list(out, torch.float32, "out")
    if len(outs) != K or outs[0].shape != (rows, D) or any(o.data_ptr() != t.data_ptr() for o, t in zip(outs, out)) \
            or any(o.data_ptr() % 16 for o in outs):
        raise ValueError(f"out must be {K} contiguous, 16-byte aligned f32[{rows},{D}] tensors")
    return outs


def gather_rows_multi(ids_list, W, out=None):
    """[W[ids] for ids in ids_list] (nn.Embedding forward, bpr.py:77-81) for K equally sized batches in ONE launch
    (mi_oov_gather_rows_multi).  Inference only.  Widths that are not a multiple of 4 floats: K single launches."""
    W = _f32(W, "W")
    ids = _batch_list(ids_list, torch.int64, "ids")
    K, B, (N, D) = len(ids), ids[0].numel(), W.shape
    if D % 4 or W.data_ptr() % 16:
        return [_gather_rows_forward(i, W) for i in ids]
    outs = _out_rows(out, K, B, D, W.device)
    tab = _ptr_table(ids, outs)
    with C.on_device(W):
        rc = C.lib().mi_oov_gather_rows_multi(tab[0].data_ptr(), tab[1].data_ptr(), K, B, C.ptr(W), N, D, C.stream_of(W))
    C.check(rc, "mi_oov_gather_rows_multi")
    return outs


def gather_mean_multi(idx_list, W, g=2, out=None):
    """[gather_mean(idx, W, g) for idx in idx_list] (the knn aggregate, knn_embedder.py:125-126) for K equally sized
    batches in ONE launch (mi_oov_gather_mean_multi).  Inference only."""
    W = _f32(W, "W")
    idx = _batch_list([i.reshape(-1) for i in idx_list], torch.int64, "idx")
    K, M, (N, D) = len(idx), idx[0].numel(), W.shape
    if D % 4 or W.data_ptr() % 16:
        return [_gather_mean_forward(i, W, g) for i in idx]
    nout = (M + g - 1) // g
    outs = _out_rows(out, K, nout, D, W.device)
    tab = _ptr_table(idx, outs)
    with C.on_device(W):
        rc = C.lib().mi_oov_gather_mean_multi(tab[0].data_ptr(), tab[1].data_ptr(), K, M, g, C.ptr(W), N, D, C.stream_of(W))
    C.check(rc, "mi_oov_gather_mean_multi")
    return outs


def slsh_embed_multi(ids_list, feat, planes, buckets, want_idx=False, out=None):
    """[slsh_embed(ids, feat, planes, buckets) for ids in ids_list] (single_lsh_embedder.py:82-109) for K equally sized
    batches in ONE launch (mi_oov_slsh_embed_multi); shapes off the hot tile run as K single launches.  Inference only.
    want_idx: also the bucket ids -> (rows list, idx list)."""
    feat, planes, buckets = _f32(feat, "feat"), _f32(planes, "planes"), _f32(buckets, "buckets")
    ids = _batch_list(ids_list, torch.int64, "ids")
    K, B, (N, F), H, (nb, D) = len(ids), ids[0].numel(), feat.shape, planes.shape[0], buckets.shape
    hot = F == 64 and 1 <= H <= 32 and D in (64, 128) and not (feat.data_ptr() % 16 or planes.data_ptr() % 16 or buckets.data_ptr() % 16)
    if not hot or B > (1 << 22):
        res = [_slsh_forward(i, feat, planes, buckets, nb) for i in ids]
        return ([r[0] for r in res], [r[1] for r in res]) if want_idx else [r[0] for r in res]
    outs = _out_rows(out, K, B, D, feat.device)
    idxs = None
    if want_idx:
        iblock = torch.empty((K, B), dtype=torch.int64, device=feat.device)
        idxs = [iblock[k] for k in range(K)]
    tab = _ptr_table(ids, outs, idxs)
    with C.on_device(feat):
        rc = C.lib().mi_oov_slsh_embed_multi(tab[0].data_ptr(), tab[1].data_ptr(), tab[2].data_ptr() if want_idx else None, K, B,
                                             C.ptr(feat), N, F, C.ptr(planes), H, C.ptr(buckets), nb, D, C.stream_of(feat))
    C.check(rc, "mi_oov_slsh_embed_multi")
    return (outs, idxs) if want_idx else outs


_BUCKET_SCRATCH = {}  # (device index, raw stream) -> zeroed reservation words of mi_oov_bucket_by_owner_fused (left zero by every launch)


def _bucket_scratch(t):
    n = int(C.lib().mi_oov_bucket_by_owner_scratch())
    if torch.cuda.is_current_stream_capturing():  # (a cached tensor must not end up in a graph's private pool)
        return torch.zeros((n,), dtype=torch.int32, device=t.device)
    key = (t.device.index, C.stream_of(t))
    c = _BUCKET_SCRATCH.get(key)
    if c is None:
        c = _BUCKET_SCRATCH[key] = torch.zeros((n,), dtype=torch.int32, device=t.device)
    return c


def bucket_by_owner(ids, n_rows, rows_per_rank, world, cap, overflow=None, my_rank=None):
    """Requester side of a sharded lookup: -> (send int64[world, cap] of owner-local rows, -1 padded; slot int32[B];
    counts int32[world]).  `overflow` (int32 scalar tensor, optional) accumulates the largest excess of a segment over
    `cap`.  Device only, no host synchronisation.  One launch for a single-node world (mi_oov_bucket_by_owner_fused; the
    three-operation mi_oov_bucket_by_owner for world > 16 or under MI_OOV_BUCKET_FUSED=0).
    my_rank (world <= 16): the lookups this rank owns are COMPACTED instead of sent -- a fourth result local_rows int64[cap]
    (their local row numbers, -1 padded), their slots are world * cap + position, the send segment of my_rank is empty."""
    ids = _ids(ids)
    B = ids.numel()
    send = torch.empty((world, cap), dtype=torch.int64, device=ids.device)
    slot = torch.empty((B,), dtype=torch.int32, device=ids.device)
    counts = torch.empty((world,), dtype=torch.int32, device=ids.device)
    fused = world <= 16 and B > 0 and (my_rank is not None or os.environ.get("MI_OOV_BUCKET_FUSED", "1") != "0")
    if my_rank is not None and not fused:
        raise ValueError("bucket_by_owner(my_rank=...) needs 1 <= world <= 16 and a non-empty batch")
    with C.on_device(ids):
        if fused:
            local_rows = torch.empty((cap,), dtype=torch.int64, device=ids.device) if my_rank is not None else None
            rc = C.lib().mi_oov_bucket_by_owner_fused(C.ptr(ids), B, n_rows, rows_per_rank, world, cap, -1 if my_rank is None else int(my_rank),
                                                      C.ptr(send), C.ptr(slot), C.ptr(counts), C.ptr(overflow), C.ptr(local_rows),
                                                      C.ptr(_bucket_scratch(ids)), C.stream_of(ids))
            C.check(rc, "mi_oov_bucket_by_owner_fused")
            return (send, slot, counts, local_rows) if my_rank is not None else (send, slot, counts)
        rc = C.lib().mi_oov_bucket_by_owner(C.ptr(ids), B, n_rows, rows_per_rank, world, cap, C.ptr(send), C.ptr(slot),
                                            C.ptr(counts), C.ptr(overflow), C.stream_of(ids))
    C.check(rc, "mi_oov_bucket_by_owner")
    return send, slot, counts


def lsh_codes_embed(codes, slot, buckets, other=None, want_emb=True, score_out=None):
    """Requester side of a sharded lsh lookup (mi_oov_lsh_codes_embed): codes u8[M,H] as the owners returned them,
    slot int32[B] -> (score f32[B] or None, emb f32[B,D] or None), bit-identical to lsh_embed / lsh_embed_score."""
    codes = C.dev_tensor(codes, torch.uint8, "codes")
    slot = C.dev_tensor(slot, torch.int32, "slot")
    buckets = _f32(buckets, "buckets")
    codes = codes.view(-1, codes.shape[-1])
    (M, H), B, D = codes.shape, slot.numel(), buckets.shape[1]
    if buckets.shape[0] != H:
        raise ValueError(f"lsh needs one bucket row per plane: {buckets.shape[0]} vs {H}")
    score = None
    if other is not None:
        other = _f32(other, "other")
        if other.shape != (B, D):
            raise ValueError(f"other must be [{B},{D}], got {tuple(other.shape)}")
        if score_out is None:
            score = torch.empty((B,), dtype=torch.float32, device=codes.device)
        else:
            score = C.dev_tensor(score_out, torch.float32, "score_out")
            if score.shape != (B,) or score.data_ptr() != score_out.data_ptr():
                raise ValueError(f"score_out must be a contiguous f32[{B}] tensor on the device")
    out = torch.empty((B, D), dtype=torch.float32, device=codes.device) if (want_emb or other is None) else None
    with C.on_device(codes):
        rc = C.lib().mi_oov_lsh_codes_embed(C.ptr(codes), M, C.ptr(slot), B, H, C.ptr(buckets), D, C.ptr(other),
                                            C.ptr(score), C.ptr(out), C.stream_of(codes))
    C.check(rc, "mi_oov_lsh_codes_embed")
    return score, out


class _LshTrainLookup(torch.autograd.Function):
    """BPR.get_*_embedding with an lsh plugin UNDER AUTOGRAD (bpr.py:48-125 + lsh_embedder.py:133-179), without the
    reference's boolean-mask indexing (each `ids[mask]` is a device -> host sync) and its zeros / scatter / scatter
    splice: every row gets both the table gather and the lsh embedding of its feature row, one elementwise select
    keeps the right one, and the backward sends each gradient row to the table (in-vocabulary) or through the
    deterministic lsh backward (out-of-vocabulary).  `feat_ids` are the ids with the prime pad stripped."""

    @staticmethod
    def forward(ctx, ids, feat_ids, table, feat, planes, buckets):
        n_vocab = table.shape[0]
        oov = ids >= n_vocab
        emb, bits = _lsh_forward(feat_ids, feat, planes, buckets, want_bits=True)
        rows = _gather_rows_forward(ids, table)  # NaN rows where ids >= n_vocab: never selected
        out = torch.where(oov[:, None], emb, rows)
        # rows that are not lsh rows must not reach the lsh backward with a 0/0: give them a one-plane code
        first = (torch.arange(bits.shape[1], device=bits.device) == 0).to(bits.dtype)[None]  # (device ops only: the
        # step stays capturable in a HIP graph; `first[0, 0] = 1` is a host -> device copy)
        safe = torch.where(oov[:, None], bits, first)  # (no boolean-mask assignment: that would sync)
        ctx.save_for_backward(ids, oov, safe)
        ctx.n_vocab = n_vocab
        ctx.need = (table.requires_grad, buckets.requires_grad)
        return out

    @staticmethod
    def backward(ctx, g):
        ids, oov, bits = ctx.saved_tensors
        g = g.contiguous()
        gt = scatter_add_rows(ids, g, ctx.n_vocab) if ctx.need[0] else None  # ids >= n_vocab skipped by the kernel
        gb = lsh_embed_backward(bits, g * oov[:, None].to(g.dtype)) if ctx.need[1] else None
        return None, None, gt, None, None, gb


class _BucketTrainLookup(torch.autograd.Function):
    """The same sync-free select for plugins whose OOV row is ONE bucket row (slsh: single_lsh_embedder.py:100,108;
    the random mapper without embedder: bpr.py:75,122): out = buckets[idx] where the id is out of vocabulary, table[id]
    elsewhere; `idx` is -1 on in-vocabulary rows, which both backward kernels skip."""

    @staticmethod
    def forward(ctx, ids, idx, table, buckets):
        oov = ids >= table.shape[0]
        out = torch.where(oov[:, None], _gather_rows_forward(idx, buckets), _gather_rows_forward(ids, table))
        ctx.save_for_backward(ids, idx)
        ctx.shapes = (table.shape[0], buckets.shape[0])
        ctx.need = (table.requires_grad, buckets.requires_grad)
        return out

    @staticmethod
    def backward(ctx, g):
        ids, idx = ctx.saved_tensors
        g = g.contiguous()
        gt = scatter_add_rows(ids, g, ctx.shapes[0]) if ctx.need[0] else None
        gb = slsh_embed_backward(idx, g, ctx.shapes[1]) if ctx.need[1] else None
        return None, None, gt, gb


def bucket_train_lookup(ids, idx, table, buckets):
    return _BucketTrainLookup.apply(_ids(ids), _ids(idx, "idx"), table, buckets)


def lsh_train_lookup(ids, feat_ids, table, feat, planes, buckets):
    return _LshTrainLookup.apply(_ids(ids), _ids(feat_ids, "feat_ids"), table, feat, planes, buckets)


def lsh_lookup(ids, table, feat, planes, buckets):
    """BPR.get_*_embedding with an lsh plugin in one launch (bpr.py:48-125).  Inference only."""
    ids, table, feat, planes, buckets = (_ids(ids), _f32(table, "table"), _f32(feat, "feat"),
                                         _f32(planes, "planes"), _f32(buckets, "buckets"))
    B, (N, F), H, D = ids.numel(), feat.shape, planes.shape[0], buckets.shape[1]
    out = torch.empty((B, D), dtype=torch.float32, device=ids.device)
    with C.on_device(ids):
        rc = C.lib().mi_oov_lsh_lookup(C.ptr(ids), B, C.ptr(table), table.shape[0], C.ptr(feat), N, F, C.ptr(planes),
                                       H, C.ptr(buckets), D, C.ptr(out), C.stream_of(ids))
    C.check(rc, "mi_oov_lsh_lookup")
    return out


def lsh_lookup_score(ids, table, feat, planes, buckets, other, want_emb=False):
    """BPR.predict for one side with an lsh plugin: lookup (in-vocab row or lsh row) fused with the
    row dot against `other` (bpr.py:94-125,145-149).  Inference only."""
    ids, table, feat, planes, buckets, other = (_ids(ids), _f32(table, "table"), _f32(feat, "feat"),
                                                _f32(planes, "planes"), _f32(buckets, "buckets"),
                                                _f32(other, "other"))
    B, (N, F), H, D = ids.numel(), feat.shape, planes.shape[0], buckets.shape[1]
    if other.shape != (B, D):
        raise ValueError(f"other must be [{B},{D}], got {tuple(other.shape)}")
    if D > 256:  # as lsh_embed_score: rows by windows of columns, then the row dot
        out = lsh_lookup(ids, table, feat, planes, buckets)
        score = _rowdot_forward(other, out)
        return (score, out) if want_emb else score
    score = torch.empty((B,), dtype=torch.float32, device=ids.device)
    out = torch.empty((B, D), dtype=torch.float32, device=ids.device) if want_emb else None
    with C.on_device(ids):
        rc = C.lib().mi_oov_lsh_lookup_score(C.ptr(ids), B, C.ptr(table), table.shape[0], C.ptr(feat), N, F,
                                             C.ptr(planes), H, C.ptr(buckets), D, C.ptr(other), C.ptr(score),
                                             C.ptr(out), C.stream_of(ids))
    C.check(rc, "mi_oov_lsh_lookup_score")
    return (score, out) if want_emb else score


def _slsh_forward(ids, feat, planes, buckets, n_buckets, want_out=True):
    ids, feat, planes = _ids(ids), _f32(feat, "feat"), _f32(planes, "planes")
    B, (N, F), H = ids.numel(), feat.shape, planes.shape[0]
    idx = torch.empty((B,), dtype=torch.int64, device=ids.device)
    out = None
    D = 0
    if want_out:
        buckets = _f32(buckets, "buckets")
        n_buckets, D = buckets.shape
        out = torch.empty((B, D), dtype=torch.float32, device=ids.device)
    with C.on_device(ids):
        rc = C.lib().mi_oov_slsh_embed(C.ptr(ids), B, C.ptr(feat), N, F, C.ptr(planes), H,
                                       C.ptr(buckets) if want_out else None, n_buckets, D, C.ptr(out), C.ptr(idx),
                                       C.stream_of(ids))
    C.check(rc, "mi_oov_slsh_embed")
    return out, idx


def slsh_index(ids, feat, planes, n_buckets):
    """(2 ** bits).sum(1).long() % n_buckets (R/inductive/single_lsh_embedder.py:82-87)."""
    return _slsh_forward(ids, feat, planes, None, n_buckets, want_out=False)[1]


class _SlshEmbed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, feat, planes, buckets):
        out, idx = _slsh_forward(ids, feat, planes, buckets, buckets.shape[0])
        ctx.save_for_backward(idx)
        ctx.shape = buckets.shape
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return None, None, None, slsh_embed_backward(idx, g, ctx.shape[0])


def slsh_embed_backward(idx, grad_out, n_buckets):
    """grad of buckets[idx] w.r.t. buckets: deterministic for n_buckets <= 64 -- one launch for every bucket
    (mi_oov_slsh_embed_backward_fused; the two-launch form walks the batch once per eight buckets) --, float atomics above."""
    idx, g = _ids(idx, "idx"), _f32(grad_out, "grad_out")
    B, D = g.shape
    lib = C.lib()
    if D > 256 and n_buckets <= 64:  # the deterministic reduction holds 256 columns: one window of columns at a time
        return torch.cat([slsh_embed_backward(idx, g[:, d0:d0 + 256].contiguous(), n_buckets) for d0 in range(0, D, 256)], dim=1)
    out = torch.empty((n_buckets, D), dtype=torch.float32, device=g.device)
    if _bwd_mode():
        ws = torch.empty((max(int(lib.mi_oov_lsh_backward_fused_workspace(B, min(n_buckets, 64), D)), 1),), dtype=torch.float32, device=g.device)
        cnt = _bwd_counters(g) if _bwd_mode() == 1 else None
        with C.on_device(g):
            rc = lib.mi_oov_slsh_embed_backward_fused(C.ptr(idx), C.ptr(g), B, n_buckets, D, C.ptr(out), C.ptr(ws), C.ptr(cnt), C.stream_of(g))
        C.check(rc, "mi_oov_slsh_embed_backward_fused")
        return out
    ws = torch.empty((max(int(lib.mi_oov_lsh_backward_workspace(B, min(n_buckets, 64), D)), 1),), dtype=torch.float32,
                     device=g.device)
    with C.on_device(g):
        rc = lib.mi_oov_slsh_embed_backward(C.ptr(idx), C.ptr(g), B, n_buckets, D, C.ptr(out), C.ptr(ws), C.stream_of(g))
    C.check(rc, "mi_oov_slsh_embed_backward")
    return out


def scatter_add_rows(idx, g, n_rows, out=None):
    """out[idx[m]] += g[m] (float atomics; entries of idx outside [0, n_rows) are skipped) -- the backward of the
    row gathers.  Returns `out` (zeros[n_rows, D] when not given)."""
    idx, g = _ids(idx, "idx"), _f32(g, "g")
    M, D = g.shape
    if idx.numel() != M:
        raise ValueError(f"idx has {idx.numel()} entries, g {M} rows")
    if out is None:
        out = torch.zeros((n_rows, D), dtype=torch.float32, device=g.device)
    with C.on_device(g):
        rc = C.lib().mi_oov_scatter_add_rows(C.ptr(idx), M, C.ptr(g), n_rows, D, C.ptr(out), C.stream_of(g))
    C.check(rc, "mi_oov_scatter_add_rows")
    return out


def slsh_embed(ids, feat, planes, buckets):
    if torch.is_grad_enabled() and buckets.requires_grad:
        return _SlshEmbed.apply(ids, feat, planes, buckets)
    return _slsh_forward(ids, feat, planes, buckets, buckets.shape[0])[0]


def siphash24_mod(ids, keys, mod=16777216, out=None):
    """f32[B,K] of SipHash-2-4(key_j, LE64(id)) % mod (R/inductive/dh_embedder.py:140-170).  out: a contiguous f32[B, ld]
    buffer with ld >= K whose first K columns receive the hashes (the others are left alone); returned as is."""
    ids = _ids(ids)
    keys = C.dev_tensor(keys, torch.uint8, "keys")
    if keys.dim() != 2 or keys.shape[1] != 16:
        raise ValueError("keys must be u8[K,16]")
    B, K = ids.numel(), keys.shape[0]
    if out is None:
        out = torch.empty((B, K), dtype=torch.float32, device=ids.device)
    else:
        if not out.is_contiguous():  # (the validator below would hand the kernel a contiguous COPY and leave `out` unwritten)
            raise ValueError("out must be contiguous")
        out = _f32(out, "out")
        if out.dim() != 2 or out.shape[0] != B or out.shape[1] < K:
            raise ValueError(f"out must be f32[{B}, >= {K}], got {tuple(out.shape)}")
    with C.on_device(ids):
        rc = C.lib().mi_oov_siphash24_mod_ld(C.ptr(ids), B, C.ptr(keys), K, mod, C.ptr(out), out.shape[1], C.stream_of(ids))
    C.check(rc, "mi_oov_siphash24_mod")
    return out


def mapper_hash(ids, kind):
    ids = _ids(ids)
    out = torch.empty_like(ids)
    with C.on_device(ids):
        rc = C.lib().mi_oov_mapper_hash(C.ptr(ids), ids.numel(), HASH_KINDS[kind], C.ptr(out), C.stream_of(ids))
    C.check(rc, "mi_oov_mapper_hash")
    return out


def mapper_map(ids, kind, n_orig, n_buckets):
    """RandomOOVInductiveMapper.map_*_ids (R/inductive/random_mapper.py:116-130)."""
    if kind not in HASH_KINDS:
        raise ValueError(f"Unknown hash function {kind}")
    ids = _ids(ids)
    out = torch.empty_like(ids)
    with C.on_device(ids):
        rc = C.lib().mi_oov_mapper_map(C.ptr(ids), ids.numel(), HASH_KINDS[kind], int(n_orig), int(n_buckets),
                                       C.ptr(out), C.stream_of(ids))
    C.check(rc, "mi_oov_mapper_map")
    return out


def _gather_mean_forward(idx, W, g):
    idx, W = _ids(idx.reshape(-1), "idx"), _f32(W, "W")
    M, (N, D) = idx.numel(), W.shape
    out = torch.empty(((M + g - 1) // g, D), dtype=torch.float32, device=idx.device)
    with C.on_device(idx):
        rc = C.lib().mi_oov_gather_mean(C.ptr(idx), M, g, C.ptr(W), N, D, C.ptr(out), C.stream_of(idx))
    C.check(rc, "mi_oov_gather_mean")
    return out


class _GatherMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, idx, W, g):
        ctx.save_for_backward(idx.reshape(-1))
        ctx.g, ctx.shape = g, W.shape
        return _gather_mean_forward(idx, W, g)

    @staticmethod
    def backward(ctx, grad):
        (idx,) = ctx.saved_tensors
        M, g = idx.numel(), ctx.g
        grp = torch.arange(M, device=idx.device) // g
        cnt = torch.bincount(grp, minlength=grad.shape[0]).to(grad.dtype)
        return None, scatter_add_rows(idx, (grad / cnt[:, None])[grp], ctx.shape[0]), None


def gather_mean(idx, W, g=2):
    """vstack(chunk.mean(0) for chunk in W[idx.ravel()].split(g)) (knn_embedder.py:125-126)."""
    if torch.is_grad_enabled() and W.requires_grad:
        return _GatherMean.apply(idx, W, g)
    return _gather_mean_forward(idx, W, g)


def _gather_rows_forward(ids, W):
    ids, W = _ids(ids), _f32(W, "W")
    out = torch.empty((ids.numel(), W.shape[1]), dtype=torch.float32, device=ids.device)
    with C.on_device(ids):
        rc = C.lib().mi_oov_gather_rows(C.ptr(ids), ids.numel(), C.ptr(W), W.shape[0], W.shape[1], C.ptr(out),
                                        C.stream_of(ids))
    C.check(rc, "mi_oov_gather_rows")
    return out


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, W):
        ctx.save_for_backward(ids)
        ctx.shape = W.shape
        return _gather_rows_forward(ids, W)

    @staticmethod
    def backward(ctx, g):
        (ids,) = ctx.saved_tensors
        return None, scatter_add_rows(ids, g, ctx.shape[0])


def gather_rows(ids, W):
    """nn.Embedding forward: W[ids]."""
    if torch.is_grad_enabled() and W.requires_grad:
        return _GatherRows.apply(ids, W)
    return _gather_rows_forward(ids, W)


def _splice_forward(ids, rank, table, oov_rows):
    D = table.shape[1]
    out = torch.empty((ids.numel(), D), dtype=torch.float32, device=ids.device)
    with C.on_device(ids):
        rc = C.lib().mi_oov_splice_rows(C.ptr(ids), C.ptr(rank), ids.numel(), C.ptr(table), table.shape[0],
                                        C.ptr(oov_rows) if oov_rows.numel() else None, oov_rows.shape[0], D,
                                        C.ptr(out), C.stream_of(ids))
    C.check(rc, "mi_oov_splice_rows")
    return out


class _Splice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, rank, table, oov_rows):
        ctx.save_for_backward(ids, rank)
        ctx.tshape, ctx.oshape = table.shape, oov_rows.shape
        return _splice_forward(ids, rank, _f32(table, "table"), _f32(oov_rows, "oov_rows"))

    @staticmethod
    def backward(ctx, g):
        ids, rank = ctx.saved_tensors
        gt = scatter_add_rows(ids, g, ctx.tshape[0])  # ids >= n_vocab are skipped by the kernel
        go = scatter_add_rows(torch.where(ids < ctx.tshape[0], -1, rank), g, ctx.oshape[0])
        return None, None, gt, go


def splice_rows(ids, table, oov_rows):
    """zeros -> table rows where id < n_vocab -> oov_rows (in order of appearance) elsewhere
    (bpr.py:62-76,108-123)."""
    ids = _ids(ids)
    oov = ids >= table.shape[0]
    rank = torch.cumsum(oov, 0) - oov.to(torch.int64)
    if torch.is_grad_enabled() and (table.requires_grad or oov_rows.requires_grad):
        return _Splice.apply(ids, rank, table, oov_rows)
    return _splice_forward(ids, rank, _f32(table, "table"), _f32(oov_rows, "oov_rows"))


def col_mean(W):
    """torch.mean(W, dim=0) with a fixed summation order (mean_embedder.py:54-56,76-78)."""
    W = _f32(W, "W")
    N, D = W.shape
    mean = torch.empty((D,), dtype=torch.float32, device=W.device)
    ws = torch.empty((max(1, C.lib().mi_oov_col_mean_workspace(N, D)),), dtype=torch.float32, device=W.device)
    with C.on_device(W):
        rc = C.lib().mi_oov_col_mean(C.ptr(W), N, D, C.ptr(mean), C.ptr(ws), C.stream_of(W))
    C.check(rc, "mi_oov_col_mean")
    return mean


def broadcast_rows(vec, B, D=None, device=None):
    """vec.repeat(B, 1); vec=None gives zeros(D).repeat(B, 1) (zero_embedder.py:36-60)."""
    if vec is not None:
        vec = _f32(vec, "vec")
        D, device = vec.numel(), vec.device
    out = torch.empty((B, D), dtype=torch.float32, device=device)
    if out.device.type != "cuda":
        raise RuntimeError("mi_oov kernels run on an MI355X (ROCm) device only; there is no CPU fallback")
    with C.on_device(out):
        rc = C.lib().mi_oov_broadcast_rows(C.ptr(vec), B, D, C.ptr(out), C.stream_of(out))
    C.check(rc, "mi_oov_broadcast_rows")
    return out


def _rowdot_forward(U, E):
    U, E = _f32(U, "U"), _f32(E, "E")
    if U.shape != E.shape:
        raise ValueError(f"shape mismatch {tuple(U.shape)} vs {tuple(E.shape)}")
    s = torch.empty((U.shape[0],), dtype=torch.float32, device=U.device)
    with C.on_device(U):
        rc = C.lib().mi_oov_rowdot(C.ptr(U), C.ptr(E), U.shape[0], U.shape[1], C.ptr(s), C.stream_of(U))
    C.check(rc, "mi_oov_rowdot")
    return s


class _RowDot(torch.autograd.Function):
    @staticmethod
    def forward(ctx, U, E):
        ctx.save_for_backward(U, E)
        return _rowdot_forward(U, E)

    @staticmethod
    def backward(ctx, g):
        U, E = ctx.saved_tensors
        return g[:, None] * E, g[:, None] * U


def rowdot(U, E):
    """torch.mul(U, E).sum(dim=1) (bpr.py:145-149)."""
    if torch.is_grad_enabled() and (U.requires_grad or E.requires_grad):
        return _RowDot.apply(U, E)
    return _rowdot_forward(U, E)


def _full_sort_forward(U, E):
    U, E = _f32(U, "U"), _f32(E, "E")
    if U.shape[1] != E.shape[1]:
        raise ValueError("embedding widths differ")
    S = torch.empty((U.shape[0], E.shape[0]), dtype=torch.float32, device=U.device)
    with C.on_device(U):
        rc = C.lib().mi_oov_full_sort_scores(C.ptr(U), U.shape[0], C.ptr(E), E.shape[0], U.shape[1], C.ptr(S),
                                             C.stream_of(U))
    C.check(rc, "mi_oov_full_sort_scores")
    return S


class _FullSort(torch.autograd.Function):
    @staticmethod
    def forward(ctx, U, E):
        ctx.save_for_backward(U, E)
        return _full_sort_forward(U, E)

    @staticmethod
    def backward(ctx, g):  # dU = g E, dE = g^T U: the same GEMM kernel, contracted dimension put last by `transpose`
        U, E = ctx.saved_tensors
        g = g.contiguous()
        dU = _full_sort_forward(g, transpose(E)) if ctx.needs_input_grad[0] else None
        dE = _full_sort_forward(transpose(g), transpose(U)) if ctx.needs_input_grad[1] else None
        return dU, dE


def full_sort_scores(U, E):
    """torch.matmul(U, E.T) on the f32 matrix cores (bpr.py:151-163)."""
    if torch.is_grad_enabled() and (U.requires_grad or E.requires_grad):
        return _FullSort.apply(U, E)
    return _full_sort_forward(U, E)


ACTS = {None: 0, "none": 0, "gelu": 1, "sigmoid": 2}


def linear_act(X, W, bias, act=None):
    """act(X @ W.T + bias) on the f32 matrix cores (nn.Linear + nn.GELU()/nn.Sigmoid() of the hash nets,
    dh_embedder.py:70-89): the oracle's fmaf chain, bit for bit.  Forward only (`hash_net_train` supplies a backward)."""
    X, W, bias = _f32(X, "X"), _f32(W, "W"), _f32(bias, "bias")
    if X.shape[1] != W.shape[1] or bias.numel() != W.shape[0]:
        raise ValueError(f"shape mismatch: X {tuple(X.shape)}, W {tuple(W.shape)}, bias {tuple(bias.shape)}")
    Y = torch.empty((X.shape[0], W.shape[0]), dtype=torch.float32, device=X.device)
    with C.on_device(X):
        rc = C.lib().mi_oov_linear_act(C.ptr(X), X.shape[0], X.shape[1], C.ptr(W), C.ptr(bias), W.shape[0], ACTS[act],
                                       C.ptr(Y), C.stream_of(X))
    C.check(rc, "mi_oov_linear_act")
    return Y


class LinearX3Weights:
    """A Linear layer's weight split into its three bf16 planes for `linear_act_x3` (mi_oov_linear_x3_prepare): made
    once and re-made when the weight tensor was written to since or points at other memory (torch's version counter and
    the data pointer, as `LshTable`; after an in-place write through `.data` call `invalidate()`)."""

    __slots__ = ("_weight", "split", "version", "addr", "transposed")

    def __init__(self, weight, transposed=False):
        """transposed: `weight` is the TRANSPOSE of the layer's weight, [K, N_out] (mi_oov_linear_x3_prepare_t: training's
        backward products take their operands as they lie)."""
        self._weight = weakref.ref(weight)  # (weak: hash_net_forward keys its cache of these by the weight tensor)
        self.split, self.version, self.addr, self.transposed = None, None, None, bool(transposed)

    def invalidate(self):
        self.version = None

    def get(self):
        w = self._weight()
        if w is None:
            raise RuntimeError("the weight tensor of this LinearX3Weights is gone")
        if self.split is None or self.version != w._version or self.addr != w.data_ptr() or self.split.device != w.device:
            src = _f32(w, "W")
            n, k = (src.shape[1], src.shape[0]) if self.transposed else src.shape
            nbytes = int(C.lib().mi_oov_linear_x3_weights_bytes(n, k))
            if nbytes <= 0:
                raise ValueError(f"unsupported weight shape {tuple(src.shape)}")
            if self.split is None or self.split.numel() != nbytes or self.split.device != w.device:
                self.split = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
            prepare = C.lib().mi_oov_linear_x3_prepare_t if self.transposed else C.lib().mi_oov_linear_x3_prepare
            with C.on_device(src):
                rc = prepare(C.ptr(src), n, k, C.ptr(self.split), C.stream_of(src))
            C.check(rc, "mi_oov_linear_x3_prepare")
            self.version, self.addr = w._version, w.data_ptr()
        return self.split


def _x3_ksplit(rows, n_out, k):
    """Shares of K for a training product: enough 128 x 128 tiles x shares to give every CU two workgroups, no share
    below 8 stages of 16 k."""
    tiles = -(-rows // 128) * -(-n_out // 128)
    return max(1, min(512 // tiles, -(-k // 16) // 8, 64))


def linear_act_x3(X, W, bias, act=None, weights=None, ksplit=1):
    """act(X @ W.T + bias) on the bf16 matrix cores at f32 accuracy (mi_oov_linear_x3: every operand as three bf16
    planes, six products accumulated in f32; csrc/linear3.hip).  The inference form of the hash nets' layers
    (dh_embedder.py:70-89): within an f32 accumulation's error of `linear_act`, not bit-identical to it.  `weights`: a
    `LinearX3Weights` of W to reuse its split."""
    X, W, bias = _f32(X, "X"), _f32(W, "W"), _f32(bias, "bias")
    if weights is not None and weights.transposed:  # W is the transposed weight [K, N_out]; only its split is used
        W = W.t()
    K = W.shape[1]
    # X may carry the columns up to the next multiple of 16 (finite values: they meet the zero weights the split pads
    # with) -- rows of a multiple of 16 floats take the pipelined kernel, see `hash_net_forward`
    if (X.shape[1] != K and X.shape[1] != -(-K // 16) * 16) or bias.numel() != W.shape[0]:
        raise ValueError(f"shape mismatch: X {tuple(X.shape)}, W {tuple(W.shape)}, bias {tuple(bias.shape)}")
    split = (weights if weights is not None else LinearX3Weights(W)).get()
    if split.numel() != int(C.lib().mi_oov_linear_x3_weights_bytes(W.shape[0], K)) or split.device != X.device:
        raise ValueError("`weights` is not the split of a weight of this shape on this device")
    Y = torch.empty((X.shape[0], W.shape[0]), dtype=torch.float32, device=X.device)
    if ksplit > 1:  # (training shapes: `_gemm_nt`; the rounding depends on ksplit, so inference never takes this)
        ws = torch.empty((int(C.lib().mi_oov_linear_x3_splitk_workspace(X.shape[0], W.shape[0], ksplit)),), dtype=torch.uint8, device=X.device)
        with C.on_device(X):
            rc = C.lib().mi_oov_linear_x3_splitk(C.ptr(X), X.shape[0], X.shape[1], C.ptr(split), C.ptr(bias), W.shape[0], ACTS[act],
                                                 C.ptr(Y), ksplit, C.ptr(ws), C.stream_of(X))
        C.check(rc, "mi_oov_linear_x3_splitk")
        return Y
    with C.on_device(X):
        rc = C.lib().mi_oov_linear_x3(C.ptr(X), X.shape[0], X.shape[1], C.ptr(split), C.ptr(bias), W.shape[0], ACTS[act],
                                      C.ptr(Y), C.stream_of(X))
    C.check(rc, "mi_oov_linear_x3")
    return Y


def _hash_net_layers(net):
    """[(Linear, act name or None), ...] of an nn.Sequential of Linear / GELU (erf form) / Sigmoid."""
    mods, layers, i = list(net), [], 0
    while i < len(mods):
        lin = mods[i]
        if not isinstance(lin, torch.nn.Linear) or lin.bias is None:
            raise TypeError("a hash net is Linear(+bias) layers, each optionally followed by GELU / Sigmoid")
        act = None
        if i + 1 < len(mods) and isinstance(mods[i + 1], torch.nn.GELU) and mods[i + 1].approximate == "none":
            act, i = "gelu", i + 1
        elif i + 1 < len(mods) and isinstance(mods[i + 1], torch.nn.Sigmoid):
            act, i = "sigmoid", i + 1
        layers.append((lin, act))
        i += 1
    return layers


# `hash_net_forward` runs every batch on the split-bf16 layers: they are quicker than the f32 matrix instruction at every
# size (65536 x 1024 -> 512: 324 vs 587 us, 1024 rows: 56 vs 91, 64 rows: 65 vs 86) and every tile form does the same
# arithmetic in the same order, so a row's result does not depend on the batch it sits in.  Training (`hash_net_train`)
# runs on the split layers too -- forward on `mi_oov_linear_x3`, the three backward products on `mi_oov_linear_x3_splitk`,
# whose rounding depends on the number of K shares and so on the batch shape (`_x3_ksplit`).  MI_OOV_LINEAR_X3=0 puts both
# on the f32 kernel (`linear_act`, bit-identical to the oracle's chain).
_x3_weights = {}  # id(Linear.weight) -> LinearX3Weights (which holds the tensor weakly; dropped when the tensor dies)


def _x3_weights_of(weight):
    w = _x3_weights.get(id(weight))
    if w is None or w._weight() is not weight:
        w = _x3_weights[id(weight)] = LinearX3Weights(weight)
        weakref.finalize(weight, _x3_weights.pop, id(weight), None)
    return w


def _x3_wanted():
    return os.environ.get("MI_OOV_LINEAR_X3", "") != "0"


def hash_net_forward(net, x):
    """Run an nn.Sequential of Linear / GELU / Sigmoid (the reference's *_hash_net, dh_embedder.py:70-89) with each
    activation fused into the producing layer's epilogue.  Inference form: the layers run on the bf16 matrix cores at
    f32 accuracy (`linear_act_x3`; the weights' three-plane split is kept per weight tensor and re-made when the tensor's
    version counter moves; an input whose width is not a multiple of 16 is padded with zero columns once).  Within an
    f32 accumulation's error of the f32 product, not the oracle's summation order; MI_OOV_LINEAR_X3=0 selects the
    bit-exact f32 kernel instead.  Under autograd `hash_net_train` keeps the pre-activations and supplies the backward,
    on the same split-bf16 GEMM (`_gemm_nt`: mi_oov_linear_x3_splitk; gradients then depend on the batch shape through the
    K shares) unless MI_OOV_LINEAR_X3=0."""
    x3 = _x3_wanted()
    for lin, act in _hash_net_layers(net):
        if x3:
            if x.shape[1] % 16:  # (fdhe: K hashes + F feature columns) zero columns up to a multiple of 16: the pipelined kernel
                x = torch.nn.functional.pad(x, (0, -x.shape[1] % 16))
            x = linear_act_x3(x, lin.weight, lin.bias, act, _x3_weights_of(lin.weight))
        else:
            x = linear_act(x, lin.weight, lin.bias, act)
    return x


def _elementwise(name, act, *tensors):
    out = torch.empty_like(tensors[-1])
    n = out.numel()
    with C.on_device(out):
        rc = getattr(C.lib(), name)(*[C.ptr(t) for t in tensors], n, ACTS[act], C.ptr(out), C.stream_of(out))
    C.check(rc, name)
    return out


def act_forward(Z, act):
    """act(Z) elementwise (mi_oov_act_forward): the expressions of mi_oov_linear_act's fused epilogue."""
    return _elementwise("mi_oov_act_forward", act, _f32(Z, "Z"))


def act_backward(dY, Z, act):
    """dY * act'(Z) (mi_oov_act_backward): GELU (erf form) Phi(z) + z phi(z); Sigmoid y (1 - y)."""
    return _elementwise("mi_oov_act_backward", act, _f32(dY, "dY"), _f32(Z, "Z"))


def transpose(A):
    """A.T as a dense row-major matrix (mi_oov_transpose): the [rows, k] layout the GEMM kernel contracts over."""
    A = _f32(A, "A")
    R, Cc = A.shape
    At = torch.empty((Cc, R), dtype=torch.float32, device=A.device)
    with C.on_device(A):
        rc = C.lib().mi_oov_transpose(C.ptr(A), R, Cc, C.ptr(At), C.stream_of(A))
    C.check(rc, "mi_oov_transpose")
    return At


_X3_TRAIN_MAX_SPLIT_BYTES = 256 << 20  # an operand whose three-plane split would be larger goes through the f32 kernel


def _gemm_nt(A, Bt):
    """A [M,K] x Bt [K,N] -> [M,N], the product every step of the hash nets' training is made of.  On the split-bf16
    kernel (`linear_act_x3` with a zero bias; Bt is split as it lies, `mi_oov_linear_x3_prepare_t`, K cut into shares for
    the shapes with few output tiles) unless MI_OOV_LINEAR_X3=0 or Bt's split would not fit
    `_X3_TRAIN_MAX_SPLIT_BYTES`: then the f32 kernel on a transposed copy."""
    if _x3_wanted() and Bt.numel() * 6 <= _X3_TRAIN_MAX_SPLIT_BYTES:
        zero = torch.zeros((Bt.shape[1],), dtype=torch.float32, device=A.device)
        return linear_act_x3(A, Bt, zero, None, LinearX3Weights(Bt, transposed=True), _x3_ksplit(A.shape[0], Bt.shape[1], A.shape[1]))
    return _full_sort_forward(A, transpose(Bt))


class _HashNet(torch.autograd.Function):
    """The reference's *_hash_net under autograd (dh_embedder.py:70-89,191-217; dnn_embedder.py:65-109) on this
    library's kernels only.  forward keeps every layer's input and pre-activation; backward per layer:
        dZ = dY * act'(Z)     dW = dZ^T X     db = 1^T dZ     dX = dZ W
    -- three products after `transpose` has put the contracted dimension last, each on the split-bf16 GEMM
    (`_gemm_nt`: f32 accuracy on the bf16 matrix cores, 1.6 x the f32 matrix instruction at a 2048-row step; round 3)
    or, under MI_OOV_LINEAR_X3=0, on the f32-MFMA GEMM (one fmaf chain per element over increasing k)."""

    @staticmethod
    def forward(ctx, x, acts, *params):
        h, saved = _f32(x, "x"), []
        x3 = _x3_wanted()
        for li, act in enumerate(acts):
            W, b = params[2 * li], params[2 * li + 1]
            if x3:
                hp = torch.nn.functional.pad(h, (0, -h.shape[1] % 16)) if h.shape[1] % 16 else h
                z = linear_act_x3(hp, W, b, None, _x3_weights_of(W))
            else:
                z = linear_act(h, W, b, None)
            saved += [h, z]
            h = act_forward(z, act) if act else z
        ctx.save_for_backward(*saved, *[p for i, p in enumerate(params) if i % 2 == 0])
        ctx.acts = acts
        ctx.x_needs = x.requires_grad
        return h

    @staticmethod
    def backward(ctx, g):
        n = len(ctx.acts)
        saved, Ws = ctx.saved_tensors[:2 * n], ctx.saved_tensors[2 * n:]
        g = g.contiguous()
        grads = [None] * (2 * n)
        ones = None
        for li in range(n - 1, -1, -1):
            h, z = saved[2 * li], saved[2 * li + 1]
            dz = act_backward(g, z, ctx.acts[li]) if ctx.acts[li] else g
            grads[2 * li] = _gemm_nt(transpose(dz), h)  # dZ^T [out, B] x X [B, in] -> [out, in]
            if ones is None:
                ones = torch.ones((1, dz.shape[0]), dtype=torch.float32, device=dz.device)
            grads[2 * li + 1] = _gemm_nt(ones, dz).view(-1)  # 1^T [1, B] x dZ [B, out] -> [out]
            if li > 0 or ctx.x_needs:
                g = _gemm_nt(dz, Ws[li])  # dZ [B, out] x W [out, in] -> [B, in]
        return (g if ctx.x_needs else None, None, *grads)


def hash_net_train(net, x):
    """`net(x)` with gradients to the Linear weights / biases (and to x when it requires one), every forward and
    backward operation a kernel of this library."""
    layers = _hash_net_layers(net)
    params = []
    for lin, _ in layers:
        params += [lin.weight, lin.bias]
    return _HashNet.apply(x, tuple(a for _, a in layers), *params)


_TOPK_WORKSPACE_MAX_BYTES = 2 << 30


class TopkCatalogue:
    """A catalogue E prepared once for many `score_topk` / `score_topk_excl` calls (`mi_oov_topk_catalogue_prepare`): the
    counterpart of the reference building its ScaNN searcher at construction (knn_embedder.py:84-93).  Holds a reference
    to E and is valid while E is not written to (`fresh()` checks torch's version counter); `TopkCatalogue.of(E)` returns
    None for shapes the fused bf16 path does not take (callers then pass the plain tensor)."""

    def __init__(self, E):
        E = _f32(E, "E")
        nbytes = int(C.lib().mi_oov_topk_catalogue_bytes(E.shape[0], E.shape[1]))
        if nbytes <= 0 or (E.shape[1] in (64, 128) and E.data_ptr() % 16):
            raise ValueError("TopkCatalogue needs a float32 table of at most 128 columns (16-byte aligned when it has 64 or 128)")
        self.E = E
        self.version = E._version
        self.buf = torch.empty((nbytes,), dtype=torch.uint8, device=E.device)
        with C.on_device(E):
            rc = C.lib().mi_oov_topk_catalogue_prepare(C.ptr(E), E.shape[0], E.shape[1], C.ptr(self.buf), C.stream_of(E))
        C.check(rc, "mi_oov_topk_catalogue_prepare")

    @staticmethod
    def of(E):
        ok = torch.is_tensor(E) and E.is_cuda and E.dtype == torch.float32 and E.dim() == 2 and E.is_contiguous() \
            and 0 < E.shape[1] <= 128 and E.shape[0] > 0 and (E.shape[1] not in (64, 128) or E.data_ptr() % 16 == 0)
        return TopkCatalogue(E) if ok else None

    def fresh(self, E=None):
        same = E is None or (E.data_ptr() == self.E.data_ptr() and E.shape == self.E.shape)
        return same and self.E._version == self.version


def _prepared_call(U, cat, k, n_skip_low, excl_ptr, excl_cols, vals, idx):
    """One mi_oov_score_topk_prepared launch; returns False when the shape is not one the fused bf16 path takes."""
    lib = C.lib()
    B, N, D = U.shape[0], cat.E.shape[0], U.shape[1]
    need = int(lib.mi_oov_score_topk_prepared_workspace(B, N, D, k, 1 if excl_ptr is not None else 0))
    if need <= 0 or (D in (64, 128) and U.data_ptr() % 16):
        return False
    ws = torch.empty((need,), dtype=torch.uint8, device=U.device)
    with C.on_device(U):
        rc = lib.mi_oov_score_topk_prepared(C.ptr(U), B, C.ptr(cat.E), N, D, k, int(n_skip_low), C.ptr(excl_ptr), C.ptr(excl_cols),
                                            C.ptr(cat.buf), C.ptr(vals), C.ptr(idx), C.ptr(ws), C.stream_of(U))
    C.check(rc, "mi_oov_score_topk_prepared")
    return True


def score_topk(U, E, k, n_skip_low=0):
    """Per-row top-k of U @ E.T without returning the [B,N] matrix.  Returns (vals, idx).  E: the table, or a
    `TopkCatalogue` of it."""
    cat = E if isinstance(E, TopkCatalogue) else None
    if cat is not None:
        if not cat.fresh():
            raise ValueError("the catalogue's table was modified after TopkCatalogue was built")
        E = cat.E
    U, E = _f32(U, "U"), _f32(E, "E")
    B, N, D = U.shape[0], E.shape[0], U.shape[1]
    vals = torch.empty((B, k), dtype=torch.float32, device=U.device)
    idx = torch.empty((B, k), dtype=torch.int64, device=U.device)
    lib = C.lib()
    def ws_bytes(rows):  # a prepared catalogue holds the bf16 copy of E: its calls need the lists only
        need = int(lib.mi_oov_score_topk_prepared_workspace(rows, N, D, k, 0)) if cat is not None else 0
        return need if need > 0 else int(lib.mi_oov_score_topk_workspace_d(rows, N, D, k))

    if cat is not None and B > 0 and ws_bytes(B) <= _TOPK_WORKSPACE_MAX_BYTES \
            and _prepared_call(U, cat, k, n_skip_low, None, None, vals, idx):
        return vals, idx
    # the fused path's workspace grows with B x N (tile maxima, candidate lists): a 10 M-row catalogue is ~165 KB per
    # user, so big batches go through in chunks of users (multiples of the 128-row tile)
    rows = B
    while rows > 128 and ws_bytes(rows) > _TOPK_WORKSPACE_MAX_BYTES:
        rows = max(128, (rows // 2 + 127) // 128 * 128)
    ws = None  # (a prepared catalogue's chunks bring their own, smaller workspace: the general one -- with room for a bf16
    # copy of E, gigabytes at 10 M rows -- is allocated only if a chunk falls through to mi_oov_score_topk)
    with C.on_device(U):
        for b0 in range(0, max(B, 1), max(rows, 1)):
            nb = min(rows, B - b0)
            if cat is not None and nb > 0 and _prepared_call(U[b0:b0 + nb], cat, k, n_skip_low, None, None, vals[b0:b0 + nb], idx[b0:b0 + nb]):
                continue
            if ws is None:
                ws = torch.empty((max(16, int(lib.mi_oov_score_topk_workspace_d(min(rows, B), N, D, k))),), dtype=torch.uint8, device=U.device)
            rc = lib.mi_oov_score_topk(C.ptr(U[b0:b0 + nb]), nb, C.ptr(E), N, D, k, n_skip_low, C.ptr(vals[b0:b0 + nb]),
                                       C.ptr(idx[b0:b0 + nb]), C.ptr(ws), C.stream_of(U))
            C.check(rc, "mi_oov_score_topk")
    return vals, idx


def segment_topk(scores, cols, seg_ptr, k, col_lo=0, col_hi=None):
    """Top-k of sparse candidate lists: what torch.topk returns on the dense -inf score matrix of
    InductiveEvaluator.neg_sample_batch_eval (R/inductive/evaluator.py:118-134), without building it.
    scores f32[M], cols i64[M], seg_ptr i64[S+1] -> (vals f32[S,k], idx i64[S,k]); (-inf, -1) where a segment
    has fewer than k candidates with col_lo <= column < col_hi."""
    scores, cols, seg_ptr = _f32(scores, "scores"), _ids(cols, "cols"), _ids(seg_ptr, "seg_ptr")
    S = seg_ptr.numel() - 1
    if scores.numel() != cols.numel():
        raise ValueError(f"{scores.numel()} scores for {cols.numel()} columns")
    if not 0 < k <= 256:
        raise ValueError("segment_topk supports 1 <= k <= 256")
    vals = torch.empty((S, k), dtype=torch.float32, device=scores.device)
    idx = torch.empty((S, k), dtype=torch.int64, device=scores.device)
    with C.on_device(scores):
        rc = C.lib().mi_oov_segment_topk(C.ptr(scores), C.ptr(cols), C.ptr(seg_ptr), S, k, int(col_lo),
                                         int(col_hi) if col_hi is not None else (1 << 62), C.ptr(vals), C.ptr(idx),
                                         C.stream_of(scores))
    C.check(rc, "mi_oov_segment_topk")
    return vals, idx


def topk_hits(idx, pos_ptr, pos_cols, col_lo=0, col_hi=None):
    """The collector's rec.topk block (collector.py:161-166): int32[S, k+1] = hit flags of idx[s,:] against the
    positives of segment s (CSR) followed by the positive count.  col_lo / col_hi: only positives with a column in
    [col_lo, col_hi) count (mi_oov_topk_hits_range: the old-item / new-item slices, no compaction of the lists)."""
    idx, pos_ptr, pos_cols = _ids(idx, "idx"), _ids(pos_ptr, "pos_ptr"), _ids(pos_cols, "pos_cols")
    S, k = idx.shape
    out = torch.empty((S, k + 1), dtype=torch.int32, device=idx.device)
    if pos_cols.numel() == 0:
        pos_cols = torch.zeros((1,), dtype=torch.int64, device=idx.device)
    with C.on_device(idx):
        rc = C.lib().mi_oov_topk_hits_range(C.ptr(idx), S, k, C.ptr(pos_ptr), C.ptr(pos_cols), int(col_lo),
                                            int(col_hi) if col_hi is not None else (1 << 62), C.ptr(out), C.stream_of(idx))
    C.check(rc, "mi_oov_topk_hits_range")
    return out


def eval_rows_build(pos_ptr, user_ids, pos_items, neg_items, n_neg, want_pos_user=False):
    """The rows of a group of NegSampleEvalDataLoader batches in one pass (mi_oov_eval_rows_build; general_dataloader.py:
    157-190): per user its positives, then n_neg sampled items per positive.  pos_ptr i64[U+1] (CSR of the positives per
    user, pos_ptr[0] = 0), user_ids i64[U], pos_items i64[P], neg_items i64[P * n_neg] in user order.
    -> (row_user i64[M], row_item i64[M], seg_ptr i64[U+1][, pos_user i64[P]]), M = P (1 + n_neg).  No host sync: P is
    taken from the shape of pos_items."""
    pos_ptr, user_ids, pos_items = _ids(pos_ptr, "pos_ptr"), _ids(user_ids, "user_ids"), _ids(pos_items, "pos_items")
    neg_items = _ids(neg_items, "neg_items")
    U, P, n_neg = user_ids.numel(), pos_items.numel(), int(n_neg)
    if pos_ptr.numel() != U + 1 or neg_items.numel() != P * n_neg:
        raise ValueError(f"need pos_ptr i64[{U + 1}] and neg_items i64[{P * n_neg}], got {pos_ptr.numel()} and {neg_items.numel()}")
    dev = pos_items.device
    M = P * (1 + n_neg)
    row_user = torch.empty((M,), dtype=torch.int64, device=dev)
    row_item = torch.empty((M,), dtype=torch.int64, device=dev)
    seg_ptr = torch.empty((U + 1,), dtype=torch.int64, device=dev)
    pos_user = torch.empty((P,), dtype=torch.int64, device=dev) if want_pos_user else None
    with C.on_device(pos_items):
        rc = C.lib().mi_oov_eval_rows_build(C.ptr(pos_ptr), U, C.ptr(user_ids), C.ptr(pos_items), C.ptr(neg_items), n_neg,
                                            C.ptr(row_user), C.ptr(row_item), C.ptr(seg_ptr), C.ptr(pos_user), C.stream_of(pos_items))
    C.check(rc, "mi_oov_eval_rows_build")
    return (row_user, row_item, seg_ptr, pos_user) if want_pos_user else (row_user, row_item, seg_ptr)


METRIC_IDS = {"recall": 0, "hit": 1, "precision": 2, "ndcg": 3, "mrr": 4, "map": 5}


def topk_metric_sums(rec, disc, idcg_base, uids=None, n_old_users=0):
    """Column sums of the TopkMetric curves of a rec.topk block over its users, in user order (mi_oov_topk_metric_sums;
    R/evaluator/metrics.py:36-235): rec i32[U, K+1] on the device, disc / idcg_base f64[K] device tensors made from the
    host's NumPy values -> (sums f64[n_sides, 6, K], counts i64[n_sides, 6]) on the device; n_sides = 3 (all users, users
    with uids < n_old_users, the others) when uids is given, else 1.  sums / counts are the means NumPy computes, bit for
    bit."""
    rec = C.dev_tensor(rec, torch.int32, "rec")
    disc, idcg_base = C.dev_tensor(disc, torch.float64, "disc"), C.dev_tensor(idcg_base, torch.float64, "idcg_base")
    U, K = rec.shape[0], rec.shape[1] - 1
    if disc.numel() != K or idcg_base.numel() != K:
        raise ValueError(f"disc and idcg_base must have {K} entries")
    n_sides = 1 if uids is None else 3
    if uids is not None:
        uids = _ids(uids, "uids")
        if uids.numel() != U:
            raise ValueError(f"uids has {uids.numel()} entries, rec {U} rows")
    sums = torch.empty((n_sides, 6, K), dtype=torch.float64, device=rec.device)
    counts = torch.empty((n_sides, 6), dtype=torch.int64, device=rec.device)
    ws = torch.empty((max(int(C.lib().mi_oov_topk_metric_sums_workspace(U, K)), 8),), dtype=torch.uint8, device=rec.device)
    with C.on_device(rec):
        rc = C.lib().mi_oov_topk_metric_sums(C.ptr(rec), U, K, C.ptr(disc), C.ptr(idcg_base), C.ptr(uids), int(n_old_users), n_sides,
                                             C.ptr(sums), C.ptr(counts), C.ptr(ws), C.stream_of(rec))
    C.check(rc, "mi_oov_topk_metric_sums")
    return sums, counts


def segment_dedup(cols, seg_ptr):
    """cols with every repeated column of a segment (after its first occurrence) replaced by -1 (mi_oov_segment_dedup):
    what the reference's dense scatter `scores[row_idx, col_idx] = origin_scores` does to a duplicated candidate
    (R/inductive/evaluator.py:118-134).  `segment_topk` skips the -1 entries when it is given a column range."""
    cols, seg_ptr = _ids(cols, "cols"), _ids(seg_ptr, "seg_ptr")
    out = torch.empty_like(cols)
    with C.on_device(cols):
        rc = C.lib().mi_oov_segment_dedup(C.ptr(cols), C.ptr(seg_ptr), seg_ptr.numel() - 1, C.ptr(out), C.stream_of(cols))
    C.check(rc, "mi_oov_segment_dedup")
    return out


_MASKED_TOPK_MAX_BYTES = 4 << 30  # workspace bound for the bitmap path (B x N / 8 bytes of mask + the fused lists)
_USE_MASKED_TOPK = True           # tests flip this to cover the k + h_max path on shapes the masked kernel takes


def score_topk_excl(U, E, k, excl_ptr, excl_cols, n_skip_low=0, h_max=None):
    """top-k of U @ E.T per user with scores[:, :n_skip_low] and the user's excluded columns (history) treated as
    -inf (InductiveEvaluator.eval_batch, R/inductive/evaluator.py:92-95), without returning the scores.
    excl_ptr i64[B+1], excl_cols i64[nnz].  Routes, all inside the library and none with a host synchronisation:
      * rows of up to 128 floats over a catalogue the fused path takes: `mi_oov_score_topk_masked` (exclusion bitmap inside
        the fused bf16 kernel; any history length); a user batch whose workspace would pass `_MASKED_TOPK_MAX_BYTES` goes
        through in chunks of users (the exclusion CSR is addressed through a shifted excl_ptr: nothing is copied);
      * everything else (D > 128, k > 256, a catalogue of fewer than 128 k rows): `mi_oov_score_topk_excl_dense` -- scores of
        <= 1 GiB of users materialised, bitmap, exact select.
    `h_max` (longest history, known to the caller) with `_USE_MASKED_TOPK = False` selects the older top-(k + h_max) route
    (`mi_oov_score_topk_excl`, k + h_max <= 256; tests and tools/tune.py compare it with the masked kernel)."""
    cat = E if isinstance(E, TopkCatalogue) else None
    if cat is not None:
        if not cat.fresh():
            raise ValueError("the catalogue's table was modified after TopkCatalogue was built")
        E = cat.E
    U, E = _f32(U, "U"), _f32(E, "E")
    excl_ptr, excl_cols = _ids(excl_ptr, "excl_ptr"), _ids(excl_cols, "excl_cols")
    B, N, D = U.shape[0], E.shape[0], U.shape[1]
    if excl_ptr.numel() != B + 1:
        raise ValueError(f"excl_ptr must have {B + 1} entries, got {excl_ptr.numel()}")
    vals = torch.empty((B, k), dtype=torch.float32, device=U.device)
    idx = torch.empty((B, k), dtype=torch.int64, device=U.device)
    if B == 0:
        return vals, idx
    if excl_cols.numel() == 0:
        excl_cols = torch.zeros((1,), dtype=torch.int64, device=U.device)
    lib = C.lib()

    def masked_bytes(rows):
        return int(lib.mi_oov_score_topk_prepared_workspace(rows, N, D, k, 1)) if cat is not None else \
            int(lib.mi_oov_score_topk_masked_workspace(rows, N, D, k))

    if _USE_MASKED_TOPK and masked_bytes(B) > 0 and U.data_ptr() % 16 == 0 and E.data_ptr() % 16 == 0:
        rows = B  # users per call: a multiple of the 128-row tile once the batch is cut (row stride D * 4: chunks stay aligned)
        while rows > 128 and masked_bytes(rows) > _MASKED_TOPK_MAX_BYTES:
            rows = max(128, (rows // 2 + 127) // 128 * 128)
        ws = None
        with C.on_device(U):
            for b0 in range(0, B, rows):
                nb = min(rows, B - b0)
                ptr_b = excl_ptr[b0:b0 + nb + 1]  # absolute offsets into excl_cols: a view is the chunk's CSR
                if cat is not None and _prepared_call(U[b0:b0 + nb], cat, k, n_skip_low, ptr_b, excl_cols, vals[b0:b0 + nb], idx[b0:b0 + nb]):
                    continue
                if ws is None:
                    ws = torch.empty((int(lib.mi_oov_score_topk_masked_workspace(min(rows, B), N, D, k)),), dtype=torch.uint8, device=U.device)
                rc = lib.mi_oov_score_topk_masked(C.ptr(U[b0:b0 + nb]), nb, C.ptr(E), N, D, k, int(n_skip_low), C.ptr(ptr_b),
                                                  C.ptr(excl_cols), C.ptr(vals[b0:b0 + nb]), C.ptr(idx[b0:b0 + nb]), C.ptr(ws), C.stream_of(U))
                C.check(rc, "mi_oov_score_topk_masked")
        return vals, idx
    if not _USE_MASKED_TOPK and h_max is not None and k + int(h_max) <= 256:
        longest = int(h_max)
        ws = torch.empty((max(int(lib.mi_oov_score_topk_excl_workspace(B, N, k, longest)), 16),), dtype=torch.uint8, device=U.device)
        with C.on_device(U):
            rc = lib.mi_oov_score_topk_excl(C.ptr(U), B, C.ptr(E), N, D, k, int(n_skip_low), C.ptr(excl_ptr),
                                            C.ptr(excl_cols), longest, C.ptr(vals), C.ptr(idx), C.ptr(ws), C.stream_of(U))
        C.check(rc, "mi_oov_score_topk_excl")
        return vals, idx
    ws = torch.empty((max(int(lib.mi_oov_score_topk_excl_dense_workspace(B, N)), 16),), dtype=torch.uint8, device=U.device)
    with C.on_device(U):
        rc = lib.mi_oov_score_topk_excl_dense(C.ptr(U), B, C.ptr(E), N, D, k, int(n_skip_low), C.ptr(excl_ptr), C.ptr(excl_cols),
                                              C.ptr(vals), C.ptr(idx), C.ptr(ws), C.stream_of(U))
    C.check(rc, "mi_oov_score_topk_excl_dense")
    return vals, idx
