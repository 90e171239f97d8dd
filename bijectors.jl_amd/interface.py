"""Host-side mirror of the reference's Bijector interface for the hot path.

Same names, argument meaning and error behaviour as Bijectors.jl v0.16.2
(src/interface.jl:133-281 and the per-bijector files under src/bijectors/), so the parity tests
read like the reference's own tests.  All array math happens in libbjx_hip.so through the C ABI
(include/bjx.h); PyTorch is used only for device memory and the HIP stream handle.

Array orientation is the reference's: a batch is a 2-D tensor of shape ``(dim, batch)`` stored
column-major (``x.stride() == (1, dim)``, i.e. ``torch.empty(batch, dim).T``), a single sample is
a 1-D tensor.  `colmajor(t)` converts.  Outputs use the same convention.

Return shapes follow the reference bijector by bijector (SURVEY.md §8a'): elementwise bijectors,
Simplex and chains return ONE scalar log-det for a matrix input; Ordered / Planar / Radial /
InvertibleBatchNorm return a per-column vector; Planar returns a namedtuple
``(result, logabsdetjac)``.  Every bijector additionally accepts ``per_sample=True`` in
`with_logabsdet_jacobian` to get the per-column vector (what a density evaluation needs,
SURVEY.md §8a'' row 2).
"""
from __future__ import annotations

import ctypes as C
import math
from collections import namedtuple
from typing import Optional, Sequence

import torch

from . import _lib as L

__all__ = [
    "Transform", "Bijector", "Inverse", "ComposedFunction", "Elementwise", "elementwise", "exp", "log", "identity",
    "Shift", "Scale", "Logit", "LeakyReLU", "TruncatedBijector", "SignFlip", "OrderedBijector", "SimplexBijector",
    "VecCholeskyBijector", "VecCorrBijector", "CorrBijector", "PDBijector", "PDVecBijector", "Permute", "PlanarLayer", "RadialLayer", "InvertibleBatchNorm", "RationalQuadraticSpline",
    "PartitionMask", "Coupling", "Stacked", "NamedStacked", "Columnwise", "columnwise", "vjp", "istraining", "training", "transform", "inverse", "logabsdetjac", "with_logabsdet_jacobian",
    "with_logabsdet_jacobian_", "transform_", "output_size", "isinvertible", "isclosedform", "colmajor", "context", "_fast_plans",
    "PlanarResult", "vjp_params", "row_moments", "MvNormal", "TorchBase", "TransformedDistribution", "transformed", "logpdf", "rand",
    "CapturedStep", "kernel_timed", "cache_params", "invalidate_params",
]

PlanarResult = namedtuple("PlanarResult", ["result", "logabsdetjac"])  # planar_layer.jl:109

# ------------------------------------------------------------------ device plumbing
_ctx_cache: dict = {}


class _Ctx:
    """One bjx_ctx per (device, stream)."""

    def __init__(self, device: int, stream_ptr: int):
        lib = L.load()
        h = C.c_void_p()
        rc = lib.bjx_create(device, C.c_void_p(stream_ptr), C.byref(h))
        if rc != 0:
            raise L.BjxError(f"bjx_create(device={device}) failed with status {rc}")
        self.h = h
        self.device = device

    def __del__(self):
        try:
            L.load().bjx_destroy(self.h)
        except Exception:
            pass


def context(device: Optional[torch.device] = None) -> _Ctx:
    if not torch.cuda.is_available():
        raise RuntimeError("bijectors_amd needs a ROCm GPU; there is no CPU fallback")
    dev = torch.cuda.current_device() if device is None else torch.device(device).index or 0
    if dev != torch.cuda.current_device():
        # the entry points launch on the context's stream and allocate their scratch lazily: both follow the CURRENT device.
        # One process per GPU (the design) never gets here; a multi-GPU process must select the device first.
        raise ValueError(f"tensor on cuda:{dev} but the current device is cuda:{torch.cuda.current_device()}: "
                         f"call torch.cuda.set_device({dev}) (one process per GPU is the supported layout)")
    stream = torch.cuda.current_stream(dev).cuda_stream
    key = (dev, stream)
    c = _ctx_cache.get(key)
    if c is None:
        c = _Ctx(dev, stream)
        _ctx_cache[key] = c
    return c


class CapturedStep:
    """`fn()` recorded ONCE into a hipGraph (bjx_graph_begin / bjx_graph_end, include/bjx.h) and replayed with
    `replay()`: for small shards a step — kernel, finalize, the 8-byte all-reduce through the library's communicator —
    costs more in per-call dispatch than on the GPU.  `fn` must read and write the same tensors on every call (their
    addresses are baked into the graph), must not synchronise with the host, and a multi-rank step must use the library
    collective (`shard.use_library_collective()`): a torch.distributed all-reduce runs on torch's own stream.

    The step runs on a private stream (the NULL stream cannot be captured); `replay()` orders it after the work already
    queued on the caller's current stream, and `wait()` makes the caller's stream wait for the replays."""

    def __init__(self, fn, device: Optional[torch.device] = None):
        self._h = C.c_void_p()
        self.stream = torch.cuda.Stream(device)
        self.stream.wait_stream(torch.cuda.current_stream(device))
        lib = L.load()
        with torch.cuda.stream(self.stream):
            fn()                                  # eager run on this stream: context, staging buffers, allocator blocks
            self.stream.synchronize()
            self._ctx = context(device)
            L.check(self._ctx.h, lib.bjx_graph_begin(self._ctx.h), "bjx_graph_begin")
            try:
                self.result = fn()                # recorded, not executed; keeps the output tensors alive
            finally:
                rc = lib.bjx_graph_end(self._ctx.h, C.byref(self._h))
            L.check(self._ctx.h, rc, "bjx_graph_end")

    def replay(self, n: int = 1):
        self.stream.wait_stream(torch.cuda.current_stream(self.stream.device))
        lib = L.load()
        for _ in range(n):
            L.check(self._ctx.h, lib.bjx_graph_launch(self._ctx.h, self._h), "bjx_graph_launch")
        return self.result

    def wait(self) -> None:
        torch.cuda.current_stream(self.stream.device).wait_stream(self.stream)

    def close(self) -> None:
        if self._h:
            self.stream.synchronize()
            L.load().bjx_graph_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def kernel_timed(fn, device: Optional[torch.device] = None):
    """-> (fn(), summed milliseconds of the DOMINANT kernels fn launched on the context stream, how many of them): the library
    brackets every hot kernel (not the parameter-prep / finalize helpers) with its own hipEvent pair between
    bjx_kernel_time_begin and _end.  The launch count is how the tests assert that a composition ran as ONE fused launch."""
    ctx = context(device)
    lib = L.load()
    L.check(ctx.h, lib.bjx_kernel_time_begin(ctx.h), "bjx_kernel_time_begin")
    try:
        out = fn()
    finally:
        ms, n = C.c_float(0), C.c_int(0)
        rc = lib.bjx_kernel_time_end(ctx.h, C.byref(ms), C.byref(n))
    L.check(ctx.h, rc, "bjx_kernel_time_end")
    return out, float(ms.value), int(n.value)


_PARAM_WATCH: dict = {}     # id(ctx) -> [epoch, last epoch handed to the library, {data_ptr: (weakref to the tensor, _version)}]
_PARAM_CACHE = {"on": False, "gen": 0}


class cache_params:
    """Opt in to the REUSE of tables derived from parameter tensors between calls: the layer-major (w, u, b) tables
    of a PlanarLayer run, the device copy of a host-resident parameter, and — through BJX_OPT_PARAM_EPOCH (include/bjx.h) — whatever
    the library keeps per parameter epoch.  OFF by default: a table is then rebuilt from the parameter arrays on every call, so a
    write the host cannot see is never missed.

    With the reuse on, staleness is decided from `tensor._version`, which torch bumps for every in-place operation on THAT tensor
    object (what `torch.optim` steps and `p.add_()` / `p.copy_()` under `no_grad` do).  It does NOT see `p.data.add_(...)` (`.data`
    is a second tensor object with its own counter), writes by other frameworks through DLPack, or raw-pointer writes — after any of
    those call `invalidate_params()`.

        bj.cache_params(True)              # process-wide
        with bj.cache_params():            # or for a region (a sampler loop whose parameters are frozen)
            ...
    """

    def __init__(self, enabled: bool = True):
        self._prev = _PARAM_CACHE["on"]
        _PARAM_CACHE["on"] = bool(enabled)
        if not enabled:
            invalidate_params()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        _PARAM_CACHE["on"] = self._prev
        if not self._prev:
            invalidate_params()
        return False


def invalidate_params() -> None:
    """Forget every table derived from parameter tensors (see `cache_params`): the next call rebuilds them from the arrays."""
    _PARAM_CACHE["gen"] += 1
    for st in _PARAM_WATCH.values():
        st[0] = st[0] + 1 if st[0] < (1 << 30) else 1
        st[2].clear()


def _note_params(ctx: "_Ctx", *tensors) -> None:
    """BJX_OPT_PARAM_EPOCH bookkeeping (include/bjx.h): the library may keep tables it derives from parameter arrays while the
    epoch is unchanged; epoch 0 (the library's default, and what this sends while `cache_params` is off) = never reuse.  With the
    reuse on: torch knows when a tensor was written (`_version`) and a weak reference tells a live tensor from a new one at a
    recycled address — so the epoch moves exactly when a parameter array that is about to be passed is not the same, unwritten
    tensor that was passed at that address before.  Conversions that make a fresh tensor on every call (a host parameter uploaded
    anew, a layout copy) change the epoch every time: no reuse, never a stale table."""
    import weakref

    st = _PARAM_WATCH.setdefault(id(ctx), [1, 0, {}])
    if not _PARAM_CACHE["on"] or not _VERSION_BUMP_OK:       # (no way to mark tensors the library wrote: never promise the library an unchanged parameter)
        if st[1] != 0:
            L.check(ctx.h, L.load().bjx_set_option(ctx.h, L.BJX_OPT_PARAM_EPOCH, 0), "bjx_set_option")
            st[1] = 0
        return
    changed = False
    for t in tensors:
        key = t.data_ptr()
        seen = st[2].get(key)
        if seen is None or seen[0]() is not t or seen[1] != t._version:
            st[2][key] = (weakref.ref(t), t._version)
            changed = seen is not None or changed           # a new address does not invalidate what is cached for the others
    if changed:
        st[0] = st[0] + 1 if st[0] < (1 << 30) else 1
    if len(st[2]) > 4096:                                   # bound the table; forgetting entries only costs a rebuild
        st[2].clear()
        st[0] = st[0] + 1 if st[0] < (1 << 30) else 1
    if st[1] != st[0]:
        L.check(ctx.h, L.load().bjx_set_option(ctx.h, L.BJX_OPT_PARAM_EPOCH, st[0]), "bjx_set_option")
        st[1] = st[0]


def _mark_written(t: Optional[torch.Tensor]) -> None:
    """The library writes through raw pointers, which torch's version counter does not see: bump it by hand for every tensor an
    entry point was asked to write IN PLACE, so that anything keyed on `_version` (torch's own autograd checks, the parameter
    tables above, the batch tag of InvertibleBatchNorm) notices."""
    if t is None:
        return
    _bump_version(t)


def _probe_version_bump():
    """The private torch call that bumps a tensor's version counter takes ([tensors], [versions]) in current builds and (tensor, int) in
    older ones (swallowing the TypeError of the wrong form made the bump a silent no-op there, and everything keyed on
    `_version` — the batch tag of InvertibleBatchNorm, the parameter tables of `cache_params` — would trust a tensor the library had
    rewritten).  Probed ONCE on a scratch CPU tensor; when neither form works the callers are told (`_VERSION_BUMP_OK` False): the
    saved-statistics shortcut and the parameter-table reuse then never trust a version."""
    fn = getattr(getattr(torch._C, "_autograd", None), "_unsafe_set_version_counter", None)
    if fn is None:
        return None
    probe = torch.zeros(1)
    for form in ("list", "scalar"):
        try:
            v0 = probe._version
            if form == "list":
                fn([probe], [v0 + 1])
            else:
                fn(probe, v0 + 1)
            if probe._version == v0 + 1:
                return form
        except Exception:
            continue
    return None


_VERSION_BUMP_FORM = _probe_version_bump()
_VERSION_BUMP_OK = _VERSION_BUMP_FORM is not None
if not _VERSION_BUMP_OK:
    import warnings

    warnings.warn("bijectors_amd: this torch build offers no way to bump a tensor's version counter; tensors written in place by the library keep their "
                  "version, so cache_params() and the saved batch statistics of InvertibleBatchNorm are disabled (always recomputed)")


def _bump_version(t: torch.Tensor) -> None:
    if _VERSION_BUMP_FORM == "list":
        torch._C._autograd._unsafe_set_version_counter([t], [t._version + 1])
    elif _VERSION_BUMP_FORM == "scalar":
        torch._C._autograd._unsafe_set_version_counter(t, t._version + 1)


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return L.BJX_F32
    if t.dtype == torch.float64:
        return L.BJX_F64
    raise TypeError(f"only Float32/Float64 arrays are supported, got {t.dtype}")


def colmajor(t: torch.Tensor) -> torch.Tensor:
    """Return `t` (shape (dim, batch)) with Julia's column-major memory layout."""
    if t.dim() != 2:
        return t.contiguous()
    if t.stride(0) == 1 and t.stride(1) == max(t.shape[0], 1) or t.shape[1] <= 1 and t.stride(0) == 1:
        return t
    return t.T.contiguous().T


def _check_dev(x: torch.Tensor):
    if not isinstance(x, torch.Tensor):
        raise TypeError("expected a torch.Tensor on a ROCm device")
    if not x.is_cuda:
        raise RuntimeError("bijectors_amd operates on ROCm device tensors only (no CPU fallback)")


def _prep(x: torch.Tensor):
    """-> (x column-major, dim, batch, is_vector)"""
    _check_dev(x)
    if x.dim() == 1:
        return x.contiguous(), x.shape[0], 1, True
    if x.dim() == 2:
        xc = colmajor(x)
        return xc, x.shape[0], x.shape[1], False
    raise ValueError("expected a vector or a (dim, batch) matrix")


_OUT_HINT: list = []  # caller-owned output buffer for the next structured launch (the `!` methods)


class _into:
    """`with _into(y):` — the next `_empty` of y's shape/dtype/layout returns `y` itself, so the
    kernel writes straight into the caller's buffer (transform!/with_logabsdet_jacobian!,
    src/interface.jl:175-218) instead of into a temporary that is copied afterwards."""

    def __init__(self, y: Optional[torch.Tensor]):
        self.y = y

    def __enter__(self):
        _OUT_HINT.append(self.y)
        return self

    def __exit__(self, *exc):
        _OUT_HINT.pop()
        return False


def _colmajor_dense(t: torch.Tensor) -> bool:
    return t.dim() == 1 and t.is_contiguous() or t.dim() == 2 and t.T.is_contiguous()


def _empty(rows: int, batch: int, like: torch.Tensor, vec: bool) -> torch.Tensor:
    if _OUT_HINT and _OUT_HINT[-1] is not None:
        h = _OUT_HINT[-1]
        want = (rows,) if vec else (rows, batch)
        if tuple(h.shape) == want and h.dtype == like.dtype and h.device == like.device and _colmajor_dense(h):
            _OUT_HINT[-1] = None  # one use per scope
            _mark_written(h)      # a caller-owned tensor written through its raw pointer
            return h
    if vec:
        return torch.empty(rows, dtype=like.dtype, device=like.device)
    return torch.empty((batch, rows), dtype=like.dtype, device=like.device).T


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class _Out:
    """Allocates the optional log-det outputs of one ABI call."""

    def __init__(self, x: torch.Tensor, batch: int, per_sample, want_ladj: bool = True, ret_vector: bool = False):
        """per_sample: False (reference shape), True (per-column vector) or "both"
        (per-column vector AND the float64 global sum, used by the sharded path)."""
        self.both = per_sample == "both"
        self.sum64 = per_sample == "sum64"          # only the float64 global sum (sharded scalar path)
        want_ps = want_ladj and not self.sum64 and (bool(per_sample) or ret_vector)
        want_sum = want_ladj and (self.both or self.sum64 or not want_ps)
        self.ps = torch.empty(batch, dtype=x.dtype, device=x.device) if want_ps else None
        self.sum = torch.empty(1, dtype=torch.float64, device=x.device) if want_sum else None
        self.dtype = x.dtype

    def scalar(self) -> torch.Tensor:
        return self.sum[0].to(self.dtype)

    def result(self, vec_scalar: bool = False):
        if self.both:
            return self.ps, self.sum
        if self.sum64:
            return self.sum
        if self.ps is not None:
            return self.ps[0] if vec_scalar else self.ps
        return self.scalar()


def _param(p, like: torch.Tensor) -> torch.Tensor:
    """device tensor for a parameter (python number / sequence / tensor)."""
    if not isinstance(p, torch.Tensor):
        p = torch.as_tensor(p, dtype=like.dtype, device=like.device)
    if p.device != like.device or p.dtype != like.dtype:
        # host-resident (or other-dtype) parameters: upload / convert ONCE per (device, dtype) and parameter version,
        # not on every call (a synchronous H2D copy per launch, and not capturable into a hipGraph); `invalidate_params()`
        # drops the copy (writes through `.data` do not move `_version`)
        cache = getattr(p, "_bjx_dev", None)
        key = (like.device, like.dtype)
        hit = cache.get(key) if cache is not None else None
        if hit is not None and hit[0] == (p._version, _PARAM_CACHE["gen"]):
            p = hit[1]
        else:
            q = p.to(device=like.device, dtype=like.dtype)
            try:
                if cache is None:
                    cache = {}
                    p._bjx_dev = cache
                cache[key] = ((p._version, _PARAM_CACHE["gen"]), q)
            except Exception:
                pass
            p = q
    if p.dim() == 2 and _colmajor_dense(p):
        return p  # already Julia-layout: no row-major round trip (2 copy kernels per call)
    return p.contiguous()


# ------------------------------------------------------------------ interface (src/interface.jl)
_BIJ_EPOCH = [0]      # bumped when a public attribute of an existing Transform is re-assigned: cached launch plans (`_fast_chain`) die with it


class Transform:
    """src/interface.jl:133-135"""

    def __setattr__(self, name, value):
        if name[0] != "_" and name in self.__dict__:      # a parameter replaced after construction (b.a = new_tensor)
            _BIJ_EPOCH[0] += 1
        object.__setattr__(self, name, value)

    def __call__(self, x):
        return transform(self, x)

    def __matmul__(self, inner):  # `outer @ inner` plays the role of Julia's `outer ∘ inner`
        return ComposedFunction(self, inner)

    def __eq__(self, other):
        return type(self) is type(other) and self._key() == other._key()

    def __hash__(self):
        return hash((type(self).__name__,))

    def _key(self):
        return ()

    # defaults: subclasses implement _wlj(x, per_sample) -> (y, ladj)
    def _wlj(self, x, per_sample: bool, want_ladj: bool = True):
        raise NotImplementedError(f"`with_logabsdet_jacobian` not implemented for {type(self).__name__}")


class Bijector(Transform):
    """src/interface.jl:271"""


def isinvertible(t) -> bool:  # interface.jl:238,273
    return isinstance(t, (Bijector, Inverse, Elementwise)) or (isinstance(t, ComposedFunction) and isinvertible(t.inner) and isinvertible(t.outer))


def isclosedform(t) -> bool:  # interface.jl:231 ; planar_layer.jl:188 ; composed.jl:2
    if isinstance(t, ComposedFunction):
        return isclosedform(t.inner) and isclosedform(t.outer)
    if isinstance(t, Inverse) and isinstance(t.orig, PlanarLayer):
        return False
    return True


class Inverse(Transform):
    """src/interface.jl:246-256"""

    def __init__(self, orig: Transform):
        if not isinvertible(orig):
            raise ValueError(f"{orig} is not invertible")
        self.orig = orig

    def _key(self):
        return (self.orig,)

    def _wlj(self, x, per_sample, want_ladj=True):
        return self.orig._wlj_inv(x, per_sample, want_ladj)


def inverse(t):
    """src/interface.jl:265-266 (+ shift.jl:12, leaky_relu.jl:16, composed inverse, stacked.jl:113-118)"""
    if t is identity:                                     # `identity` is its own bijector (stacked.jl:21-23, transformed_distribution.jl:20)
        return identity
    if type(t).__name__ in ("Stacked", "NamedStacked"):
        return t._inverse()
    if type(t).__name__ == "Columnwise":                  # interface.jl:71
        return Columnwise(inverse(t.x))
    if isinstance(t, Inverse):
        return t.orig
    if isinstance(t, ComposedFunction):
        return ComposedFunction(inverse(t.inner), inverse(t.outer))
    if isinstance(t, Elementwise):
        if t.f is exp:
            return Elementwise(log)
        if t.f is log:
            return Elementwise(exp)
        return t
    if isinstance(t, Shift):
        return Shift(-t.a if not isinstance(t.a, (list, tuple)) else [-v for v in t.a])
    if isinstance(t, LeakyReLU):
        return LeakyReLU(1.0 / t.alpha)
    if isinstance(t, SignFlip):
        return SignFlip()
    if isinstance(t, Permute):
        inv = [0] * len(t.src)
        for i, s in enumerate(t.src):
            inv[s] = i
        return Permute._from_src(inv)
    if isinstance(t, Transform):
        return Inverse(t)
    raise TypeError(f"cannot invert {t!r}")


def transform(b, x):
    """src/interface.jl:156-166"""
    if b is identity:
        return x
    return b._wlj(x, per_sample=False, want_ladj=False)[0]


def logabsdetjac(b, x):
    """src/interface.jl:183-192.  For ONE fusable elementwise chain the values are not stored at all
    (bjx_chain with y = NULL: the input is read once, half the traffic of the fused pair)."""
    ops = _fused_ops(b)
    if ops is not None:
        return _run_chain(ops, x, False, True, store=False)[1]
    if isinstance(b, _MatrixBijector):                                   # corr.jl:82-92, pd.jl:22-31: no output matrix is written
        return b._wlj(x, per_sample=False, store=False)[1]
    if isinstance(b, Inverse) and isinstance(b.orig, _MatrixBijector):   # corr.jl:81, :150 (_logabsdetjac_inv_corr)
        return b.orig._wlj_inv(x, per_sample=False, store=False)[1]
    return _shape_result(b, *b._wlj(x, per_sample=False))[1]


# ------------------------------------------------------------------ launch plans: the small-call fast path
# What a sampler calls on every log-density evaluation is the SAME chain on a small (param_dim x n_chains) array
# (src/vector/product/fill.jl:146-165, 192-213).  The general path walks the composition, builds the op list, hashes it, allocates
# three tensors and converts the Float64 sum with a torch launch: 31-41 us of host time for a 5-15 us kernel
# (profiles/r05_host_overhead.txt).  Here the op list of (bijector object, dtype, rows, device, stream, return shape) is validated
# ONCE into a `bjx_plan` (include/bjx.h "plans") kept on the bijector object; a call is then: shape / layout checks, two
# `torch.empty`, one ctypes call.  The plan holds parameter POINTERS (the tensors are kept alive next to it): values rewritten in
# place are seen by the next call; a parameter ATTRIBUTE that is re-assigned bumps `_BIJ_EPOCH` and the plan is rebuilt.
class _FastPlan:
    __slots__ = ("h", "epoch", "keep", "ctx", "f32")

    def __init__(self, h, epoch, keep, ctx, f32):
        self.h, self.epoch, self.keep, self.ctx, self.f32 = h, epoch, keep, ctx, f32

    def __del__(self):
        try:
            if self.h is not None:
                L.load().bjx_plan_destroy(self.h)
        except Exception:
            pass


_FAST_OFF = [False]       # tests / A-B: `bj._fast_plans(False)` takes every call through the general path


def _fast_plans(on: bool = True) -> None:
    _FAST_OFF[0] = not on


def _fast_build(owner, get_ops, x, dim, per_sample, key):
    def not_applicable():                              # remembered too: the next call of this shape goes straight to the general path
        fp = _FastPlan(None, _BIJ_EPOCH[0], None, None, False)
        owner.__dict__.setdefault("_fast", {})[key] = fp
        return fp

    ops = get_ops()
    if ops is None or len(ops) > L.BJX_MAX_OPS:
        return not_applicable()
    ops2 = get_ops()                                   # a parameter that is a TEMPORARY (e.g. the negated shift of an inverse) is a new tensor every time
    for (k1, a0, a1), (k2, b0, b1) in zip(ops, ops2):
        for p, q in ((a0, b0), (a1, b1)):
            if isinstance(p, torch.Tensor) and p is not q:
                return not_applicable()
    if _chain_key(ops, x, dim) is None:                # host / other-dtype / broadcast parameters: the general path converts them per call
        return not_applicable()
    ctx = context(x.device)
    arr = (L.BjxOp * max(len(ops), 1))()
    keep = []
    for i, (kind, p0, p1) in enumerate(ops):
        o = arr[i]
        o.kind, o.param_len, o.p0, o.p1, o.v0, o.v1 = kind, 0, 0.0, 0.0, None, None
        for j, p in enumerate((p0, p1)):
            if p is None:
                continue
            if isinstance(p, torch.Tensor) and p.dim() > 0 and p.numel() != 1:
                keep.append(p)
                o.param_len = dim
                setattr(o, f"v{j}", p.data_ptr())
            else:
                o.param_len = max(o.param_len, 1)
                setattr(o, f"p{j}", float(p))
    flags = L.BJX_REF_VECTOR_SCALE_LADJ if per_sample is False else 0
    h = C.c_void_p()
    L.check(ctx.h, L.load().bjx_plan_chain(ctx.h, _dt(x), arr, len(ops), dim, flags, C.byref(h)), "bjx_plan_chain")
    fp = _FastPlan(h, _BIJ_EPOCH[0], keep, ctx, x.dtype == torch.float32)
    owner.__dict__.setdefault("_fast", {})[key] = fp
    return fp


def _fast_build_stacked(owner, get_ops, x, dim, per_sample, key):
    """A `Stacked` whose segments are all elementwise chains of <= 4 ops, same height in and out: one bjx_stacked launch (bjx_plan_stacked)."""
    def not_applicable():
        fp = _FastPlan(None, _BIJ_EPOCH[0], None, None, False)
        owner.__dict__.setdefault("_fast", {})[key] = fp
        return fp

    if dim != owner.length_in or owner.length_out != dim:
        return not_applicable()
    segs_ops = [_elementwise_ops(b) for b in owner.bs]
    if any(o is None or len(o) > L.BJX_MAX_SEG_OPS for o in segs_ops):
        return not_applicable()
    segs2 = [_elementwise_ops(b) for b in owner.bs]                    # parameters that are temporaries are new tensors every time
    for o1, o2 in zip(segs_ops, segs2):
        for (k1, a0, a1), (k2, b0, b1) in zip(o1, o2):
            for p_, q_ in ((a0, b0), (a1, b1)):
                if isinstance(p_, torch.Tensor) and p_ is not q_:
                    return not_applicable()
    arr, keep = owner._segments(segs_ops, list(range(len(owner.bs))), x)
    if keep:                                                           # converted / broadcast parameters: the general path makes them per call
        return not_applicable()
    params = [p_ for o in segs_ops for (_, a0, a1) in o for p_ in (a0, a1) if isinstance(p_, torch.Tensor)]
    ctx = context(x.device)
    h = C.c_void_p()
    rc = L.load().bjx_plan_stacked(ctx.h, _dt(x), arr, len(owner.bs), dim, 0, C.byref(h))
    L.check(ctx.h, rc, "bjx_plan_stacked")
    fp = _FastPlan(h, _BIJ_EPOCH[0], params + [arr], ctx, x.dtype == torch.float32)
    owner.__dict__.setdefault("_fast", {})[key] = fp
    return fp


def _fast_chain(owner, get_ops, x, per_sample, build=None):
    """-> (y, ladj) through a cached bjx_plan, or None when this call does not qualify (then the general path runs)."""
    if _FAST_OFF[0] or not isinstance(x, torch.Tensor) or not x.is_cuda or _OUT_HINT:
        return None
    nd = x.dim()
    if nd == 2:
        dim, batch = x.shape
        if x.stride(0) != 1 or (batch > 1 and x.stride(1) != dim) or dim == 0 or batch == 0:
            return None
    elif nd == 1:
        dim, batch = x.shape[0], 1
        if x.stride(0) != 1 or dim == 0:
            return None
    else:
        return None
    dt = x.dtype
    if dt is not torch.float32 and dt is not torch.float64:
        return None
    dev = x.device.index
    if dev != torch._C._cuda_getDevice():
        return None
    key = (dt, dim, per_sample, dev, torch._C._cuda_getCurrentRawStream(dev))
    cache = owner.__dict__.get("_fast")
    fp = cache.get(key) if cache is not None else None
    if fp is None or fp.epoch != _BIJ_EPOCH[0]:
        fp = (build or _fast_build)(owner, get_ops, x, dim, per_sample, key)
    if fp.h is None:
        return None
    y = torch.empty(dim, dtype=dt, device=x.device) if nd == 1 else torch.empty((batch, dim), dtype=dt, device=x.device).T
    run = L.load().bjx_plan_run
    if per_sample:
        l = torch.empty(batch, dtype=dt, device=x.device)
        rc = run(fp.h, x.data_ptr(), y.data_ptr(), l.data_ptr(), None, None, batch)
    else:
        l = torch.empty((), dtype=dt, device=x.device)
        if fp.f32:
            rc = run(fp.h, x.data_ptr(), y.data_ptr(), None, None, l.data_ptr(), batch)       # the library writes the sum as Float32 too
        else:
            rc = run(fp.h, x.data_ptr(), y.data_ptr(), None, l.data_ptr(), None, batch)
    if rc == L.ERR_UNSUPPORTED:                         # a shape the planned entry does not take (bjx_stacked: > 2 nonlinear stages in a segment): remember, general path
        try:
            L.load().bjx_plan_destroy(fp.h)
        finally:
            fp.h = None
        return None
    if rc != 0:
        L.check(fp.ctx.h, rc, "bjx_plan_run")
    return y, l


def _fast_vjp_build(owner, get_ops, x, dim, key):
    def not_applicable():
        fp = _FastPlan(None, _BIJ_EPOCH[0], None, None, False)
        owner.__dict__.setdefault("_fast", {})[key] = fp
        return fp

    ops = get_ops()
    if ops is None or len(ops) > L.BJX_MAX_SEG_OPS:
        return not_applicable()
    ops2 = get_ops()
    for (k1, a0, a1), (k2, b0, b1) in zip(ops, ops2):
        for p, q in ((a0, b0), (a1, b1)):
            if isinstance(p, torch.Tensor) and p is not q:
                return not_applicable()
    if _chain_key(ops, x, dim) is None:
        return not_applicable()
    ctx = context(x.device)
    seg = (L.BjxSegment * 1)()
    sg = seg[0]
    sg.in_lo, sg.out_lo, sg.len, sg.n_ops = 0, 0, dim, len(ops)
    keep = []
    for k, (kind, p0, p1) in enumerate(ops):
        o = sg.ops[k]
        o.kind, o.param_len, o.p0, o.p1, o.v0, o.v1 = kind, 0, 0.0, 0.0, None, None
        for j, p in enumerate((p0, p1)):
            if p is None:
                continue
            if isinstance(p, torch.Tensor) and p.dim() > 0 and p.numel() != 1:
                keep.append(p)
                o.param_len = dim
                setattr(o, f"v{j}", p.data_ptr())
            else:
                o.param_len = max(o.param_len, 1)
                setattr(o, f"p{j}", float(p))
    h = C.c_void_p()
    L.check(ctx.h, L.load().bjx_plan_stacked_vjp(ctx.h, _dt(x), seg, 1, dim, C.byref(h)), "bjx_plan_stacked_vjp")
    fp = _FastPlan(h, _BIJ_EPOCH[0], keep, ctx, x.dtype == torch.float32)
    owner.__dict__.setdefault("_fast", {})[key] = fp
    return fp


def _fast_chain_vjp(owner, get_ops, x, out_bar, ladj_bar):
    """Input pullback of an elementwise chain through a cached pullback plan (bjx_plan_stacked_vjp); None when the call does not qualify."""
    if _FAST_OFF[0] or not isinstance(x, torch.Tensor) or not x.is_cuda or _OUT_HINT or not isinstance(out_bar, torch.Tensor):
        return None
    if ladj_bar is not None and not isinstance(ladj_bar, torch.Tensor):
        return None                                     # a python number: the general path broadcasts it
    nd = x.dim()
    if nd == 2:
        dim, batch = x.shape
        if x.stride(0) != 1 or (batch > 1 and x.stride(1) != dim) or dim == 0 or batch == 0:
            return None
    elif nd == 1:
        dim, batch = x.shape[0], 1
        if x.stride(0) != 1 or dim == 0:
            return None
    else:
        return None
    dt = x.dtype
    if (dt is not torch.float32 and dt is not torch.float64) or out_bar.dtype is not dt or out_bar.shape != x.shape or out_bar.stride() != x.stride() or out_bar.device != x.device:
        return None
    if ladj_bar is not None and (ladj_bar.dtype is not dt or ladj_bar.dim() != 1 or ladj_bar.shape[0] != batch or ladj_bar.stride(0) != 1 or ladj_bar.device != x.device):
        return None
    dev = x.device.index
    if dev != torch._C._cuda_getDevice():
        return None
    key = ("vjp", dt, dim, dev, torch._C._cuda_getCurrentRawStream(dev))
    cache = owner.__dict__.get("_fast")
    fp = cache.get(key) if cache is not None else None
    if fp is None or fp.epoch != _BIJ_EPOCH[0]:
        fp = _fast_vjp_build(owner, get_ops, x, dim, key)
    if fp.h is None:
        return None
    xb = torch.empty(dim, dtype=dt, device=x.device) if nd == 1 else torch.empty((batch, dim), dtype=dt, device=x.device).T
    rc = L.load().bjx_plan_run_vjp(fp.h, x.data_ptr(), out_bar.data_ptr(), None if ladj_bar is None else ladj_bar.data_ptr(), xb.data_ptr(), batch)
    if rc != 0:
        L.check(fp.ctx.h, rc, "bjx_plan_run_vjp")
    return xb


def with_logabsdet_jacobian(b, x, per_sample: bool = False):
    """ChangesOfVariables.with_logabsdet_jacobian for the hot-path bijectors.

    per_sample=False reproduces the reference's return shape; per_sample=True always returns the
    per-column log-det vector."""
    if (per_sample is False or per_sample is True) and isinstance(b, (ComposedFunction, _ChainOp, Elementwise, Inverse)):
        r = _fast_chain(b, lambda: _fused_ops(b), x, per_sample)
        if r is not None:
            return r
    if (per_sample is False or per_sample is True) and isinstance(b, Stacked) and isinstance(x, torch.Tensor) and x.dim() == 2:
        r = _fast_chain(b, None, x, per_sample, build=_fast_build_stacked)       # heterogeneous products: one bjx_stacked launch through a plan
        if r is not None:
            return r
    if b is identity:
        return _run_chain(_stage_ops(b), x, per_sample, True)
    y, l = b._wlj(x, per_sample=per_sample)
    if per_sample:
        return y, l
    return _shape_result(b, y, l)


def _shape_result(b, y, l):
    if isinstance(b, PlanarLayer):
        return PlanarResult(y, l)  # planar_layer.jl:109 returns a NamedTuple
    return y, l


def _fused_ops(b):
    """Op list when `b` is ONE fusable elementwise chain, else None."""
    if isinstance(b, ComposedFunction):
        ops = []
        for s in b._stages():
            o = _stage_ops(s)
            if o is None:
                return None
            ops.extend(o)
        return ops if len(ops) <= L.BJX_MAX_OPS else None
    return _stage_ops(b)


def transform_(b, x, y=None):
    """transform!(b, x[, y]) — src/interface.jl:175-176 (composed.jl:7-10 for chains)"""
    tgt = x if y is None else y
    ops = _fused_ops(b)
    if ops is not None:
        _run_chain(ops, x, False, False, out_y=tgt)
        _mark_written(tgt)
        return tgt
    with _into(tgt if tgt is not x else None):
        out = transform(b, x)
    if out.data_ptr() != tgt.data_ptr():
        tgt.copy_(out)
    return tgt


def with_logabsdet_jacobian_(b, x, y=None, logjac=0.0):
    """with_logabsdet_jacobian!(b, x[, y, logjac]) — src/interface.jl:212-218: (y, logjac + new).
    Elementwise chains write directly into `y` in one launch (composed.jl:22-25)."""
    tgt = x if y is None else y
    ops = _fused_ops(b)
    if ops is not None:
        _, l = _run_chain(ops, x, False, True, out_y=tgt)
        _mark_written(tgt)
        return tgt, (l if isinstance(logjac, float) and logjac == 0.0 else logjac + l)
    with _into(tgt if tgt is not x else None):
        out, l = with_logabsdet_jacobian(b, x)
    if out.data_ptr() != tgt.data_ptr():
        tgt.copy_(out)
    return tgt, logjac + l


def output_size(b, sz):
    """src/interface.jl:85-86 and the per-bijector overloads"""
    sz = tuple(sz)
    if isinstance(b, ComposedFunction):
        return output_size(b.outer, output_size(b.inner, sz))
    if type(b).__name__ == "Stacked":  # stacked.jl:127
        return (b.length_out,) + sz[1:]
    if isinstance(b, SimplexBijector):  # simplex.jl:6-12
        return (sz[0] - 1,) + sz[1:]
    if isinstance(b, Inverse) and isinstance(b.orig, SimplexBijector):
        return (sz[0] + 1,) + sz[1:]
    if isinstance(b, (VecCorrBijector, PDVecBijector)):  # corr.jl:150-160, pd.jl:50-60
        if len(sz) < 2 or sz[0] != sz[1]:
            raise ValueError(f"sizes should be equal; received {sz}")
        return (b._n(sz[0]),)
    if isinstance(b, Inverse) and isinstance(b.orig, (VecCorrBijector, PDVecBijector)):
        n = b.orig._K(sz[0])
        return (n, n)
    if isinstance(b, VecCholeskyBijector):  # corr.jl:256-259
        n = sz[0]
        return (n * (n - 1) // 2,)
    if isinstance(b, Inverse) and isinstance(b.orig, VecCholeskyBijector):
        n = _triu1_dim_from_length(sz[0])
        return (n, n)
    return sz


def _triu1_dim_from_length(d: int) -> int:  # src/utils.jl:99
    return (1 + math.isqrt(1 + 8 * d)) // 2


# ------------------------------------------------------------------ elementwise chain (F1)
def exp(x):  # scalar functions, used as tags by `elementwise`
    return math.exp(x)


def log(x):
    return math.log(x)


def identity(x):
    return x


class Elementwise(Transform):
    """`Base.Fix1(broadcast, f)` — src/interface.jl:6,33"""

    def __init__(self, f):
        if f not in (exp, log):
            raise NotImplementedError("only elementwise(exp) / elementwise(log) are device kernels (SURVEY.md §8b)")
        self.f = f

    def _key(self):
        return (self.f,)

    def _ops(self, inv=False):
        kind = L.OP_EXP if (self.f is exp) != inv else L.OP_LOG
        return [(kind, None, None)]

    def _wlj(self, x, per_sample, want_ladj=True):
        return _run_chain(self._ops(), x, per_sample, want_ladj)


def elementwise(f):
    """src/interface.jl:33-39"""
    if f is identity:
        return identity
    if isinstance(f, ComposedFunction):
        return ComposedFunction(elementwise(f.outer), elementwise(f.inner))
    return Elementwise(f)


class _ChainOp(Bijector):
    """Bijectors that are single ops of the fused chain kernel."""

    def _ops(self, inv=False):
        raise NotImplementedError

    def _wlj(self, x, per_sample, want_ladj=True):
        return _run_chain(self._ops(False), x, per_sample, want_ladj)

    def _wlj_inv(self, x, per_sample, want_ladj=True):
        return _run_chain(self._ops(True), x, per_sample, want_ladj)


def _is_seq(p):
    return isinstance(p, (list, tuple)) or (isinstance(p, torch.Tensor) and p.dim() >= 1)


class Shift(_ChainOp):
    """shift.jl:4-24"""

    def __init__(self, a):
        self.a = a

    def _key(self):
        return (_keyify(self.a),)

    def _ops(self, inv=False):
        a = self.a
        if inv:
            a = -a if not isinstance(a, (list, tuple)) else [-v for v in a]
        return [(L.OP_SHIFT, a, None)]


class Scale(_ChainOp):
    """scale.jl:1-36: scalar `a`, vector `a` (one value per row) and MATRIX `a` (`a * x`, `a \\ y`, logabsdet(a); :14,17,35-36).
    The matrix form is its own launch (bjx_scale_matrix: LDS-resident matrix, dim <= 128), not a stage of the fused chain."""

    def __init__(self, a, batched: bool = False):
        # `batched=True`: `a` has shape (rows, batch) and scales element-wise — only meaningful as the
        # law returned by a Coupling's θ for a batch of columns.  A plain 2-D `a` is the reference's matrix Scale.
        self.a = a
        self.matrix = isinstance(a, torch.Tensor) and a.dim() == 2 and not batched
        if self.matrix and a.shape[0] != a.shape[1]:
            raise ValueError("DimensionMismatch: Scale with a matrix parameter needs a square matrix")

    def _key(self):
        return (_keyify(self.a),)

    def _ops(self, inv=False):
        if self.matrix:
            return None
        return [(L.OP_SCALE_INV if inv else L.OP_SCALE, self.a, None)]

    def _run_matrix(self, x, inv, per_sample, want_ladj):
        xc, dim, batch, vec = _prep(x)
        a = colmajor(_param(self.a, xc))
        if a.shape[0] != dim:
            raise ValueError(f"DimensionMismatch: Scale with a {tuple(a.shape)} matrix applied to {dim} rows")
        _note_params(context(xc.device), a)      # an unchanged matrix keeps its factorisation under `cache_params` (BJX_OPT_PARAM_EPOCH)
        # scale.jl:35-36: logabsdet(a) ONCE for a matrix of columns in the reference's scalar; per-column vector otherwise
        return _call_struct("bjx_scale_matrix", x, dim, False, per_sample, want_ladj, (int(inv), _ptr(a)), (dim,),
                            flags=L.BJX_REF_VECTOR_SCALE_LADJ if per_sample is False else 0)

    def _wlj(self, x, per_sample, want_ladj=True):
        if self.matrix:
            return self._run_matrix(x, False, per_sample, want_ladj)
        return super()._wlj(x, per_sample, want_ladj)

    def _wlj_inv(self, x, per_sample, want_ladj=True):
        if self.matrix:
            return self._run_matrix(x, True, per_sample, want_ladj)
        return super()._wlj_inv(x, per_sample, want_ladj)


class Logit(_ChainOp):
    """logit.jl:4-30"""

    def __init__(self, a, b):
        self.a, self.b = a, b

    def _key(self):
        return (self.a, self.b)

    def _ops(self, inv=False):
        return [(L.OP_LOGIT_INV if inv else L.OP_LOGIT, self.a, self.b)]


class LeakyReLU(_ChainOp):
    """leaky_relu.jl:10-29"""

    def __init__(self, alpha):
        self.alpha = alpha

    def _key(self):
        return (self.alpha,)

    def _ops(self, inv=False):
        return [(L.OP_LEAKY_RELU, 1.0 / self.alpha if inv else self.alpha, None)]


class TruncatedBijector(_ChainOp):
    """truncated.jl:4-91"""

    def __init__(self, lb, ub):
        self.lb, self.ub = lb, ub

    def _key(self):
        return (_keyify(self.lb), _keyify(self.ub))

    def _ops(self, inv=False):
        return [(L.OP_TRUNCATED_INV if inv else L.OP_TRUNCATED, self.lb, self.ub)]


class SignFlip(_ChainOp):
    """ordered.jl:1-7"""

    def _ops(self, inv=False):
        return [(L.OP_SIGNFLIP, None, None)]


def _keyify(p):
    if isinstance(p, torch.Tensor):
        return tuple(p.detach().cpu().reshape(-1).tolist())
    if isinstance(p, (list, tuple)):
        return tuple(p)
    return p


class ComposedFunction(Transform):
    """Base.ComposedFunction: `outer ∘ inner` — src/bijectors/composed.jl:1-25"""

    def __init__(self, outer, inner):
        self.outer, self.inner = outer, inner

    def _key(self):
        return (self.outer, self.inner)

    def _stages(self):
        """Application order (inner first)."""
        out = []
        for part in (self.inner, self.outer):
            if isinstance(part, ComposedFunction):
                out.extend(part._stages())
            elif part is identity:
                continue
            else:
                out.append(part)
        return out

    def _plan(self):
        """(planned stages, spans) of `_planned`, kept while the chain's stage objects are the same (the `_PlanarRun`s carry the
        gathered parameter tables)."""
        stages = self._stages()
        key = tuple(id(st) for st in stages)
        hit = getattr(self, "_plan_cache", None)
        if hit is None or hit[0] != key:
            hit = (key, _planned(stages), stages)
            self._plan_cache = hit
        return hit[1]

    def _wlj(self, x, per_sample, want_ladj=True):
        # Walk the chain; maximal runs of fusable elementwise stages become ONE kernel launch, and so do runs of PlanarLayers.
        stages = self._plan()[0]
        y = x
        total = None
        run: list = []

        def flush(y, total):
            if not run:
                return y, total
            y2, l = _run_chain(list(run), y, per_sample, want_ladj)
            run.clear()
            return y2, _add_ladj(total, l)

        for s in stages:
            ops = _stage_ops(s)
            if ops is not None and len(run) + len(ops) <= L.BJX_MAX_OPS:
                run.extend(ops)
                continue
            y, total = flush(y, total)
            if ops is not None:
                run.extend(ops)
            else:
                y, l = s._wlj(y, per_sample, want_ladj)
                total = _add_ladj(total, l)
        y, total = flush(y, total)
        return y, total


def _add_ladj(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if isinstance(a, tuple):  # per_sample="both": (per-column vector, float64 sum)
        return (a[0] + b[0], a[1] + b[1])
    return a + b  # scalar + per-column vector broadcasts exactly like Julia's `+` in composed.jl:13


def _stage_ops(s):
    if s is identity:
        return [(L.OP_IDENTITY, None, None)]
    if isinstance(s, Elementwise):
        return s._ops()
    if isinstance(s, _ChainOp):
        return s._ops(False)
    if isinstance(s, Inverse) and isinstance(s.orig, _ChainOp):
        return s.orig._ops(True)
    if getattr(s, "_elementwise_stage", False):          # the scalar links of src/vector/ (vector.py: ScalarToScalarBijector)
        return s._ops(False)
    if isinstance(s, Inverse) and getattr(s.orig, "_elementwise_stage", False):
        return s.orig._ops(True)
    return None


_CHAIN_OPS: dict = {}


def _chain_key(ops, xc, dim):
    """Hashable identity of a marshalled op list, or None when it holds a temporary (a host parameter, a scalar broadcast next to a
    vector, another dtype / device / layout): kinds, scalar values, and (address, length) of device-resident vector parameters."""
    key = [dim, xc.dtype, xc.device.index]
    for kind, p0, p1 in ops:
        key.append(kind)
        seq = False
        for p in (p0, p1):
            if p is None:
                key.append(None)
            elif isinstance(p, torch.Tensor) and p.dim() > 0 and p.numel() != 1:
                if p.dtype != xc.dtype or p.device != xc.device or not p.is_contiguous() or p.numel() != dim:
                    return None
                key.append((p.data_ptr(), p.numel()))
                seq = True
            elif isinstance(p, (int, float)):
                key.append(float(p))
            else:
                return None
        if seq and any(isinstance(p, (int, float)) for p in (p0, p1)):
            return None                                   # a scalar broadcast to a temporary vector next to a vector parameter
    return tuple(key)


def _run_chain(ops: Sequence, x: torch.Tensor, per_sample: bool, want_ladj: bool = True, out_y: Optional[torch.Tensor] = None, store: bool = True, flags: int = 0):
    """One bjx_chain launch.  ops: [(kind, p0, p1)] in application order.  store=False: the values are not
    written (log-det / log-density only: half the traffic)."""
    xc, dim, batch, vec = _prep(x)
    ctx = context(xc.device)
    # The marshalled op list (a ctypes array of kinds, scalars and parameter POINTERS — no parameter values) is kept per (ops, dim,
    # dtype, device): building it costs 10-29 us of a 35-48 us call (profiles/r05_host_overhead.txt).  Only op lists whose vector
    # parameters are device tensors of the input's dtype are kept (no temporaries to keep alive); the key holds their addresses, so a
    # parameter that moved is a different key, and a pointer is all the kept array holds — nothing that could go stale.
    ckey = _chain_key(ops, xc, dim)
    hit = _CHAIN_OPS.get(ckey) if ckey is not None else None
    arr = hit if hit is not None else (L.BjxOp * max(len(ops), 1))()
    keep = []
    for i, (kind, p0, p1) in enumerate(() if hit is not None else ops):
        o = arr[i]
        o.kind, o.param_len, o.p0, o.p1, o.v0, o.v1 = kind, 0, 0.0, 0.0, None, None
        seq = any(_is_seq(p) for p in (p0, p1) if p is not None)
        for j, p in enumerate((p0, p1)):
            if p is None:
                continue
            if seq:
                t = _param(p, xc).reshape(-1) if _is_seq(p) else torch.full((dim,), float(p), dtype=xc.dtype, device=xc.device)
                if t.numel() != dim:
                    raise ValueError(f"DimensionMismatch: parameter of length {t.numel()} for input with {dim} rows")
                keep.append(t)
                o.param_len = dim
                setattr(o, f"v{j}", t.data_ptr())
            else:
                o.param_len = 1
                setattr(o, f"p{j}", float(p))
    if hit is None and ckey is not None:
        if len(_CHAIN_OPS) >= 512:
            _CHAIN_OPS.clear()
        _CHAIN_OPS[ckey] = arr
    if not store:
        y = None
    elif out_y is None:
        y = _empty(dim, batch, xc, vec)
    else:  # transform!/with_logabsdet_jacobian!: write straight into the caller's buffer (may alias x)
        y = out_y
        if y.shape != x.shape or y.dtype != xc.dtype or y.device != xc.device or (y.dim() == 2 and colmajor(y) is not y):
            raise ValueError("DimensionMismatch: output buffer must match the input's shape, dtype and column-major layout")
    out = _Out(xc, batch, per_sample, want_ladj)
    if per_sample is False:
        # reproduce scale.jl:31-32 (vector-`a` Scale on a matrix: Σ log|a_i|, NOT times batch) only in the scalar the
        # reference itself returns.  The sharded modes ("both" / "sum64") must stay additive over column blocks: there
        # ladj_sum is the mathematically consistent Σ_n ladj_ps[n], so the all-reduced value does not depend on the
        # shard count and equals sum(ladj_ps).
        flags |= L.BJX_REF_VECTOR_SCALE_LADJ
    rc = L.load().bjx_chain(ctx.h, _dt(xc), arr, len(ops), None if flags & L.BJX_INPUT_STDNORMAL else _ptr(xc), _ptr(y), _ptr(out.ps), _ptr(out.sum), dim, batch, flags)
    L.check(ctx.h, rc, "bjx_chain")
    del keep
    if not want_ladj:
        return y, None
    return y, out.result()


# ------------------------------------------------------------------ structured bijectors
def _call_struct(fn_name: str, x, rows_out: int, per_sample_ret: bool, per_sample: bool, want_ladj: bool, pre_args, post_dims, flags: int = 0, store: bool = True):
    """Shared launcher: fn(ctx, dt, *pre_args, in, out, ladj_ps, ladj_sum, *post_dims, flags)."""
    xc, dim, batch, vec = _prep(x)
    ctx = context(xc.device)
    y = _empty(rows_out, batch, xc, vec) if store else None
    out = _Out(xc, batch, per_sample, want_ladj, ret_vector=per_sample_ret)
    fn = getattr(L.load(), fn_name)
    rc = fn(ctx.h, _dt(xc), *pre_args, _ptr(xc), _ptr(y), _ptr(out.ps), _ptr(out.sum), *post_dims, batch, flags)
    L.check(ctx.h, rc, fn_name)
    if not want_ladj:
        return y, None
    return y, out.result(vec_scalar=vec and not per_sample)


class OrderedBijector(Bijector):
    """ordered.jl:9-80.  Matrix input -> per-column log-det vector (:80)."""

    def _wlj(self, x, per_sample, want_ladj=True):
        return _call_struct("bjx_ordered", x, x.shape[0], True, per_sample, want_ladj, (0,), (x.shape[0],))

    def _wlj_inv(self, x, per_sample, want_ladj=True):
        return _call_struct("bjx_ordered", x, x.shape[0], True, per_sample, want_ladj, (1,), (x.shape[0],))


class SimplexBijector(Bijector):
    """simplex.jl:4-143.  Matrix input -> scalar sum over columns (:141-143)."""

    def _wlj(self, x, per_sample, want_ladj=True):
        K = x.shape[0]
        return _call_struct("bjx_simplex", x, K - 1, False, per_sample, want_ladj, (0,), (K,))

    def _wlj_inv(self, y, per_sample, want_ladj=True):
        K = y.shape[0] + 1
        return _call_struct("bjx_simplex", y, K, False, per_sample, want_ladj, (1,), (K,))


class VecCholeskyBijector(Bijector):
    """corr.jl:164-259, batched: W is (K, K) or (K, K, batch) column-major; y is (n,) or (n, batch)."""

    def __init__(self, mode="U"):
        s = str(mode).lstrip(":")
        if s not in ("U", "L"):
            raise ValueError("mode must be either :U (upper triangular) or :L (lower triangular)")  # corr.jl:215-219
        self.mode = s

    def _key(self):
        return (self.mode,)

    def _wlj(self, W, per_sample, want_ladj=True):
        _check_dev(W)
        if W.dim() not in (2, 3) or W.shape[0] != W.shape[1]:
            raise ValueError("DimensionMismatch: expected a square (K, K[, batch]) factor")
        K = W.shape[0]
        vec = W.dim() == 2
        batch = 1 if vec else W.shape[2]
        Wc = W if vec else W.permute(2, 1, 0).contiguous().permute(2, 1, 0)  # [K,K,batch] column-major
        Wc = Wc.T.contiguous().T if vec else Wc
        ctx = context(W.device)
        n = K * (K - 1) // 2
        y = _empty(n, batch, W, vec)
        out = _Out(W, batch, per_sample, want_ladj)
        rc = L.load().bjx_vec_cholesky(ctx.h, _dt(W), 0, ord(self.mode), _ptr(Wc), _ptr(y), _ptr(out.ps), _ptr(out.sum), K, batch, 0)
        L.check(ctx.h, rc, "bjx_vec_cholesky")
        if not want_ladj:
            return y, None
        return y, out.result()

    def _wlj_inv(self, y, per_sample, want_ladj=True):
        yc, n, batch, vec = _prep(y)
        K = _triu1_dim_from_length(n)
        if K * (K - 1) // 2 != n:
            raise ValueError(f"DimensionMismatch: {n} is not a triangular number K(K-1)/2")
        ctx = context(yc.device)
        if vec:
            W = torch.empty((K, K), dtype=yc.dtype, device=yc.device).T
        else:
            W = torch.empty((batch, K, K), dtype=yc.dtype, device=yc.device).permute(2, 1, 0)
        out = _Out(yc, batch, per_sample, want_ladj)
        rc = L.load().bjx_vec_cholesky(ctx.h, _dt(yc), 1, ord(self.mode), _ptr(yc), _ptr(W), _ptr(out.ps), _ptr(out.sum), K, batch, 0)
        L.check(ctx.h, rc, "bjx_vec_cholesky")
        if not want_ladj:
            return W, None
        return W, out.result()


def _triu_dim_from_length(d: int) -> int:  # src/utils.jl:135
    return (-1 + math.isqrt(1 + 8 * d)) // 2


def _dense3(X: torch.Tensor) -> torch.Tensor:
    """(K, K[, batch]) -> the same logical array in Julia's column-major layout (strides (1, K, K*K))."""
    if X.dim() == 2:
        return X if X.T.is_contiguous() else X.T.contiguous().T
    K = X.shape[0]
    if X.stride() == (1, K, K * K) or X.shape[2] == 0:
        return X
    return X.permute(2, 1, 0).contiguous().permute(2, 1, 0)


class _MatrixBijector(Bijector):
    """Shared launcher of the matrix-variate constraint bijectors: the constrained side is a
    (K, K) matrix or a (K, K, batch) stack, column-major; the unconstrained side a (n,) / (n, batch) vector
    (`_VEC`) or a (K, K[, batch]) matrix.  The reference defines these for ONE matrix; a stack returns the
    sum of the log-dets (per_sample=True: one value per matrix), like every other bijector here."""

    _FN = ""
    _VEC = False

    @staticmethod
    def _n(K):            # packed length for a K x K matrix
        raise NotImplementedError

    @staticmethod
    def _K(n):
        raise NotImplementedError

    def _wlj(self, X, per_sample, want_ladj=True, store=True):
        _check_dev(X)
        if X.dim() not in (2, 3) or X.shape[0] != X.shape[1]:
            raise ValueError(f"DimensionMismatch: {type(self).__name__} expects a square (K, K[, batch]) matrix")   # checksquare
        K = X.shape[0]
        single = X.dim() == 2
        batch = 1 if single else X.shape[2]
        Xc = _dense3(X)
        ctx = context(X.device)
        if not store:
            y = None
        elif self._VEC:
            y = _empty(self._n(K), batch, X, single)
        else:
            y = torch.empty((K, K), dtype=X.dtype, device=X.device).T if single else torch.empty((batch, K, K), dtype=X.dtype, device=X.device).permute(2, 1, 0)
        out = _Out(X, batch, per_sample, want_ladj)
        rc = getattr(L.load(), self._FN)(ctx.h, _dt(X), 0, _ptr(Xc), _ptr(y), _ptr(out.ps), _ptr(out.sum), K, batch, 0)
        L.check(ctx.h, rc, self._FN)
        return (y, out.result(vec_scalar=single and per_sample is True)) if want_ladj else (y, None)

    def _wlj_inv(self, y, per_sample, want_ladj=True, store=True):
        _check_dev(y)
        if self._VEC:
            yc, n, batch, single = _prep(y)
            K = self._K(n)
            if self._n(K) != n:
                raise ValueError(f"DimensionMismatch: {n} is not a valid packed length for {type(self).__name__}")
        else:
            if y.dim() not in (2, 3) or y.shape[0] != y.shape[1]:
                raise ValueError(f"DimensionMismatch: inverse({type(self).__name__}) expects a square (K, K[, batch]) matrix")
            K, single = y.shape[0], y.dim() == 2
            batch = 1 if single else y.shape[2]
            yc = _dense3(y)
        ctx = context(y.device)
        if not store:
            X = None
        elif single:
            X = torch.empty((K, K), dtype=y.dtype, device=y.device).T
        else:
            X = torch.empty((batch, K, K), dtype=y.dtype, device=y.device).permute(2, 1, 0)
        out = _Out(yc, batch, per_sample, want_ladj)
        rc = getattr(L.load(), self._FN)(ctx.h, _dt(y), 1, _ptr(yc), _ptr(X), _ptr(out.ps), _ptr(out.sum), K, batch, 0)
        L.check(ctx.h, rc, self._FN)
        return (X, out.result(vec_scalar=single and per_sample is True)) if want_ladj else (X, None)


class VecCorrBijector(_MatrixBijector):
    """corr.jl:94-162 — correlation matrix <-> unconstrained vector of length K(K-1)/2 (what `bijector(::LKJ)` returns)."""
    _FN, _VEC = "bjx_vec_corr", True
    _n = staticmethod(lambda K: K * (K - 1) // 2)
    _K = staticmethod(_triu1_dim_from_length)


class CorrBijector(_MatrixBijector):
    """corr.jl:1-92 — correlation matrix <-> strictly upper triangular unconstrained matrix."""
    _FN, _VEC = "bjx_corr", False


class PDBijector(_MatrixBijector):
    """pd.jl:1-36 — positive definite matrix <-> lower triangular matrix with the log of the Cholesky diagonal."""
    _FN, _VEC = "bjx_pd", False


class PDVecBijector(_MatrixBijector):
    """pd.jl:38-60 — positive definite matrix <-> unconstrained vector of length K(K+1)/2."""
    _FN, _VEC = "bjx_pd_vec", True
    _n = staticmethod(lambda K: K * (K + 1) // 2)
    _K = staticmethod(_triu_dim_from_length)


class Permute(Bijector):
    """permute.jl:85-157.  Built from an index vector, pairs or a permutation matrix (1-based like Julia)."""

    def __init__(self, arg, *pairs):
        if isinstance(arg, int) and pairs:  # Permute(n, src => dst, ...) : permute.jl:102-150
            n = arg
            dst_of = list(range(n))
            dests, sources = set(), set()
            for src, dst in pairs:
                srcs = src if isinstance(src, (list, tuple)) else [src]
                dsts = dst if isinstance(dst, (list, tuple)) else [dst]
                if len(srcs) != len(dsts):
                    raise ValueError(f"{srcs} => {dsts} is not bijective")
                for s_, d_ in zip(srcs, dsts):
                    if d_ in dests or s_ in sources:
                        raise ValueError(f"{s_} => {d_}: index used more than once")
                    dests.add(d_)
                    sources.add(s_)
                    dst_of[s_ - 1] = d_ - 1
            if dests != sources:
                raise ValueError(f"{sources} ∩ {dests} ≠ {sources} ∪ {dests}")
            src_of = [0] * n
            for s_, d_ in enumerate(dst_of):
                src_of[d_] = s_
            self.src = src_of
        elif isinstance(arg, (list, tuple)) and arg and isinstance(arg[0], (list, tuple)):  # matrix A: y = A x
            self.src = [row.index(1) for row in [list(map(int, r)) for r in arg]]
        else:  # Permute(indices): A[idx, i] = 1  =>  y[idx_i] = x[i]   (permute.jl:90-100)
            idx = [int(i) - 1 for i in arg]
            src = [0] * len(idx)
            for i, d_ in enumerate(idx):
                src[d_] = i
            self.src = src
        if sorted(self.src) != list(range(len(self.src))):
            raise ValueError("not a permutation")
        self._dev = {}

    @classmethod
    def _from_src(cls, src):
        p = cls.__new__(cls)
        p.src = list(src)
        p._dev = {}
        return p

    def _key(self):
        return (tuple(self.src),)

    def _wlj(self, x, per_sample, want_ladj=True):
        xc, dim, batch, vec = _prep(x)
        if dim != len(self.src):
            raise ValueError(f"DimensionMismatch: permutation of {len(self.src)} rows applied to {dim} rows")
        ctx = context(xc.device)
        src = self._dev.get(xc.device)
        if src is None:
            src = torch.tensor(self.src, dtype=torch.int32, device=xc.device)
            self._dev[xc.device] = src
        y = _empty(dim, batch, xc, vec)
        rc = L.load().bjx_permute(ctx.h, _dt(xc), _ptr(src), _ptr(xc), _ptr(y), dim, batch)
        L.check(ctx.h, rc, "bjx_permute")
        if not want_ladj:
            return y, None
        if per_sample == "both":
            return y, (torch.zeros(batch, dtype=xc.dtype, device=xc.device), torch.zeros(1, dtype=torch.float64, device=xc.device))
        if per_sample == "sum64":
            return y, torch.zeros(1, dtype=torch.float64, device=xc.device)
        z = torch.zeros(batch if per_sample else (), dtype=xc.dtype, device=xc.device)  # permute.jl:155
        return y, z


class PlanarLayer(Bijector):
    """planar_layer.jl:12-188.  `w`, `u`: (dim,) tensors, `b`: 1-element tensor.
    A composition of layers written the reference's way, `l8 @ ... @ l1` (docs/src/flows.md:115), is ONE launch: the
    composition planner (`_planned`) merges the run into a `_PlanarRun`.  `PlanarLayer.stack([...])` builds the same object."""

    def __init__(self, w, u, b):
        self.w = torch.as_tensor(w)
        self.u = torch.as_tensor(u)
        self.b = torch.as_tensor(b).reshape(-1)
        self.n_layers = 1 if self.w.dim() == 1 else self.w.shape[1]
        self._tab = None

    @classmethod
    def stack(cls, layers):
        """layer[-1] ∘ ... ∘ layer[0] evaluated by ONE kernel."""
        return _PlanarRun(layers)

    def _key(self):
        return (_keyify(self.w), _keyify(self.u), _keyify(self.b))

    def _single_layers(self):
        """The stack as one-layer PlanarLayers in application order."""
        if self.n_layers == 1 and self.w.dim() == 1:
            return [self]
        return [PlanarLayer(self.w[:, k], self.u[:, k], self.b[k:k + 1]) for k in range(self.n_layers)]

    def _tables(self, xc, dim):
        """(w, u, b) on xc's device as the layer-major tables bjx_planar takes.  A (dim, n_layers) parameter matrix is transposed
        once per parameter version, not per call."""
        w, u, b = _param(self.w, xc), _param(self.u, xc), _param(self.b, xc)
        if w.numel() != dim * self.n_layers or u.numel() != w.numel():
            raise ValueError(f"DimensionMismatch: PlanarLayer of dimension {w.numel() // self.n_layers} applied to {dim} rows")
        if w.dim() != 2:
            return w, u, b
        key = (w.data_ptr(), w._version, u.data_ptr(), u._version, _PARAM_CACHE["gen"])
        if not _PARAM_CACHE["on"] or self._tab is None or self._tab[0] != key:
            self._tab = (key, w.T.contiguous(), u.T.contiguous(), (w, u))
        return self._tab[1], self._tab[2], b

    def _run(self, x, inv, per_sample, want_ladj, flags=0, store=True):
        xc, dim, batch, vec = _prep(x)
        w, u, b = self._tables(xc, dim)
        return _call_struct("bjx_planar", x, dim, True, per_sample, want_ladj,
                            (int(inv), _ptr(w), _ptr(u), _ptr(b), self.n_layers), (dim,), flags=flags, store=store)

    def _wlj(self, x, per_sample, want_ladj=True):
        return self._run(x, False, per_sample, want_ladj)

    def _wlj_inv(self, x, per_sample, want_ladj=True):
        return self._run(x, True, per_sample, want_ladj)


class _PlanarRun(PlanarLayer):
    """A maximal run of PlanarLayer stages of a composition, in APPLICATION order (layers[0] first) — what the composition
    planner hands to one bjx_planar / bjx_planar_vjp / bjx_planar_vjp_params launch.  The layers keep their own parameter tensors
    (they are what a training loop updates); the layer-major tables the kernel reads are gathered on the device by
    bjx_pack_vectors and rebuilt only when a parameter tensor changed (`_version`) or moved."""

    def __init__(self, layers):
        flat = []
        for l in layers:
            if not isinstance(l, PlanarLayer):
                raise TypeError(f"PlanarLayer.stack: {l!r} is not a PlanarLayer")
            flat.extend(l._single_layers())
        if not flat:
            raise ValueError("PlanarLayer.stack: no layers")
        self.layers = flat
        self.n_layers = len(flat)
        self._tab = None
        self._srcs = None
        self._bufs = {}

    # (dim, n_layers) views of the parameters, for code that treats the run as one stacked layer (oracle comparisons, _key)
    @property
    def w(self):
        return torch.stack([l.w.reshape(-1) for l in self.layers], dim=1)

    @property
    def u(self):
        return torch.stack([l.u.reshape(-1) for l in self.layers], dim=1)

    @property
    def b(self):
        return torch.cat([l.b.reshape(-1)[:1] for l in self.layers])

    def _single_layers(self):
        return list(self.layers)

    def _tables(self, xc, dim):
        """Layer-major (w, u, b) tables of the run.  Two things are kept on the run object, neither of which can go stale: the
        POINTER ARRAYS of the layers' parameter tensors (re-read when a layer's attribute is another tensor object or its storage
        moved) and the destination buffers per (stream, dtype, dim).  The VALUES are gathered on the device on every call — two
        launches (w and u together, then b) — unless `cache_params` is on and torch's version counters stand still."""
        srcs = self._srcs
        fresh = srcs is None or srcs[0] != xc.device or srcs[1] != xc.dtype
        if not fresh:
            for l, (tw, tu, tb) in zip(self.layers, srcs[2]):
                if l.w is not tw[0] or l.u is not tu[0] or l.b is not tb[0] or tw[0].data_ptr() != tw[2] or tu[0].data_ptr() != tu[2] or tb[0].data_ptr() != tb[2]:
                    fresh = True
                    break
        if fresh:
            per = []
            for l in self.layers:
                ent = []
                for t in (l.w, l.u, l.b):
                    c = _param(t, xc).reshape(-1)
                    ent.append((t, c, t.data_ptr() if isinstance(t, torch.Tensor) else None, c.data_ptr()))
                per.append(tuple(ent))
            for tw, tu, _ in per:
                if tw[1].numel() != dim or tu[1].numel() != dim:
                    raise ValueError(f"DimensionMismatch: PlanarLayer of dimension {tw[1].numel()} applied to {dim} rows")
            n = self.n_layers
            wu_ptrs = (C.c_void_p * (2 * n))(*([e[0][3] for e in per] + [e[1][3] for e in per]))
            b_ptrs = (C.c_void_p * n)(*[e[2][3] for e in per])
            # a host-resident / other-dtype parameter is converted by `_param` on every sighting: its device copy is a temporary, so such
            # runs are re-read on every call (never kept)
            stable = all(isinstance(e[k][0], torch.Tensor) and e[k][1].data_ptr() == e[k][2] for e in per for k in range(3))
            srcs = (xc.device, xc.dtype, per, wu_ptrs, b_ptrs, dim)
            self._srcs = srcs if stable else None
        per, wu_ptrs, b_ptrs = srcs[2], srcs[3], srcs[4]
        if srcs[5] != dim:
            raise ValueError(f"DimensionMismatch: PlanarLayer of dimension {srcs[5]} applied to {dim} rows")
        n = self.n_layers
        if _PARAM_CACHE["on"]:
            key = (xc.device, xc.dtype, dim, _PARAM_CACHE["gen"], tuple((e[k][3], e[k][1]._version, e[k][0]._version if isinstance(e[k][0], torch.Tensor) else 0) for e in per for k in range(3)))
            if self._tab is not None and self._tab[0] == key:
                return self._tab[1]
        ctx = context(xc.device)
        bkey = (id(ctx), xc.dtype, dim)
        buf = self._bufs.get(bkey)
        if buf is None or _PARAM_CACHE["on"]:          # (a kept table must not be overwritten by the next gather: fresh buffers under cache_params)
            buf = (torch.empty(2 * n * dim, dtype=xc.dtype, device=xc.device), torch.empty(n, dtype=xc.dtype, device=xc.device))
            if not _PARAM_CACHE["on"]:
                self._bufs = {bkey: buf}                # one (stream, dtype, dim) at a time: launches on a stream are ordered, so the next gather cannot overtake a reader
        wu, b = buf
        lib = L.load()
        L.check(ctx.h, lib.bjx_pack_vectors(ctx.h, _dt(xc), 2 * n, wu_ptrs, dim, _ptr(wu)), "bjx_pack_vectors")
        L.check(ctx.h, lib.bjx_pack_vectors(ctx.h, _dt(xc), n, b_ptrs, 1, _ptr(b)), "bjx_pack_vectors")
        tabs = (wu[:n * dim], wu[n * dim:], b)
        if (n * dim * xc.element_size()) % 16:          # the û table would start off a 16-byte boundary (the kernels' vector paths ask for one): its own buffer
            u2 = torch.empty(n * dim, dtype=xc.dtype, device=xc.device)
            u2.copy_(tabs[1])
            tabs = (tabs[0], u2, b)
        if _PARAM_CACHE["on"]:
            self._tab = (key, tabs, per)
        return tabs


def _planar_kind(s) -> int:
    """+1: a PlanarLayer stage, -1: inverse(PlanarLayer), 0: anything else."""
    if isinstance(s, PlanarLayer):
        return 1
    if isinstance(s, Inverse) and isinstance(s.orig, PlanarLayer):
        return -1
    return 0


def _planned(stages):
    """The composition planner (src/bijectors/composed.jl:4-25 applies a chain stage by stage; docs/src/flows.md:115 is how the
    reference writes a flow).  `stages` in application order -> (planned stages, spans): every maximal run of PlanarLayer stages
    becomes one `_PlanarRun` (ONE bjx_planar launch over the batch instead of one per layer: 1 028 instead of 8 224 B/sample for
    eight layers at dim 128), a run of inverse(PlanarLayer) stages the inverse of the reversed run; everything else stays.
    spans[i] = (lo, hi): planned stage i covers stages[lo:hi]."""
    out, spans, i = [], [], 0
    while i < len(stages):
        kind = _planar_kind(stages[i])
        j = i + 1
        if kind:
            while j < len(stages) and _planar_kind(stages[j]) == kind:
                j += 1
        if j - i == 1:
            out.append(stages[i])
        elif kind == 1:
            out.append(_PlanarRun(stages[i:j]))
        else:       # inv(l_a) applied first, then inv(l_b), ... = inverse(l_a ∘ l_b ∘ ...): the forward run applies the LAST stage's layer first
            out.append(Inverse(_PlanarRun([st.orig for st in reversed(stages[i:j])])))
        spans.append((i, j))
        i = j
    return out, spans


class RadialLayer(Bijector):
    """radial_layer.jl:11-133"""

    def __init__(self, alpha_, beta, z_0):
        self.alpha_ = torch.as_tensor(alpha_).reshape(-1)
        self.beta = torch.as_tensor(beta).reshape(-1)
        self.z_0 = torch.as_tensor(z_0)

    def _key(self):
        return (_keyify(self.alpha_), _keyify(self.beta), _keyify(self.z_0))

    def _run(self, x, inv, per_sample, want_ladj):
        xc, dim, batch, vec = _prep(x)
        z0 = _param(self.z_0, xc)
        if z0.numel() != dim:
            raise ValueError(f"DimensionMismatch: RadialLayer of dimension {z0.numel()} applied to {dim} rows")
        a, be = _param(self.alpha_, xc), _param(self.beta, xc)
        return _call_struct("bjx_radial", x, dim, True, per_sample, want_ladj, (int(inv), _ptr(a), _ptr(be), _ptr(z0)), (dim,))

    def _wlj(self, x, per_sample, want_ladj=True):
        return self._run(x, False, per_sample, want_ladj)

    def _wlj_inv(self, x, per_sample, want_ladj=True):
        return self._run(x, True, per_sample, want_ladj)


class _BatchTag:
    """Identity of the batch a training-mode forward call of InvertibleBatchNorm saw: the tensor OBJECT (weak reference — an address
    is recycled by the allocator, an object is not) and its version counter (in-place writes, including the library's own, which
    `_mark_written` reports).  `matches(x)` is False for any other tensor, for the same tensor after a write, and once the
    forward call's tensor is gone — the pullback then RECOMPUTES the batch statistics from the x it was given (one more pass
    over x; exact, since the statistics are a function of x alone) instead of trusting, or refusing, the saved ones."""
    __slots__ = ("ref", "version", "shape", "dtype")

    def __init__(self, x):
        import weakref

        self.ref, self.version, self.shape, self.dtype = weakref.ref(x), x._version, tuple(x.shape), x.dtype

    def matches(self, x) -> bool:
        return _VERSION_BUMP_OK and self.ref() is x and self.version == x._version and self.shape == tuple(x.shape) and self.dtype == x.dtype


def _batch_tag(x):
    return _BatchTag(x)


_TRAINING = [False]
_BN_RECOMPUTE = [False]      # True while a pullback re-runs forward stages: training-mode BatchNorm keeps its moving statistics


def istraining() -> bool:
    """normalise.jl:7 — the reference switches modes by redefining this global function."""
    return _TRAINING[0]


class training:
    """`with bj.training():` — the scope in which `istraining()` is true (normalise.jl:7,51)."""

    def __init__(self, on: bool = True):
        self.on = on

    def __enter__(self):
        self.prev = _TRAINING[0]
        _TRAINING[0] = self.on
        return self

    def __exit__(self, *exc):
        _TRAINING[0] = self.prev
        return False


class InvertibleBatchNorm(Bijector):
    """normalise.jl:9-92.  Eval mode (`istraining() == false`, :7) uses the moving statistics; inside
    `with bj.training():` the batch statistics are used and `m`, `v` are updated in place (:51-60) — the
    struct is mutable in the reference too."""

    def __init__(self, chs_or_b, logs=None, m=None, v=None, eps=1e-5, mtm=0.1, dtype=torch.float32, sync=None):
        # sync: None -> the statistics of THIS process's batch (the reference's behaviour); True -> the batch is sharded over the
        # default torch.distributed group; a ProcessGroup -> over that group.  Every rank of the group must then call the
        # bijector in lockstep (one all-reduce of 2·dim+1 Float64 sums per training-mode call).
        self.sync = sync
        if isinstance(chs_or_b, int):  # normalise.jl:26-37
            c = chs_or_b
            self.b, self.logs = torch.zeros(c, dtype=dtype), torch.zeros(c, dtype=dtype)
            self.m, self.v = torch.zeros(c, dtype=dtype), torch.ones(c, dtype=dtype)
        else:
            self.b, self.logs, self.m, self.v = (torch.as_tensor(t) for t in (chs_or_b, logs, m, v))
        self.eps, self.mtm = float(eps), float(mtm)

    def _key(self):
        return tuple(_keyify(t) for t in (self.b, self.logs, self.m, self.v)) + (self.eps, self.mtm)

    def _run(self, x, inv, per_sample, want_ladj):
        _check_dev(x)
        if x.dim() < 2:
            raise ValueError("InvertibleBatchNorm needs an input with at least 2 dimensions")
        if x.shape[-2] != self.b.numel():  # normalise.jl:43-45
            raise RuntimeError(f"InvertibleBatchNorm expected {self.b.numel()} channels, got {x.shape[-2]}")
        dim = x.shape[0]
        if istraining():
            if inv:
                raise AssertionError("`with_logabsdet_jacobian(::Inverse{InvertibleBatchNorm})` is only available in test mode.")  # :71
            if x.dim() != 2:
                raise NotImplementedError("training mode: (channels, batch) matrices only")
            # the moving statistics live on the device from now on and are updated in place (:58-59)
            self.m, self.v = _param(self.m, x).clone() if not (self.m.is_cuda and self.m.dtype == x.dtype) else self.m, \
                _param(self.v, x).clone() if not (self.v.is_cuda and self.v.dtype == x.dtype) else self.v
            b, logs = _param(self.b, x), _param(self.logs, x)
            # statistics of this rank's columns -> ONE sum all-reduce of 2·dim+1 Float64 values when the batch is sharded
            # (SURVEY.md §8e "Exception"; torch.distributed or the library's communicator) -> update + transform
            stats = self.batch_stats(x)
            grp = self._sync_group()
            if grp is not False:
                from . import shard as _shard

                _shard.allreduce_logabsdetjac(stats, grp)
            return self.apply_batch_stats(x, stats, per_sample, want_ladj)
        ps = [_param(t, x) for t in (self.b, self.logs, self.m, self.v)]
        return _call_struct("bjx_batchnorm", x, dim, True, per_sample, want_ladj,
                            (int(inv), *[_ptr(p) for p in ps], self.eps), (dim,))

    _warned_unsynced = [False]

    def _sync_group(self):
        """False: statistics of this process's columns only; None: all-reduce over the default group; a ProcessGroup: over it.
        `sync=None` (unspecified) inside a multi-process job is almost always a sharded batch whose ranks would silently
        normalise with different statistics and let their m / v replicas drift apart: warn once and say how to choose."""
        if self.sync is None:
            import torch.distributed as dist

            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and not InvertibleBatchNorm._warned_unsynced[0]:
                import warnings

                InvertibleBatchNorm._warned_unsynced[0] = True
                warnings.warn("InvertibleBatchNorm(sync=None) in a multi-process job: training mode normalises with THIS rank's batch statistics "
                              "and every rank updates its own m / v.  Pass sync=True (batch sharded over the default group: one all-reduce of the "
                              "2·dim+1 sums per call), a ProcessGroup, or sync=False to keep per-rank statistics on purpose.", RuntimeWarning, stacklevel=4)
            return False
        if self.sync is False:
            return False
        return None if self.sync is True else self.sync

    def batch_stats(self, x):
        """(Σ(x − m), Σ(x − m)², n) of these columns as a float64 tensor of 2·dim+1 entries (bjx_batchnorm_stats); sums of
        column blocks add up to the sums of the whole batch, which is all a sharded batch has to exchange."""
        xc, dim, batch, _ = _prep(x)
        if not (self.m.is_cuda and self.m.dtype == xc.dtype):
            self.m, self.v = _param(self.m, xc).clone(), _param(self.v, xc).clone()
        ctx = context(xc.device)
        stats = torch.empty(2 * dim + 1, dtype=torch.float64, device=xc.device)
        L.check(ctx.h, L.load().bjx_batchnorm_stats(ctx.h, _dt(xc), _ptr(self.m), _ptr(xc), _ptr(stats), dim, batch), "bjx_batchnorm_stats")
        return stats

    def apply_batch_stats(self, x, stats, per_sample=False, want_ladj=True):
        """Training-mode transform of these columns with the statistics behind the GLOBAL sums `stats`; updates m, v in place
        (call it once per rank: every rank holds its own replica of the moving statistics)."""
        b, logs = _param(self.b, x), _param(self.logs, x)
        dim = x.shape[0]
        # the batch statistics the launch below normalises with (shift = the moving mean BEFORE its update): kept for the
        # training-mode pullback (bjx_batchnorm_train_vjp); dim-sized device arithmetic, Float64
        n = stats[2 * dim]
        s1, s2 = stats[:dim] / n, stats[dim:2 * dim] / n
        # … tagged with the batch they belong to: the pullback refuses any other (two forward calls before one backward — micro-batches,
        # a validation batch, the layer used twice in a flow — would otherwise give wrong gradients without an error)
        self._batch_stats = ((self.m.double() + s1).to(x.dtype), (s2 - s1 * s1).to(x.dtype), _batch_tag(x))
        # (a RECOMPUTATION of a forward pass — the stage inputs a composition's pullback needs — must not move m / v a second time:
        #  momentum 0 leaves them as they are)
        mtm = 0.0 if _BN_RECOMPUTE[0] else self.mtm
        return _call_struct("bjx_batchnorm_train_apply", x, dim, True, per_sample, want_ladj,
                            (_ptr(b), _ptr(logs), _ptr(self.m), _ptr(self.v), self.eps, mtm, _ptr(stats)), (dim,))

    def _wlj(self, x, per_sample, want_ladj=True):
        return self._run(x, False, per_sample, want_ladj)

    def _wlj_inv(self, x, per_sample, want_ladj=True):
        return self._run(x, True, per_sample, want_ladj)


class RationalQuadraticSpline(Bijector):
    """rational_quadratic_spline.jl:75-367 with matrix parameters (dim, K+1), batched over columns.

    RationalQuadraticSpline(widths, heights, derivatives)          — knot arrays as in :79-97
    RationalQuadraticSpline(widths, heights, derivatives, B)       — raw (dim,K),(dim,K),(dim,K-1) via :109-123
    """

    def __init__(self, widths, heights, derivatives, B=None):
        self._raw = None
        w, h, d = (torch.as_tensor(t) for t in (widths, heights, derivatives))
        if w.dim() == 1:
            w, h, d = w[None, :], h[None, :], d[None, :]
        if B is not None:
            if not w.is_cuda:
                raise RuntimeError("the B-constructor runs on the device: pass ROCm tensors")
            K = w.shape[1]
            if h.shape != w.shape or d.shape != (w.shape[0], K - 1):
                raise ValueError("DimensionMismatch: expected widths/heights (dim,K) and derivatives (dim,K-1)")
            dim = w.shape[0]
            ctx = context(w.device)
            rw, rh, rd = (colmajor(t) for t in (w, h, d.to(w.dtype)))
            outs = [torch.empty((K + 1, dim), dtype=w.dtype, device=w.device).T for _ in range(3)]
            rc = L.load().bjx_rqs_params(ctx.h, _dt(w), _ptr(rw), _ptr(rh), _ptr(rd), K, dim, float(B), *[_ptr(o) for o in outs])
            L.check(ctx.h, rc, "bjx_rqs_params")
            self._raw = (rw, rh, rd, float(B))      # for the pullback onto the unconstrained parameters (vjp_params)
            w, h, d = outs
        else:
            if not (w.shape[1] == h.shape[1] == d.shape[1]):  # :93
                raise AssertionError("widths, heights and derivatives need the same number of knots")
            if not bool((d > 0).all()):  # :94
                raise AssertionError("derivatives need to be positive")
        self.widths, self.heights, self.derivatives = w, h, d

    def _key(self):
        return tuple(_keyify(t) for t in (self.widths, self.heights, self.derivatives))

    def _run(self, x, inv, per_sample, want_ladj):
        xc, dim, batch, vec = _prep(x)
        if dim != self.widths.shape[0]:
            raise ValueError(f"DimensionMismatch: spline with {self.widths.shape[0]} rows applied to {dim} rows")
        w, h, d = (colmajor(_param(t, xc)) for t in (self.widths, self.heights, self.derivatives))
        _note_params(context(xc.device), w, h, d)           # under `cache_params` an unchanged spline keeps its LDS blob (BJX_OPT_PARAM_EPOCH)
        return _call_struct("bjx_rqs", x, dim, False, per_sample, want_ladj,
                            (int(inv), _ptr(w), _ptr(h), _ptr(d), int(self.widths.shape[1])), (dim,))

    def _wlj(self, x, per_sample, want_ladj=True):
        return self._run(x, False, per_sample, want_ladj)

    def _wlj_inv(self, x, per_sample, want_ladj=True):
        return self._run(x, True, per_sample, want_ladj)


class PartitionMask:
    """coupling.jl:51-118 with index lists instead of one-hot sparse matrices (1-based indices)."""

    def __init__(self, n: int, indices_1, indices_2=None, indices_3=None):
        i1 = [int(i) for i in indices_1]
        if indices_2 is None and indices_3 is None:  # :105-113: split, x_3 empty
            i2 = [i for i in range(1, n + 1) if i not in set(i1)]
            i3 = []
        elif indices_3 is None:  # :83-90
            i2 = [int(i) for i in indices_2]
            i3 = [i for i in range(1, n + 1) if i not in set(i1) | set(i2)]
        elif indices_2 is None:  # :92-99
            i3 = [int(i) for i in indices_3]
            i2 = [i for i in range(1, n + 1) if i not in set(i1) | set(i3)]
        else:
            i2, i3 = [int(i) for i in indices_2], [int(i) for i in indices_3]
        self.n, self.indices_1, self.indices_2, self.indices_3 = n, i1, i2, i3

    def partition(self, x):  # coupling.jl:132-134
        idx = lambda l: torch.tensor([i - 1 for i in l], dtype=torch.long, device=x.device)
        return x[idx(self.indices_1)], x[idx(self.indices_2)], x[idx(self.indices_3)]

    def rows2(self, x):
        """x_2 = A_2' x (coupling.jl:133): a view when indices_2 is a row range, one gather otherwise."""
        i2 = self.indices_2
        if i2 and i2 == list(range(i2[0], i2[0] + len(i2))):
            return x[i2[0] - 1:i2[0] - 1 + len(i2)]
        cache = self.__dict__.setdefault("_dev2", {})
        t = cache.get(x.device)
        if t is None:
            t = cache[x.device] = torch.tensor([i - 1 for i in i2], dtype=torch.long, device=x.device)
        return x[t]

    def idx1_dev(self, device):
        cache = self.__dict__.setdefault("_dev1", {})
        t = cache.get(device)
        if t is None:
            t = cache[device] = torch.tensor([i - 1 for i in self.indices_1], dtype=torch.int32, device=device)
        return t


class Coupling(Bijector):
    """coupling.jl:174-259.  θ maps x₂ (rows of partition 2, shape (n2[, batch])) to a bijector for x₁.
    θ runs on the host side (it is an arbitrary closure, SURVEY.md §8b last row); the laws it may
    return that have device kernels are `Shift`, `Scale`, `Shift @ Scale` with parameters of shape
    (n1[, batch]) and `RationalQuadraticSpline` with (n1, K+1) knots."""

    def __init__(self, theta, mask):
        if isinstance(mask, int):  # coupling.jl:183-186
            mask = PartitionMask(mask, list(range(1, mask // 2 + 1)))
        self.theta, self.mask = theta, mask

    def _run(self, x, inv, per_sample, want_ladj):
        xc, dim, batch, vec = _prep(x)
        if dim != self.mask.n:
            raise ValueError(f"DimensionMismatch: mask for {self.mask.n} rows applied to {dim} rows")
        # only x_2 feeds θ (coupling.jl:210, :240); x_1 and x_3 are read in place by the kernel.  A row-range partition is a
        # strided view (no gather kernel); scattered indices cost one index gather of x_2.
        law = self.theta(self.mask.rows2(xc))
        idx1 = self.mask.idx1_dev(xc.device)
        n1 = idx1.numel()
        if isinstance(law, RationalQuadraticSpline):
            w, h, d = (colmajor(_param(t, xc)) for t in (law.widths, law.heights, law.derivatives))
            return _call_struct("bjx_coupling_rqs", x, dim, False, per_sample, want_ladj,
                                (int(inv), _ptr(idx1), n1, _ptr(w), _ptr(h), _ptr(d), int(law.widths.shape[1])), (dim,))
        scale = shift = None
        stages = law._stages() if isinstance(law, ComposedFunction) else [law]
        for s in stages:
            if isinstance(s, Scale) and scale is None and shift is None:
                scale = s.a
            elif isinstance(s, Shift) and shift is None:
                shift = s.a
            else:
                raise NotImplementedError(f"coupling law {law!r} has no device kernel (supported: Shift, Scale, Shift∘Scale, RQS)")

        def full(p, bcast_flag):
            """-> (device array, flag): a scalar or (n1,) parameter stays T[n1] (broadcast over the columns inside the kernel)."""
            if p is None:
                return None, 0
            t = _param(p, xc)
            if t.dim() == 0:
                t = t.expand(n1).contiguous()
            if t.dim() == 1:
                if t.numel() != n1:
                    raise ValueError(f"DimensionMismatch: coupling parameter of length {t.numel()} for {n1} rows")
                return t.contiguous(), (0 if vec else bcast_flag)
            if tuple(t.shape) != (n1, batch):
                raise ValueError(f"DimensionMismatch: coupling parameter of shape {tuple(t.shape)} for ({n1}, {batch})")
            return colmajor(t), 0

        (s_t, f_s), (t_t, f_t) = full(scale, L.BJX_COUPLING_SCALE_BCAST), full(shift, L.BJX_COUPLING_SHIFT_BCAST)
        return _call_struct("bjx_coupling_affine", x, dim, False, per_sample, want_ladj,
                            (int(inv), _ptr(idx1), n1, _ptr(s_t), _ptr(t_t)), (dim,), flags=f_s | f_t)

    def _wlj(self, x, per_sample, want_ladj=True):
        return self._run(x, False, per_sample, want_ladj)

    def _wlj_inv(self, x, per_sample, want_ladj=True):
        return self._run(x, True, per_sample, want_ladj)

    def _vjp(self, x, out_bar, ladj_bar, inv):
        """Pullback of the affine coupling (bjx_coupling_affine_vjp): x̄ with the x₁ rows scaled and the rest passed
        through, plus — when θ is made of torch operations — the part that flows back through θ's outputs
        (s̄, t̄ from the kernel, θ's own pullback by torch.autograd on the host: θ is an arbitrary closure)."""
        xc, dim, batch, vec = _prep(x)
        gc, gdim, gbatch, _ = _prep(out_bar)
        if dim != self.mask.n or (gdim, gbatch) != (dim, batch) or gc.dtype != xc.dtype:
            raise ValueError("DimensionMismatch: out_bar must have the shape and dtype of the output")
        i2 = torch.tensor([i - 1 for i in self.mask.indices_2], dtype=torch.long, device=xc.device)
        x2 = xc[i2].detach().clone().requires_grad_(True) if vec else xc[i2, :].detach().clone().requires_grad_(True)
        with torch.enable_grad():
            law = self.theta(x2)
            if isinstance(law, RationalQuadraticSpline):
                # spline law (coupling.jl:206-259 with b = RationalQuadraticSpline(w, h, d), knots shared by the batch):
                # x̄₁ = the elementwise spline pullback on the x₁ rows (bjx_rqs_vjp), every other row passes ȳ through.
                # When θ builds the knots from x₂ with torch operations (the neural-spline coupling), the knot cotangents of
                # the same pass (bjx_rqs_vjp_knots) go back through θ by torch.autograd and land on the x₂ rows — as the
                # affine branch below does with (s̄, t̄).
                i1 = self.mask.idx1_dev(xc.device).long()
                x1 = colmajor(xc[i1] if not vec else xc[i1])
                g1 = colmajor(gc[i1])
                knots = [t for t in (law.widths, law.heights, law.derivatives)]
                through_theta = any(isinstance(t, torch.Tensor) and t.requires_grad for t in knots)
                xb = gc.clone() if vec else colmajor(gc.clone())
                if not through_theta:
                    xb[i1] = vjp(inverse(law) if inv else law, x1, g1, ladj_bar)
                    return xb
                plain = RationalQuadraticSpline(*[t.detach() for t in knots])
                sub, kb = _vjp_params_rqs(inverse(plain) if inv else plain, x1, g1, ladj_bar)
                xb[i1] = sub
                outs = [t for t in knots if t.requires_grad]
                cots = [kb[n].reshape(t.shape).to(t.dtype) for n, t in zip(("widths", "heights", "derivatives"), knots) if t.requires_grad]
                g2, = torch.autograd.grad(outs, [x2], cots, allow_unused=True)
                if g2 is not None:
                    xb[i2] += g2
                return xb
            scale = shift = None
            for st in (law._stages() if isinstance(law, ComposedFunction) else [law]):
                if isinstance(st, Scale) and scale is None and shift is None:
                    scale = st.a
                elif isinstance(st, Shift) and shift is None:
                    shift = st.a
                else:
                    raise NotImplementedError(f"no device pullback for the coupling law {law!r} (affine laws only)")
            n1 = len(self.mask.indices_1)

            def full(p):
                if p is None:
                    return None
                t = p if isinstance(p, torch.Tensor) else torch.as_tensor(p, dtype=xc.dtype, device=xc.device)
                t = t.to(device=xc.device, dtype=xc.dtype)
                if t.dim() == 0:
                    t = t.expand(n1)
                if t.dim() == 1 and not vec:
                    t = t[:, None].expand(n1, batch)
                return t
            s_f, t_f = full(scale), full(shift)
        s_c = None if s_f is None else colmajor(s_f.detach().contiguous() if s_f.dim() == 1 else s_f.detach())
        t_c = None if t_f is None else colmajor(t_f.detach().contiguous() if t_f.dim() == 1 else t_f.detach())
        idx1 = torch.tensor([i - 1 for i in self.mask.indices_1], dtype=torch.int32, device=xc.device)
        lb = _ladj_bar(ladj_bar, batch, xc)
        ctx = context(xc.device)
        xb = _empty(dim, batch, xc, vec)
        sb = None if s_c is None else torch.empty((batch, n1), dtype=xc.dtype, device=xc.device).T
        tb = None if t_c is None else torch.empty((batch, n1), dtype=xc.dtype, device=xc.device).T
        rc = L.load().bjx_coupling_affine_vjp(ctx.h, _dt(xc), int(inv), _ptr(idx1), n1, _ptr(s_c), _ptr(t_c), _ptr(xc), _ptr(gc), _ptr(lb),
                                              _ptr(xb), _ptr(sb), _ptr(tb), dim, batch)
        L.check(ctx.h, rc, "bjx_coupling_affine_vjp")
        outs, cots = [], []
        for f_, b_ in ((s_f, sb), (t_f, tb)):
            if f_ is not None and f_.requires_grad:
                outs.append(f_)
                cots.append(b_.reshape(-1) if vec else b_)
        if outs:                                   # θ depends on x₂ through torch ops: add its pullback to the x₂ rows
            g2, = torch.autograd.grad(outs, [x2], cots, allow_unused=True)
            if g2 is not None:
                if vec:
                    xb[i2] += g2
                else:
                    xb[i2, :] += g2
        return xb


# ------------------------------------------------------------------ Stacked
def _elementwise_ops(b):
    """[(kind, p0, p1)] in application order if `b` is a (chain of) elementwise bijector(s), else None."""
    if b is identity:          # `identity` is its own bijector in the reference (stacked.jl:21-23 example)
        return []
    return _fused_ops(b)


class Stacked(Transform):
    """src/bijectors/stacked.jl:27-252.  `Stacked(bs)` applies bs[i] to row i; `Stacked(bs, ranges)` applies
    bs[i] to x[ranges[i]] where ranges[i] = (lo, hi) is Julia's lo:hi (1-based, inclusive).  The output is
    the concatenation of the pieces in the order of `bs` (ranges_out are cumulative, :50-57).

    Every segment whose bijector is a chain of at most 4 elementwise ops is evaluated by ONE fused launch
    (bjx_stacked) over all columns; a structured segment (Simplex, Ordered, ...) is sliced out, sent through
    its own entry point and written back."""

    def __init__(self, bs, ranges=None):
        self.bs = list(bs) if isinstance(bs, (list, tuple)) else [bs]
        if ranges is None:
            ranges = [(i + 1, i + 1) for i in range(len(self.bs))]                    # :46-48
        self.ranges_in = [(int(lo), int(hi)) for lo, hi in ranges]
        if len(self.ranges_in) != len(self.bs):
            raise ValueError("length(bs) == length(ranges) needs to be true")           # :14
        self.ranges_out, off = [], 0
        for b, (lo, hi) in zip(self.bs, self.ranges_in):                                # determine_output_ranges :50-57
            n_out = (hi - lo + 1) if b is identity else output_size(b, (hi - lo + 1,))[0]
            self.ranges_out.append((off + 1, off + n_out))
            off += n_out
        self.length_in = sum(hi - lo + 1 for lo, hi in self.ranges_in)
        self.length_out = off

    def _key(self):
        return (tuple(self.bs), tuple(self.ranges_in))

    def _inverse(self):                                                                  # :113-118
        inv = Stacked.__new__(Stacked)
        inv.bs = [b if b is identity else inverse(b) for b in self.bs]
        inv.ranges_in, inv.ranges_out = list(self.ranges_out), list(self.ranges_in)
        inv.length_in, inv.length_out = self.length_out, self.length_in
        return inv

    def _segments(self, segs_ops, fused, xc):
        """bjx_segment[] for the fusable segments (+ the parameter tensors that must stay alive).  Kept per (segments, ranges, dtype,
        device) like the op list of a chain (`_chain_key`): pointers and scalars only, nothing that could go stale."""
        skey = [len(self.bs)]
        for i in fused:
            (lo, hi), (olo, _) = self.ranges_in[i], self.ranges_out[i]
            k = _chain_key(segs_ops[i], xc, hi - lo + 1)
            if k is None:
                skey = None
                break
            skey.append((i, lo, hi, olo, k))
        if skey is not None:
            skey = ("seg",) + tuple(skey)
            hit = _CHAIN_OPS.get(skey)
            if hit is not None:
                return hit, []
        arr = (L.BjxSegment * max(len(self.bs), 1))()
        keep = []
        for si, i in enumerate(fused):
            (lo, hi), (olo, _) = self.ranges_in[i], self.ranges_out[i]
            sg = arr[si]
            sg.in_lo, sg.out_lo, sg.len, sg.n_ops = lo - 1, olo - 1, hi - lo + 1, len(segs_ops[i])
            for k, (kind, p0, p1) in enumerate(segs_ops[i]):
                o = sg.ops[k]
                o.kind, o.param_len, o.p0, o.p1, o.v0, o.v1 = kind, 0, 0.0, 0.0, None, None
                seq = any(_is_seq(p) for p in (p0, p1) if p is not None)
                for j, p in enumerate((p0, p1)):
                    if p is None:
                        continue
                    if seq:
                        t = _param(p, xc).reshape(-1) if _is_seq(p) else torch.full((sg.len,), float(p), dtype=xc.dtype, device=xc.device)
                        if t.numel() != sg.len:
                            raise ValueError(f"DimensionMismatch: parameter of length {t.numel()} for a segment of {sg.len} rows")
                        keep.append(t)
                        o.param_len = sg.len
                        setattr(o, f"v{j}", t.data_ptr())
                    else:
                        o.param_len = 1
                        setattr(o, f"p{j}", float(p))
        if skey is not None:
            if len(_CHAIN_OPS) >= 512:
                _CHAIN_OPS.clear()
            _CHAIN_OPS[skey] = arr
        return arr, keep

    def _vjp(self, x, out_bar, ladj_bar, moments=False):
        """moments=True: also (Σ_n x̄, Σ_n x̄·x) per row from the same pass (bjx_stacked_vjp_moments)."""
        xc, dim, batch, vec = _prep(x)
        gc, gdim, gbatch, _ = _prep(out_bar)
        if dim != self.length_in:
            raise ValueError(f"input length mismatch ({self.length_in} != {dim})")
        if (gdim, gbatch) != (self.length_out, batch) or gc.dtype != xc.dtype:
            raise ValueError("DimensionMismatch: out_bar must have the shape and dtype of the output")
        segs_ops = [_elementwise_ops(b) for b in self.bs]
        if any(o is None or len(o) > L.BJX_MAX_SEG_OPS for o in segs_ops) or self.length_out != dim:
            if moments:
                raise NotImplementedError("row moments of a Stacked pullback need every segment to be a chain of <= 4 elementwise bijectors")
            return self._vjp_by_segment(xc, gc, ladj_bar, batch, vec)
        arr, keep = self._segments(segs_ops, list(range(len(self.bs))), xc)
        lb = _ladj_bar(ladj_bar, batch, xc)
        ctx = context(xc.device)
        xb = _empty(dim, batch, xc, vec)
        if moments:
            mom = torch.empty(2 * dim + 1, dtype=torch.float64, device=xc.device)
            rc = L.load().bjx_stacked_vjp_moments(ctx.h, _dt(xc), arr, len(self.bs), _ptr(xc), _ptr(gc), _ptr(lb), _ptr(xb), _ptr(mom), dim, batch)
            del keep
            L.check(ctx.h, rc, "bjx_stacked_vjp_moments")
            return xb, mom[:dim], mom[dim:2 * dim]
        rc = L.load().bjx_stacked_vjp(ctx.h, _dt(xc), arr, len(self.bs), _ptr(xc), _ptr(gc), _ptr(lb), _ptr(xb), dim, batch)
        del keep
        L.check(ctx.h, rc, "bjx_stacked_vjp")
        return xb

    def _vjp_by_segment(self, xc, gc, ladj_bar, batch, vec):
        """Pullback of a Stacked with structured segments (Simplex / Ordered / LKJ blocks, long chains): the segments act on
        disjoint row ranges and the log-det is their sum (stacked.jl:168-252), so x̄[ranges_in[i]] = vjp(bs[i], x[ranges_in[i]],
        ȳ[ranges_out[i]], ℓ̄) with the same ℓ̄ for every segment — what HMC differentiates for a mixed-constraint model.  The
        row ranges are gathered into dense blocks (one copy in, one out per segment)."""
        xb = torch.empty((xc.shape[0],), dtype=xc.dtype, device=xc.device) if vec else _empty(xc.shape[0], batch, xc, False)
        covered = torch.zeros(xc.shape[0], dtype=torch.bool)
        for b, (ilo, ihi), (olo, ohi) in zip(self.bs, self.ranges_in, self.ranges_out):
            xi = xc[ilo - 1:ihi] if vec else colmajor(xc[ilo - 1:ihi, :])
            gi = gc[olo - 1:ohi] if vec else colmajor(gc[olo - 1:ohi, :])
            if b is identity:
                xb[ilo - 1:ihi] = gi
            else:
                xb[ilo - 1:ihi] = vjp(b, xi.contiguous() if vec else xi, gi.contiguous() if vec else gi, ladj_bar)
            covered[ilo - 1:ihi] = True
        if not bool(covered.all()):
            raise ValueError("Stacked: the input ranges must cover every row for a pullback")
        return xb

    def _wlj_in_place(self, xc, dim, batch, vec, y, out, segs_ops, fused, rest, per_sample, want_ladj, ctx):
        """The copy-free path for Stacked with Simplex / Ordered segments; None when a segment has no `_ld` entry point."""
        def struct_of(b):
            inv = isinstance(b, Inverse)
            base = b.orig if inv else b
            if isinstance(base, SimplexBijector):
                return "bjx_simplex_ld", inv
            if isinstance(base, OrderedBijector):
                return "bjx_ordered_ld", inv
            return None
        if vec or batch == 0 or any(struct_of(self.bs[i]) is None for i in rest):
            return None
        if any((self.ranges_in[i][1] - self.ranges_in[i][0] + 1) > 250 for i in rest):     # LDS tile of the strided kernel
            return None
        dout = self.length_out
        ps = torch.zeros(batch, dtype=xc.dtype, device=xc.device) if (want_ladj and out.ps is not None) else None
        sm = torch.zeros(1, dtype=torch.float64, device=xc.device) if (want_ladj and out.sum is not None) else None
        lib = L.load()
        es = xc.element_size()

        def fill_segment(sg, i, x_off, y_off, keep):
            lo, hi = self.ranges_in[i]
            ln, ops = hi - lo + 1, segs_ops[i]
            sg.in_lo, sg.out_lo, sg.len, sg.n_ops = lo - 1 - x_off, self.ranges_out[i][0] - 1 - y_off, ln, len(ops)
            for k, (kind, p0, p1) in enumerate(ops):
                o = sg.ops[k]
                o.kind, o.param_len, o.p0, o.p1, o.v0, o.v1 = kind, 0, 0.0, 0.0, None, None
                seq = any(_is_seq(p) for p in (p0, p1) if p is not None)
                for j, p in enumerate((p0, p1)):
                    if p is None:
                        continue
                    if seq:
                        t = _param(p, xc).reshape(-1) if _is_seq(p) else torch.full((ln,), float(p), dtype=xc.dtype, device=xc.device)
                        if t.numel() != ln:
                            raise ValueError(f"DimensionMismatch: parameter of length {t.numel()} for a segment of {ln} rows")
                        keep.append(t)
                        o.param_len = ln
                        setattr(o, f"v{j}", t.data_ptr())
                    else:
                        o.param_len = 1
                        setattr(o, f"p{j}", float(p))

        # ONE launch when a whole column fits the LDS tile of bjx_stacked_mixed (a lane walks its column: elementwise rows and
        # structured blocks in the same pass); taller columns: the window launches below
        order_all = sorted(range(len(self.bs)), key=lambda i: self.ranges_out[i][0])
        in_ascending = all(self.ranges_in[a][0] < self.ranges_in[b][0] for a, b in zip(order_all, order_all[1:]))
        if in_ascending and len(rest) <= 64:
            el = [i for i in order_all if i in fused]
            st = [i for i in order_all if i in rest]
            arr = (L.BjxSegment * max(len(el), 1))()
            keep = []
            for si, i in enumerate(el):
                fill_segment(arr[si], i, 0, 0, keep)
            blk = (L.BjxBlock * max(len(st), 1))()
            for bi, i in enumerate(st):
                fn, inv = struct_of(self.bs[i])
                (lo, hi), (olo, ohi) = self.ranges_in[i], self.ranges_out[i]
                b_ = blk[bi]
                b_.kind = (L.BLOCK_SIMPLEX_INV if inv else L.BLOCK_SIMPLEX) if fn == "bjx_simplex_ld" else (L.BLOCK_ORDERED_INV if inv else L.BLOCK_ORDERED)
                b_.in_lo, b_.out_lo, b_.len_in, b_.len_out = lo - 1, olo - 1, hi - lo + 1, ohi - olo + 1
            rc = lib.bjx_stacked_mixed(ctx.h, _dt(xc), arr, len(el), blk, len(st), _ptr(xc), dim, _ptr(y), dout, _ptr(ps), _ptr(sm), batch, 0)
            del keep
            if rc != L.ERR_UNSUPPORTED:
                L.check(ctx.h, rc, "bjx_stacked_mixed")
                return self._ladj_result(y, ps, sm, out, per_sample, want_ladj, xc)
        first = True
        if fused:
            # maximal runs of elementwise segments that are contiguous on BOTH sides keep their row offsets inside a window of
            # x and y: one bjx_stacked_ld launch per run, streaming packs (no gather), log-dets accumulated run after run
            order = sorted(fused, key=lambda i: self.ranges_out[i][0])
            runs, cur = [], []
            for i in order:
                if cur and self.ranges_out[i][0] == self.ranges_out[cur[-1]][1] + 1 and self.ranges_in[i][0] == self.ranges_in[cur[-1]][1] + 1:
                    cur.append(i)
                else:
                    if cur:
                        runs.append(cur)
                    cur = [i]
            if cur:
                runs.append(cur)
            for run in runs:
                x_off, y_off = self.ranges_in[run[0]][0] - 1, self.ranges_out[run[0]][0] - 1
                rows_run = sum(self.ranges_in[i][1] - self.ranges_in[i][0] + 1 for i in run)
                arr = (L.BjxSegment * len(run))()
                keep = []
                for si, i in enumerate(run):
                    fill_segment(arr[si], i, x_off, y_off, keep)
                rc = lib.bjx_stacked_ld(ctx.h, _dt(xc), arr, len(run), C.c_void_p(xc.data_ptr() + x_off * es), dim, C.c_void_p(y.data_ptr() + y_off * es), dout,
                                        _ptr(ps), _ptr(sm), rows_run, batch, 0 if first else L.BJX_ACCUMULATE)
                del keep
                if rc == L.ERR_UNSUPPORTED:
                    return None
                L.check(ctx.h, rc, "bjx_stacked_ld")
                first = False
        for i in rest:
            fn, inv = struct_of(self.bs[i])
            (lo, hi), (olo, ohi) = self.ranges_in[i], self.ranges_out[i]
            n_in, n_out = hi - lo + 1, ohi - olo + 1
            size = (n_out if inv else n_in) if fn == "bjx_simplex_ld" else n_in          # K of the simplex side / rows of Ordered
            rc = getattr(lib, fn)(ctx.h, _dt(xc), int(inv), C.c_void_p(xc.data_ptr() + (lo - 1) * es), dim,
                                  C.c_void_p(y.data_ptr() + (olo - 1) * es), dout, _ptr(ps), _ptr(sm), size, batch, L.BJX_ACCUMULATE)
            if rc == L.ERR_UNSUPPORTED:
                return None
            L.check(ctx.h, rc, fn)
        return self._ladj_result(y, ps, sm, out, per_sample, want_ladj, xc)

    @staticmethod
    def _ladj_result(y, ps, sm, out, per_sample, want_ladj, xc):
        if not want_ladj:
            return y, None
        if out.both:
            return y, (ps, sm)
        if out.sum64:
            return y, sm
        if per_sample:
            return y, ps
        return y, sm[0].to(xc.dtype)

    def _wlj(self, x, per_sample, want_ladj=True):
        xc, dim, batch, vec = _prep(x)
        if dim != self.length_in:
            raise ValueError(f"input length mismatch ({self.length_in} != {dim})")       # :157
        segs_ops = [_elementwise_ops(b) for b in self.bs]
        fused = [i for i, o in enumerate(segs_ops) if o is not None and len(o) <= L.BJX_MAX_SEG_OPS]
        rest = [i for i in range(len(self.bs)) if i not in fused]
        ctx = context(xc.device)
        y = _empty(self.length_out, batch, xc, vec)
        out = _Out(xc, batch, per_sample, want_ladj)
        if not rest and self.length_out == dim:    # (all-elementwise stacks: the row-owner kernel, 71 % vs 49 % through the column walker)
            arr, keep = self._segments(segs_ops, fused, xc)
            rc = L.load().bjx_stacked(ctx.h, _dt(xc), arr, len(fused), _ptr(xc), _ptr(y), _ptr(out.ps), _ptr(out.sum), dim, batch, 0)
            del keep
            if rc != L.ERR_UNSUPPORTED:   # a chain with > 2 nonlinear stages is evaluated per segment below
                L.check(ctx.h, rc, "bjx_stacked")
                return (y, out.result(vec_scalar=vec and bool(per_sample) and per_sample is True)) if want_ladj else (y, None)
        # structured segments (Simplex / Ordered blocks): no slicing copies — the elementwise segments in one
        # bjx_stacked_ld launch (identity placeholders on the structured rows), then the structured entry points with a
        # leading dimension overwrite their rows in place and ACCUMULATE their log-dets
        r_ = self._wlj_in_place(xc, dim, batch, vec, y, out, segs_ops, fused, rest, per_sample, want_ladj, ctx)
        if r_ is not None:
            return r_
        # general case: per-segment launches on row slices (copies); log-dets are summed like :236-244
        total = None
        x2 = xc if not vec else xc[:, None]
        y2 = y if not vec else y[:, None]
        for b, (lo, hi), (olo, ohi) in zip(self.bs, self.ranges_in, self.ranges_out):
            piece = colmajor(x2[lo - 1:hi, :])
            if b is identity:
                b = Shift(0.0)
            yp, lp = b._wlj(piece, per_sample="both" if want_ladj else False, want_ladj=want_ladj)
            y2[olo - 1:ohi, :] = yp
            if want_ladj:
                total = lp if total is None else (total[0] + lp[0], total[1] + lp[1])
        if not want_ladj:
            return y, None
        ps, sm = total
        if out.both:
            return y, (ps, sm)
        if out.sum64:
            return y, sm
        if per_sample:
            return y, (ps[0] if vec else ps)
        return y, sm[0].to(xc.dtype)


class NamedStacked(Transform):
    """src/bijectors/named_stacked.jl:1-60 — a NamedTuple of bijectors for `ProductNamedTupleDistribution` samples.
    `transforms` and `ranges` are dicts with the same keys (insertion order = field order); `ranges[name]` is the
    1-based index (int) or (lo, hi) range of that field's OUTPUT in the stacked vector.  The forward direction takes a
    dict of values (python numbers, (len,) or (len, batch) tensors) and returns ONE vector / matrix: the fields are
    concatenated and sent through `Stacked`, so elementwise fields cost one fused launch (bjx_stacked) whatever their
    number; the inverse takes the vector and returns a dict (:118-150)."""

    def __init__(self, transforms: dict, ranges: dict, _inv: bool = False):
        if list(transforms.keys()) != list(ranges.keys()):
            raise ValueError("transforms and ranges need the same field names")           # NamedTuple{names} on both, :53-58
        self.names = list(transforms.keys())
        self.transforms = dict(transforms)
        self.ranges = {n: ((int(r), int(r)) if isinstance(r, int) else (int(r[0]), int(r[1]))) for n, r in ranges.items()}
        self._int_fields = {n for n, r in ranges.items() if isinstance(r, int)} if not isinstance(ranges, NamedStacked) else set()
        self._ranges_arg = dict(ranges)
        self._inv = _inv

    def _key(self):
        return (tuple(self.names), tuple(self.transforms[n] for n in self.names), tuple(self.ranges[n] for n in self.names), self._inv)

    def _inverse(self):
        return NamedStacked(self.transforms, self._ranges_arg, not self._inv)

    def _stacked(self):
        """The equivalent `Stacked` over the concatenated fields (input ranges are cumulative input lengths)."""
        bs, rin, off = [], [], 0
        for n in self.names:
            b = self.transforms[n]
            lo, hi = self.ranges[n]
            n_out = hi - lo + 1
            n_in = n_out if b is identity else output_size(inverse(b), (n_out,))[0]
            bs.append(b)
            rin.append((off + 1, off + n_in))
            off += n_in
        st = Stacked(bs, rin)
        if [tuple(r) for r in st.ranges_out] != [self.ranges[n] for n in self.names]:
            raise ValueError(f"ranges {self.ranges} do not match the output sizes of the transforms ({st.ranges_out})")
        return st

    def _wlj(self, x, per_sample, want_ladj=True):
        st = self._stacked()
        if self._inv:                                                                    # :118-150: vector -> NamedTuple
            xs, l = st._inverse()._wlj(x, per_sample, want_ladj)
            out = {}
            for n, (lo, hi) in zip(self.names, st.ranges_in):
                piece = xs[lo - 1:hi]
                out[n] = piece[0] if n in self._int_fields else piece               # an Int range is a scalar field (:19-21)
            return out, l
        return st._wlj(self._cat(x), per_sample, want_ladj)

    def _cat(self, x):
        """A dict of fields -> the concatenated (rows[, batch]) device tensor `Stacked` works on."""
        if not isinstance(x, dict) or list(x.keys()) != self.names:
            raise ValueError(f"expected a dict with the fields {self.names}")             # x::NamedTuple{names}, :66
        like = next((v for v in x.values() if isinstance(v, torch.Tensor) and v.is_cuda), None)
        if like is None:
            raise RuntimeError("bijectors_amd operates on ROCm device tensors only (no CPU fallback): at least one field must be a device tensor")
        # the batch size comes from the 2-D fields (all of them must agree), never from an unbatched 1-D field
        widths = {int(v.shape[1]) for v in x.values() if isinstance(v, torch.Tensor) and v.dim() == 2}
        if len(widths) > 1:
            raise ValueError(f"DimensionMismatch: fields with different batch sizes {sorted(widths)}")
        batched = bool(widths)
        nb = widths.pop() if widths else 1
        pieces = []
        for n in self.names:
            v = x[n]
            t = v if isinstance(v, torch.Tensor) else torch.as_tensor(v, dtype=like.dtype)
            t = t.to(device=like.device, dtype=like.dtype)
            if t.dim() == 0:
                t = t.reshape(1)
            if batched and t.dim() == 1:
                if n in self._int_fields and t.shape[0] != 1:                            # a scalar field holds one value per column
                    if t.shape[0] != nb:
                        raise ValueError(f"DimensionMismatch: scalar field {n!r} has {t.shape[0]} values for a batch of {nb}")
                    t = t[None, :]
                else:                                                                    # an unbatched vector field: the same for every column
                    t = t[:, None].expand(t.shape[0], nb)
            pieces.append(t)
        cat = torch.cat(pieces, dim=0)
        return colmajor(cat) if batched else cat.contiguous()

    def _split(self, xs, ranges):
        out = {}
        for n, (lo, hi) in zip(self.names, ranges):
            piece = xs[lo - 1:hi]
            out[n] = piece[0] if n in self._int_fields else piece
        return out

    def _vjp(self, x, out_bar, ladj_bar):
        """Pullback in the field layout of the forward value: forward (dict -> vector) takes the vector cotangent and returns a
        dict; the inverse (vector -> dict, what a log-density evaluation of a ProductNamedTupleDistribution differentiates) takes a
        dict of field cotangents and returns the vector's."""
        st = self._stacked()
        if self._inv:
            return vjp(st._inverse(), x, self._cat(out_bar), ladj_bar)
        return self._split(vjp(st, self._cat(x), out_bar, ladj_bar), st.ranges_in)


# ------------------------------------------------------------------ reverse-mode pullbacks
def _ladj_bar(ladj_bar, batch, like):
    if ladj_bar is None:
        return None
    lb = ladj_bar if isinstance(ladj_bar, torch.Tensor) else torch.full((batch,), float(ladj_bar), dtype=like.dtype, device=like.device)
    lb = lb.to(device=like.device, dtype=like.dtype).reshape(-1).contiguous()
    if lb.numel() == 1 and batch != 1:
        lb = lb.expand(batch).contiguous()
    if lb.numel() != batch:
        raise ValueError("DimensionMismatch: ladj_bar needs one entry per column")
    return lb


def vjp(b, x, out_bar, ladj_bar=None):
    """Pullback of `with_logabsdet_jacobian(b, x)`: returns x_bar = J(x)^T out_bar + ladj_bar * grad_x logabsdetjac.

    `out_bar` has the shape of b(x); `ladj_bar` is the cotangent of the PER-COLUMN log-det (a (batch,) tensor,
    a python number broadcast over the batch, or None = 0).  Input pullbacks only (parameters: `vjp_params` for the
    PlanarLayer stack).  Device kernels: elementwise chains / `Stacked` (bjx_stacked_vjp); the rules the reference
    ships (ext/BijectorsChainRulesCoreExt.jl) — OrderedBijector and its inverse (:65-197), the LKJ-Cholesky link
    (:199-311) and its inverse (:311-320); SimplexBijector, PlanarLayer, RadialLayer, RationalQuadraticSpline and the
    affine Coupling in both directions; Permute (the inverse gather); InvertibleBatchNorm in eval mode and
    `columnwise(f)` through the kernels above."""
    if isinstance(b, Stacked):
        return b._vjp(x, out_bar, ladj_bar)
    if isinstance(b, NamedStacked):
        return b._vjp(x, out_bar, ladj_bar)
    if isinstance(b, (ComposedFunction, _ChainOp, Elementwise, Inverse)):
        r = _fast_chain_vjp(b, lambda: _elementwise_ops(b), x, out_bar, ladj_bar)       # cached pullback plan (include/bjx.h "plans")
        if r is not None:
            return r
    ops_ = _elementwise_ops(b)
    if ops_ is not None and len(ops_) <= L.BJX_MAX_SEG_OPS:   # a chain of elementwise bijectors = one segment over all rows
        dim = x.shape[0]
        return Stacked([b], [(1, dim)])._vjp(x, out_bar, ladj_bar)
    if isinstance(b, ComposedFunction):                  # longer chains, compositions of layers: the chain rule piece by piece
        return _vjp_composed(b, x, out_bar, ladj_bar)
    inv = isinstance(b, Inverse)
    base = b.orig if inv else b
    if inv and isinstance(base, VecCholeskyBijector):
        # inverse(VecCholeskyBijector): y (n[, batch]) -> W (K, K[, batch]); corr.jl:402-451
        yc, n, batch, vec = _prep(x)
        K = _triu1_dim_from_length(n)
        if K * (K - 1) // 2 != n:
            raise ValueError(f"DimensionMismatch: {n} is not a triangular number K(K-1)/2")
        Wb = out_bar
        if tuple(Wb.shape) != ((K, K) if vec else (K, K, batch)) or Wb.dtype != yc.dtype:
            raise ValueError("DimensionMismatch: out_bar must be (K, K[, batch]) with the dtype of y")
        Wc = Wb.T.contiguous().T if vec else Wb.permute(2, 1, 0).contiguous().permute(2, 1, 0)   # column-major per sample
        lb = _ladj_bar(ladj_bar, batch, yc)
        ctx = context(yc.device)
        yb = _empty(n, batch, yc, vec)
        rc = L.load().bjx_vec_cholesky_inv_vjp(ctx.h, _dt(yc), ord(base.mode), _ptr(yc), _ptr(Wc), _ptr(lb), _ptr(yb), K, batch)
        L.check(ctx.h, rc, "bjx_vec_cholesky_inv_vjp")
        return yb
    if isinstance(b, VecCholeskyBijector):
        # forward link W (K, K[, batch]) -> y: the rule of ext/BijectorsChainRulesCoreExt.jl:199-311 (no log-det cotangent)
        if ladj_bar is not None:
            raise NotImplementedError("the reference's rule for the forward LKJ link has no log-det cotangent")
        W = x
        vec = W.dim() == 2
        K = W.shape[0]
        if W.shape[1] != K:
            raise ValueError("DimensionMismatch: W must be K x K[ x batch]")
        batch = 1 if vec else W.shape[2]
        n = K * (K - 1) // 2
        Wc = W.T.contiguous().T if vec else W.permute(2, 1, 0).contiguous().permute(2, 1, 0)
        gb = out_bar.reshape(n, 1) if vec else out_bar
        gc = colmajor(gb)
        if tuple(gc.shape) != (n, batch) or gc.dtype != Wc.dtype:
            raise ValueError("DimensionMismatch: out_bar must be (K(K-1)/2[, batch]) with the dtype of W")
        _check_dev(Wc)
        ctx = context(Wc.device)
        Wb = torch.empty((batch, K, K), dtype=Wc.dtype, device=Wc.device).permute(2, 1, 0)
        rc = L.load().bjx_vec_cholesky_fwd_vjp(ctx.h, _dt(Wc), ord(b.mode), _ptr(Wc), _ptr(gc), _ptr(Wb), K, batch)
        L.check(ctx.h, rc, "bjx_vec_cholesky_fwd_vjp")
        return Wb[:, :, 0] if vec else Wb
    inv = isinstance(b, Inverse)
    base = b.orig if inv else b
    if isinstance(base, _MatrixBijector):
        return _vjp_matrix(base, inv, x, out_bar, ladj_bar)
    if isinstance(base, Scale) and base.matrix:
        # y = a x: x̄ = aᵀ ȳ; x = a \ y: ȳ = a⁻ᵀ x̄ (the input side of ext/BijectorsReverseDiffExt.jl:72-115; the log-det does not
        # depend on the input) — the same entry with the transposed matrix
        at = colmajor(_param(base.a, out_bar)).T
        return transform(inverse(Scale(at)) if inv else Scale(at), out_bar)
    if isinstance(base, SimplexBijector):
        xc, rows, batch, vec = _prep(x)
        gc, grows, gbatch, _ = _prep(out_bar)
        K = rows + 1 if inv else rows
        if (grows, gbatch) != ((K if inv else K - 1), batch) or gc.dtype != xc.dtype:
            raise ValueError("DimensionMismatch: out_bar must have the shape and dtype of the output")
        lb = _ladj_bar(ladj_bar, batch, xc)
        ctx = context(xc.device)
        xb = _empty(rows, batch, xc, vec)
        rc = L.load().bjx_simplex_vjp(ctx.h, _dt(xc), int(inv), _ptr(xc), _ptr(gc), _ptr(lb), _ptr(xb), K, batch)
        L.check(ctx.h, rc, "bjx_simplex_vjp")
        return xb
    if isinstance(base, PlanarLayer):
        # fused PlanarLayer stack, either direction (closed-form derivatives of planar_layer.jl:65-127; find_alpha's
        # implicit-function rule ext/BijectorsChainRulesCoreExt.jl:42-46 for the inverse)
        xc, dim, batch, vec = _prep(x)
        gc, gdim, gbatch, _ = _prep(out_bar)
        if (gdim, gbatch) != (dim, batch) or gc.dtype != xc.dtype:
            raise ValueError("DimensionMismatch: out_bar must have the shape and dtype of the output")
        w, u, bb = base._tables(xc, dim)
        lb = _ladj_bar(ladj_bar, batch, xc)
        ctx = context(xc.device)
        xb = _empty(dim, batch, xc, vec)
        rc = L.load().bjx_planar_vjp(ctx.h, _dt(xc), int(inv), _ptr(w), _ptr(u), _ptr(bb), base.n_layers, _ptr(xc), _ptr(gc), _ptr(lb), _ptr(xb), dim, batch)
        L.check(ctx.h, rc, "bjx_planar_vjp")
        return xb
    if isinstance(base, Permute):
        # y = A x with A a permutation matrix: x̄ = Aᵀȳ = the inverse permutation of ȳ (bit-exact gather; log-det 0)
        return transform(base if inv else inverse(base), out_bar)
    if isinstance(base, Columnwise):
        return vjp(inverse(base.x) if inv else base.x, x, out_bar, ladj_bar)
    if isinstance(base, Coupling):
        return base._vjp(x, out_bar, ladj_bar, inv)
    if isinstance(base, RationalQuadraticSpline):
        xc, dim, batch, vec = _prep(x)
        gc, gdim, gbatch, _ = _prep(out_bar)
        if (gdim, gbatch) != (dim, batch) or gc.dtype != xc.dtype:
            raise ValueError("DimensionMismatch: out_bar must have the shape and dtype of the output")
        if dim != base.widths.shape[0]:
            raise ValueError(f"DimensionMismatch: spline with {base.widths.shape[0]} rows applied to {dim} rows")
        w, h, d = (colmajor(_param(t, xc)) for t in (base.widths, base.heights, base.derivatives))
        lb = _ladj_bar(ladj_bar, batch, xc)
        ctx = context(xc.device)
        xb = _empty(dim, batch, xc, vec)
        rc = L.load().bjx_rqs_vjp(ctx.h, _dt(xc), int(inv), _ptr(w), _ptr(h), _ptr(d), int(base.widths.shape[1]), _ptr(xc), _ptr(gc), _ptr(lb), _ptr(xb), dim, batch)
        L.check(ctx.h, rc, "bjx_rqs_vjp")
        return xb
    if isinstance(base, RadialLayer):
        xc, dim, batch, vec = _prep(x)
        gc, gdim, gbatch, _ = _prep(out_bar)
        if (gdim, gbatch) != (dim, batch) or gc.dtype != xc.dtype:
            raise ValueError("DimensionMismatch: out_bar must have the shape and dtype of the output")
        z0 = _param(base.z_0, xc)
        if z0.numel() != dim:
            raise ValueError(f"DimensionMismatch: RadialLayer of dimension {z0.numel()} applied to {dim} rows")
        a, be = _param(base.alpha_, xc), _param(base.beta, xc)
        lb = _ladj_bar(ladj_bar, batch, xc)
        ctx = context(xc.device)
        xb = _empty(dim, batch, xc, vec)
        rc = L.load().bjx_radial_vjp(ctx.h, _dt(xc), int(inv), _ptr(a), _ptr(be), _ptr(z0), _ptr(xc), _ptr(gc), _ptr(lb), _ptr(xb), dim, batch)
        L.check(ctx.h, rc, "bjx_radial_vjp")
        return xb
    if isinstance(base, InvertibleBatchNorm) and not istraining():
        # eval mode is a per-row affine map (normalise.jl:62,83): its input pullback is the one of Shift ∘ Scale ∘ Shift
        like = x
        s_ = torch.exp(_param(base.logs, like)) / torch.sqrt(_param(base.v, like) + base.eps)
        aff = Shift(_param(base.b, like)) @ Scale(s_) @ Shift(-_param(base.m, like))
        return vjp(inverse(aff) if inv else aff, x, out_bar, ladj_bar)
    if isinstance(base, InvertibleBatchNorm) and not inv:
        return _vjp_params_batchnorm_training(base, x, out_bar, ladj_bar)[0]      # training mode: the batch statistics depend on x
    if not isinstance(base, OrderedBijector):
        raise NotImplementedError(f"no device pullback for {b!r} yet (SURVEY.md §8f f-1)")
    xc, dim, batch, vec = _prep(x)
    gc, gdim, gbatch, _ = _prep(out_bar)
    if (gdim, gbatch) != (dim, batch) or gc.dtype != xc.dtype:
        raise ValueError("DimensionMismatch: out_bar must have the shape and dtype of the output")
    lb = _ladj_bar(ladj_bar, batch, xc)
    ctx = context(xc.device)
    xb = _empty(dim, batch, xc, vec)
    rc = L.load().bjx_ordered_vjp(ctx.h, _dt(xc), int(inv), _ptr(xc), _ptr(gc), _ptr(lb), _ptr(xb), dim, batch)
    L.check(ctx.h, rc, "bjx_ordered_vjp")
    return xb


def _vjp_matrix(base, inv, x, out_bar, ladj_bar):
    """Pullback of VecCorrBijector / CorrBijector / PDBijector / PDVecBijector and their inverses (bjx_*_vjp): the rules the
    reference ships, chained per sample — pd_from_upper / pd_from_lower (ext/BijectorsChainRulesCoreExt.jl:324-331,
    ext/BijectorsReverseDiffExt.jl:143-193), _inv_link_chol_lkj (corr.jl:402-451), replace_diag — and, for the forward direction, the
    cotangent of the link through the reverse of cholesky(Hermitian(X)), on the triangle the reference reads.  `out_bar` has the
    shape of the output: a (K, K[, batch]) matrix for the inverse direction (any matrix, not assumed symmetric)."""
    _check_dev(x)
    fn = base._FN + "_vjp"
    if inv:
        if base._VEC:
            xc, n, batch, single = _prep(x)
            K = base._K(n)
            if base._n(K) != n:
                raise ValueError(f"DimensionMismatch: {n} is not a valid packed length for {type(base).__name__}")
        else:
            if x.dim() not in (2, 3) or x.shape[0] != x.shape[1]:
                raise ValueError(f"DimensionMismatch: inverse({type(base).__name__}) expects a square (K, K[, batch]) matrix")
            K, single = x.shape[0], x.dim() == 2
            batch = 1 if single else x.shape[2]
            xc = _dense3(x)
        want = (K, K) if single else (K, K, batch)
        if tuple(out_bar.shape) != want or out_bar.dtype != x.dtype:
            raise ValueError(f"DimensionMismatch: out_bar must be {want} with the dtype of the input")
        gc = _dense3(out_bar)
    else:
        if x.dim() not in (2, 3) or x.shape[0] != x.shape[1]:
            raise ValueError(f"DimensionMismatch: {type(base).__name__} expects a square (K, K[, batch]) matrix")
        K, single = x.shape[0], x.dim() == 2
        batch = 1 if single else x.shape[2]
        xc = _dense3(x)
        if base._VEC:
            gc, gn, gbatch, _ = _prep(out_bar)
            if (gn, gbatch) != (base._n(K), batch) or gc.dtype != x.dtype:
                raise ValueError("DimensionMismatch: out_bar must have the shape and dtype of the output")
        else:
            if tuple(out_bar.shape) != tuple(x.shape) or out_bar.dtype != x.dtype:
                raise ValueError("DimensionMismatch: out_bar must have the shape and dtype of the output")
            gc = _dense3(out_bar)
    lb = _ladj_bar(ladj_bar, batch, xc)
    ctx = context(x.device)
    if inv and base._VEC:
        xb = _empty(xc.shape[0], batch, xc, single)
    elif single:
        xb = torch.empty((K, K), dtype=x.dtype, device=x.device).T
    else:
        xb = torch.empty((batch, K, K), dtype=x.dtype, device=x.device).permute(2, 1, 0)
    rc = getattr(L.load(), fn)(ctx.h, _dt(x), int(inv), _ptr(xc), _ptr(gc), _ptr(lb), _ptr(xb), K, batch)
    L.check(ctx.h, rc, fn)
    return xb


def _vjp_params_scale_matrix(b, x, out_bar, ladj_bar=None):
    """Scale with a MATRIX parameter (scale.jl:14,17,35-36; the rules of ext/BijectorsReverseDiffExt.jl:72-115): for y = a x with the
    per-column log-det logabsdet(a):  x̄ = aᵀȳ,  ā = ȳ xᵀ + (Σ_n ℓ̄_n) a⁻ᵀ; for the inverse x = a⁻¹y (log-det -logabsdet(a)):
    ȳ = a⁻ᵀx̄,  ā = -ȳ xᵀ - (Σ ℓ̄) a⁻ᵀ.  The input side is the library's own entry with the transposed matrix; the batch reduction
    ȳ xᵀ (a sum of `batch` outer products) and the log-det term are ONE library entry since round 6 (bjx_scale_matrix_vjp_params: matrix cores,
    deterministic fold, a⁻ᵀ from the library's own factorisation) — rounds 4-5 used the host runtime's GEMM and `torch.linalg.inv` here.
    -> (in_bar, {"a": ā})."""
    inv = isinstance(b, Inverse)
    base = b.orig if inv else b
    in_bar = vjp(b, x, out_bar, ladj_bar)
    a = colmajor(_param(base.a, x))
    dim = x.shape[0]
    batch = 1 if x.dim() == 1 else x.shape[1]
    lb = None if ladj_bar is None else _ladj_bar(ladj_bar, batch, a)
    G2 = colmajor((in_bar if inv else out_bar).reshape(dim, -1))
    X2 = colmajor((transform(b, x) if inv else x).reshape(dim, -1))
    a_bar = torch.empty((dim, dim), dtype=a.dtype, device=a.device).T             # column-major like `a`
    ctx = context(a.device)
    rc = L.load().bjx_scale_matrix_vjp_params(ctx.h, _dt(a), _ptr(a), _ptr(G2), _ptr(X2), _ptr(lb), -1.0 if inv else 1.0, _ptr(a_bar), dim, batch)
    L.check(ctx.h, rc, "bjx_scale_matrix_vjp_params")
    return in_bar, {"a": a_bar}


def _pieces(b):
    """A composition cut into pieces that have a device pullback: runs of elementwise stages of at most BJX_MAX_SEG_OPS ops
    (one fused launch each), every other stage alone.  Application order."""
    pieces, run, nops = [], [], 0
    for st in b._plan()[0]:
        o = _stage_ops(st)
        if o is not None and len(o) <= L.BJX_MAX_SEG_OPS:
            if nops + len(o) > L.BJX_MAX_SEG_OPS:
                pieces.append(_chain_of(run))
                run, nops = [], 0
            run.append(st)
            nops += len(o)
            continue
        if run:
            pieces.append(_chain_of(run))
            run, nops = [], 0
        pieces.append(st)
    if run:
        pieces.append(_chain_of(run))
    return pieces


def _vjp_composed(b, x, out_bar, ladj_bar=None):
    """Pullback of a composition f_n ∘ … ∘ f_1 (what an AD package does with the per-stage rules): forward through the pieces to
    get every piece's input, then x̄ = Σ-free chain rule backwards — the log-det is the SUM of the pieces' log-dets, so every
    piece receives the same ℓ̄.  Flows composed of different layers (planar ∘ radial ∘ …) and elementwise chains longer than
    one fused segment come here."""
    pieces = _pieces(b)
    inputs = [x]
    for pc in pieces[:-1]:
        inputs.append(transform(pc, inputs[-1]))
    g = out_bar
    for pc, xin in zip(reversed(pieces), reversed(inputs)):
        g = vjp(pc, xin, g, ladj_bar)
    return g


def _has_own_params(st):
    base = st.orig if isinstance(st, Inverse) else st
    return isinstance(base, (PlanarLayer, RadialLayer, RationalQuadraticSpline, InvertibleBatchNorm))


def _vjp_params_composed(b, x, out_bar, ladj_bar=None):
    """Input AND parameter pullback of a composition of layers (planar ∘ radial ∘ spline ∘ affine …): the chain rule of
    _vjp_composed with `vjp_params` at every stage that owns parameters (flow layers, splines, BatchNorm, Scale / Shift) and
    `vjp` at the others, on the PLANNED stages — a run of PlanarLayers is one bjx_planar_vjp_params launch, and its (w̄, ū, b̄)
    tables are handed back layer by layer.  Returns (x_bar, {"stages": [None | that stage's dictionary, ...]}) aligned with
    `b._stages()` (application order)."""
    stages, spans = b._plan()
    inputs = [x]
    _BN_RECOMPUTE[0] = True
    try:
        for st in stages[:-1]:
            inputs.append(transform(st, inputs[-1]))
    finally:
        _BN_RECOMPUTE[0] = False
    g = out_bar
    grads = [None] * spans[-1][1]
    for i in range(len(stages) - 1, -1, -1):
        st = stages[i]
        lo, hi = spans[i]
        if _has_own_params(st) or (isinstance(st, (Scale, Shift)) and not getattr(st, "matrix", False)):
            g, gr = vjp_params(st, inputs[i], g, ladj_bar)
            if hi - lo == 1:
                grads[lo] = gr
            else:     # a merged run: column k of the tables is layer k of the forward run = stage lo+k (reversed for an inverse run)
                order = range(lo, hi) if _planar_kind(st) == 1 else range(hi - 1, lo - 1, -1)
                for k, j in enumerate(order):
                    grads[j] = {"w": gr["w"][:, k], "u": gr["u"][:, k], "b": gr["b"][k:k + 1]}
        else:
            g = vjp(st, inputs[i], g, ladj_bar)
    return g, {"stages": grads}


def row_moments(a, b=None):
    """(Σ_n a[:, n], Σ_n a[:, n]·b[:, n]) over the batch as two float64 (dim,) tensors (bjx_row_moments; b=None: a²)."""
    ac, dim, batch, _ = _prep(a)
    bc = None
    if b is not None:
        bc, bdim, bbatch, _ = _prep(b)
        if (bdim, bbatch) != (dim, batch) or bc.dtype != ac.dtype:
            raise ValueError("DimensionMismatch: a and b must have the same shape and dtype")
    ctx = context(ac.device)
    out = torch.empty(2 * dim + 1, dtype=torch.float64, device=ac.device)
    L.check(ctx.h, L.load().bjx_row_moments(ctx.h, _dt(ac), _ptr(ac), _ptr(bc), _ptr(out), dim, batch), "bjx_row_moments")
    return out[:dim], out[dim:2 * dim]


def _vjp_params_leading_affine(b, x, out_bar, ladj_bar):
    """Parameter pullback of a chain `tail ∘ Shift(μ) ∘ Scale(σ)` (either stage optional) — the mean-field family
    y = tail(μ + σ ⊙ z): with z̄ the input cotangent, v̄ = z̄/σ is the cotangent behind the affine stage, so
        μ̄ = Σ_n z̄_n / σ,      σ̄ = (Σ_n z̄_n ⊙ z_n + Σ_n ℓ̄_n) / σ          (log|σ| enters every column's log-det)
    — two row reductions over the batch (bjx_row_moments) after the input pullback.  Scalar parameters get the sum over
    the rows.  Returns (z_bar, {"scale": σ̄ or None, "shift": μ̄ or None})."""
    stages = b._stages() if isinstance(b, ComposedFunction) else [b]
    scale = shift = None
    k = 0
    if k < len(stages) and isinstance(stages[k], Scale):
        scale = stages[k]
        k += 1
    if k < len(stages) and isinstance(stages[k], Shift):
        shift = stages[k]
        k += 1
    if scale is None and shift is None:
        raise NotImplementedError(f"no device parameter pullback for {b!r}: expected a chain that starts with Scale and/or Shift (SURVEY.md §8f f-1)")
    if _elementwise_ops(b) is not None and x.dim() == 2:
        # input pullback and both row reductions in one pass over x and ȳ (bjx_stacked_vjp_moments)
        zb, s1, s2 = Stacked([b], [(1, x.shape[0])])._vjp(x, out_bar, ladj_bar, moments=True)
    else:
        zb = vjp(b, x, out_bar, ladj_bar)
        s1, s2 = row_moments(zb, x)
    xc, dim, batch, _ = _prep(x)
    sig = None
    if scale is not None:
        sig = _param(scale.a, xc).to(torch.float64).reshape(-1)
    inv_sig = 1.0 if sig is None else 1.0 / sig
    lsum = 0.0
    if ladj_bar is not None:
        lsum = float(ladj_bar) * batch if not isinstance(ladj_bar, torch.Tensor) else ladj_bar.to(torch.float64).sum()
    out = {"scale": None, "shift": None}
    if shift is not None:
        mb = s1 * inv_sig
        out["shift"] = (mb if _is_seq(shift.a) else mb.sum()).to(xc.dtype)
    if scale is not None:
        sb = (s2 + lsum) * inv_sig
        out["scale"] = (sb if _is_seq(scale.a) else sb.sum()).to(xc.dtype)
    return zb, out


def _chain_of(stages):
    """ComposedFunction of `stages` given in APPLICATION order (first applied first); one stage stays itself."""
    out = stages[0]
    for st in stages[1:]:
        out = ComposedFunction(st, out)
    return out


def _vjp_params_affine_anywhere(b, x, out_bar, ladj_bar=None):
    """Parameter pullback of EVERY `Scale(a)` / `Shift(b)` stage of a chain, wherever it sits (the parameter side of the Scale
    adjoints in ext/BijectorsReverseDiffExt.jl:69-115; the reference leaves chains to the AD package).  The chain is cut at its
    affine stages: forward through the pieces to get each stage's input z, then backwards — `vjp` of a piece, and at a stage with
    the cotangent g of its output
        Shift(b):  b̄ = Σ_n g_n,                     input cotangent g
        Scale(a):  ā = Σ_n g_n ⊙ z_n + Σ_n ℓ̄_n / a,   input cotangent a ⊙ g        (log|a| is in every column's log-det)
    (row sums: bjx_row_moments; scalar parameters get the sum over the rows).  A host composition of existing launches — a few
    passes per affine stage; the mean-field head `tail ∘ Shift ∘ Scale` keeps its one-pass kernel (_vjp_params_leading_affine).
    Returns (x_bar, {"stages": [None | cotangent, ...]}) aligned with the chain's stages in application order."""
    stages = b._stages() if isinstance(b, ComposedFunction) else [b]
    aff = [i for i, st in enumerate(stages) if isinstance(st, (Scale, Shift)) and not getattr(st, "matrix", False)]
    if not aff:
        raise NotImplementedError(f"no device parameter pullback for {b!r}: the chain has no Scale / Shift stage (SURVEY.md §8f f-1)")
    xc, dim, batch, _ = _prep(x)
    lsum = 0.0
    if ladj_bar is not None:
        lsum = float(ladj_bar) * batch if not isinstance(ladj_bar, torch.Tensor) else ladj_bar.to(torch.float64).sum()
    # forward: the input of every piece / affine stage
    cuts = []                       # (piece stages before the affine stage, input of the piece, affine stage index, input of the affine stage)
    cur, lo = x, 0
    for i in aff:
        piece = stages[lo:i]
        z = transform(_chain_of(piece), cur) if piece else cur
        cuts.append((piece, cur, i, z))
        cur = transform(stages[i], z)
        lo = i + 1
    tail = stages[lo:]
    # backward
    g = vjp(_chain_of(tail), cur, out_bar, ladj_bar) if tail else out_bar
    grads = [None] * len(stages)
    for piece, pin, i, z in reversed(cuts):
        st = stages[i]
        s1, s2 = row_moments(g, z)
        if isinstance(st, Shift):
            grads[i] = (s1 if _is_seq(st.a) else s1.sum()).to(xc.dtype)
        else:
            a64 = _param(st.a, xc).to(torch.float64).reshape(-1)
            ab = s2 + lsum / a64
            grads[i] = (ab if _is_seq(st.a) else ab.sum()).to(xc.dtype)
            g = transform(Scale(st.a), g)
        if piece:
            g = vjp(_chain_of(piece), pin, g, ladj_bar)
    return g, {"stages": grads}


def vjp_params(b, x, out_bar, ladj_bar=None):
    """Pullback of `with_logabsdet_jacobian(b, x)` onto the input AND the parameters of a PlanarLayer (stack):
    returns (x_bar, {"w": w_bar, "u": u_bar, "b": b_bar}) with the parameter cotangents summed over the batch and the
    shapes of b.w / b.u / b.b (bjx_planar_vjp_params; closed-form derivatives of planar_layer.jl:65-110).
    For a RadialLayer: (x_bar, {"alpha_", "beta", "z_0"}) — see _vjp_params_radial.
    For inverse(PlanarLayer) / inverse(RadialLayer): the same dictionaries through the implicit function theorem — see _vjp_params_inverse.
    For a RationalQuadraticSpline or its inverse: (x_bar, {"widths", "heights", "derivatives"[, "raw_widths", ...]}) — see _vjp_params_rqs.
    For a chain that starts with Scale and/or Shift: (z_bar, {"scale": σ̄, "shift": μ̄}) — see _vjp_params_leading_affine.
    For a chain with Scale / Shift stages anywhere else: (x_bar, {"stages": [...]}) — see _vjp_params_affine_anywhere.
    For a composition that contains flow layers / splines / BatchNorm: (x_bar, {"stages": [...]}) — see _vjp_params_composed."""
    if isinstance(b, Inverse) and isinstance(b.orig, (PlanarLayer, RadialLayer)):
        return _vjp_params_inverse(b, x, out_bar, ladj_bar)
    if isinstance(b, RadialLayer):
        return _vjp_params_radial(b, x, out_bar, ladj_bar)
    if isinstance(b, RationalQuadraticSpline) or (isinstance(b, Inverse) and isinstance(b.orig, RationalQuadraticSpline)):
        return _vjp_params_rqs(b, x, out_bar, ladj_bar)
    if isinstance(b, InvertibleBatchNorm):
        return _vjp_params_batchnorm(b, x, out_bar, ladj_bar)
    if (isinstance(b, Scale) and b.matrix) or (isinstance(b, Inverse) and isinstance(b.orig, Scale) and b.orig.matrix):
        return _vjp_params_scale_matrix(b, x, out_bar, ladj_bar)
    if isinstance(b, ComposedFunction) and any(_has_own_params(st) for st in b._stages()):
        return _vjp_params_composed(b, x, out_bar, ladj_bar)
    if not isinstance(b, PlanarLayer):
        stages = b._stages() if isinstance(b, ComposedFunction) else [b]
        lead = 0
        if lead < len(stages) and isinstance(stages[lead], Scale):
            lead += 1
        if lead < len(stages) and isinstance(stages[lead], Shift):
            lead += 1
        if lead > 0 and not any(isinstance(st, (Scale, Shift)) for st in stages[lead:]):
            return _vjp_params_leading_affine(b, x, out_bar, ladj_bar)
        return _vjp_params_affine_anywhere(b, x, out_bar, ladj_bar)
    xc, dim, batch, vec = _prep(x)
    gc, gdim, gbatch, _ = _prep(out_bar)
    if (gdim, gbatch) != (dim, batch) or gc.dtype != xc.dtype:
        raise ValueError("DimensionMismatch: out_bar must have the shape and dtype of the output")
    w, u, bb = b._tables(xc, dim)
    two_d = b.n_layers > 1 or isinstance(b, _PlanarRun) or b.w.dim() == 2
    lb = _ladj_bar(ladj_bar, batch, xc)
    ctx = context(xc.device)
    xb = _empty(dim, batch, xc, vec)
    wb, ub, bbar = torch.empty_like(w), torch.empty_like(u), torch.empty_like(bb)
    work = torch.empty(2 * b.n_layers * max(batch, 1), dtype=xc.dtype, device=xc.device)
    rc = L.load().bjx_planar_vjp_params(ctx.h, _dt(xc), _ptr(w), _ptr(u), _ptr(bb), b.n_layers, _ptr(xc), _ptr(gc), _ptr(lb), _ptr(xb),
                                        _ptr(wb), _ptr(ub), _ptr(bbar), _ptr(work), dim, batch)
    L.check(ctx.h, rc, "bjx_planar_vjp_params")
    if two_d:
        wb, ub = wb.reshape(b.n_layers, dim).T, ub.reshape(b.n_layers, dim).T                       # back to (dim, n_layers)
    return xb, {"w": wb, "u": ub, "b": bbar}


def _vjp_params_inverse(ib, y, x_bar, ladj_bar=None):
    """Parameter pullback of the INVERSE of a flow layer (maximum-likelihood training evaluates inverse(flow) on the data;
    the reference differentiates it through the `find_alpha` rule, ext/BijectorsChainRulesCoreExt.jl:42-46, layer by layer).
    Implicit function theorem on the whole stack: with x = f⁻¹(y; θ) and the inverse's log-det -ℓ(x; θ),
        ȳ = J⁻ᵀ (x̄ - ℓ̄ ∇ₓℓ)                       (the input pullback of the inverse, bjx_planar_vjp / bjx_radial_vjp)
        θ̄ = (∂f/∂θ)ᵀ(-ȳ) + (-ℓ̄) ∂ℓ/∂θ              (the FORWARD parameter pullback at x with cotangents -ȳ, -ℓ̄)
    so three existing launches — inverse transform, inverse input pullback, forward parameter pullback — and no new kernel;
    Newton's root is not differentiated through.  -> (y_bar, the forward layer's parameter dictionary)."""
    f = ib.orig
    x = transform(ib, y)
    y_bar = vjp(ib, y, x_bar, ladj_bar)
    _, batch = _prep(y)[1:3]
    lb = None if ladj_bar is None else -_ladj_bar(ladj_bar, batch, _prep(y)[0])
    _, grads = vjp_params(f, x, -y_bar, lb)
    return y_bar, grads


def _vjp_params_rqs(b, x, out_bar, ladj_bar=None):
    """RationalQuadraticSpline (or inverse(spline)): input pullback + the cotangents of the knot arrays summed over
    the batch, ONE pass over x, ȳ, ℓ̄ (bjx_rqs_vjp_knots with in_bar; closed-form derivatives of rational_quadratic_spline.jl:128-357 — the reference leaves them to
    the AD package).  A spline built with the `B` constructor (:109-123) also gets the cotangents of its unconstrained
    parameters (bjx_rqs_params_vjp: softmax/cumsum and log1pexp backwards) as "raw_widths", "raw_heights", "raw_derivatives"."""
    inv = isinstance(b, Inverse)
    sp = b.orig if inv else b
    xc, dim, batch, vec = _prep(x)
    gc, gdim, gbatch, _ = _prep(out_bar)
    if (gdim, gbatch) != (dim, batch) or gc.dtype != xc.dtype:
        raise ValueError("DimensionMismatch: out_bar must have the shape and dtype of the output")
    if dim != sp.widths.shape[0]:
        raise ValueError(f"DimensionMismatch: spline with {sp.widths.shape[0]} rows applied to {dim} rows")
    w, h, d = (colmajor(_param(t, xc)) for t in (sp.widths, sp.heights, sp.derivatives))
    K1 = int(sp.widths.shape[1])
    lb = _ladj_bar(ladj_bar, batch, xc)
    ctx = context(xc.device)
    outs = [torch.empty((K1, dim), dtype=xc.dtype, device=xc.device).T for _ in range(3)]
    xb = _empty(dim, batch, xc, vec)
    rc = L.load().bjx_rqs_vjp_knots(ctx.h, _dt(xc), int(inv), _ptr(w), _ptr(h), _ptr(d), K1, _ptr(xc), _ptr(gc), _ptr(lb),
                                    _ptr(xb), *[_ptr(o) for o in outs], dim, batch)
    L.check(ctx.h, rc, "bjx_rqs_vjp_knots")
    grads = {"widths": outs[0], "heights": outs[1], "derivatives": outs[2]}
    if sp._raw is not None and sp._raw[0].dtype == xc.dtype:
        rw, rh, rd, B = sp._raw
        K = K1 - 1
        routs = [torch.empty((k, dim), dtype=xc.dtype, device=xc.device).T for k in (K, K, max(K - 1, 1))]
        rc = L.load().bjx_rqs_params_vjp(ctx.h, _dt(xc), _ptr(rw), _ptr(rh), _ptr(rd), K, dim, B, *[_ptr(o) for o in outs], *[_ptr(o) for o in routs])
        L.check(ctx.h, rc, "bjx_rqs_params_vjp")
        grads.update({"raw_widths": routs[0], "raw_heights": routs[1], "raw_derivatives": routs[2][:, :K - 1]})
    return xb, grads


def _vjp_params_batchnorm(bn, x, out_bar, ladj_bar=None):
    """InvertibleBatchNorm in eval mode (normalise.jl:39 marks `b, logs` trainable): y = γ (x − m) + b, γ = exp(logs)/√(v+ε),
    logabsdetjac[n] = Σ_c (logs_c − ½ log(v_c+ε)).  x̄ = γ ȳ;  b̄ = Σ_n ȳ;  l̄ogs_c = Σ_n ȳ_{c,n} (y_{c,n} − b_c) + Σ_n ℓ̄_n.
    ONE pass: the input pullback of the affine chain with its row moments (bjx_stacked_vjp_moments: Σ_n x̄ and Σ_n x̄·x),
    from which b̄ = Σx̄/γ and l̄ogs = Σx̄x − m Σx̄ + Σℓ̄.  -> (x_bar, {"b": ..., "logs": ...})."""
    if istraining():
        return _vjp_params_batchnorm_training(bn, x, out_bar, ladj_bar)
    xc, dim, batch, vec = _prep(x)
    if vec:
        raise ValueError("InvertibleBatchNorm needs an input with at least 2 dimensions")
    logs, v, m, bb = (_param(t, xc) for t in (bn.logs, bn.v, bn.m, bn.b))
    gam = torch.exp(logs) / torch.sqrt(v + bn.eps)
    aff = Shift(bb) @ Scale(gam) @ Shift(-m)
    xb, m1, m2 = Stacked([aff], [(1, dim)])._vjp(x, out_bar, ladj_bar, moments=True)
    lb = _ladj_bar(ladj_bar, batch, xc)
    lsum = lb.double().sum() if lb is not None else 0.0
    b_bar = (m1 / gam.double()).to(xc.dtype)
    logs_bar = (m2 - m.double() * m1 + lsum).to(xc.dtype)
    return xb, {"b": b_bar, "logs": logs_bar}


def _vjp_params_batchnorm_training(bn, x, out_bar, ladj_bar=None):
    """InvertibleBatchNorm in TRAINING mode (normalise.jl:51-60): the batch mean and variance are functions of x, so the input
    cotangent has the two centring terms of batch normalisation plus the derivative of -½ Σℓ̄ log(v + ε):
        x̄ = γ/σ [ȳ − mean ȳ − x̂ mean(ȳ x̂)] − (Σℓ̄/N) x̂/σ,   b̄ = Σ ȳ,   l̄ogs = γ Σ ȳ x̂ + Σℓ̄
    Two passes over (ȳ, x): bjx_row_moments (Σȳ, Σȳ·x per channel) → one all-reduce of 2·dim+2 doubles when the bijector is
    sharded (`sync`) → bjx_batchnorm_train_vjp.  Uses the batch statistics saved by the training-mode forward call
    when that call saw this very tensor (object identity + version), and recomputes them from x otherwise.  -> (x_bar, {"b": ..., "logs": ...})."""
    xc, dim, batch, vec = _prep(x)
    if vec:
        raise ValueError("InvertibleBatchNorm needs an input with at least 2 dimensions")
    gc, gdim, gbatch, _ = _prep(out_bar)
    if (gdim, gbatch) != (dim, batch) or gc.dtype != xc.dtype:
        raise ValueError("DimensionMismatch: out_bar must have the shape and dtype of the output")
    st = getattr(bn, "_batch_stats", None)
    if st is not None and st[0].numel() == dim and st[0].dtype == xc.dtype and st[2].matches(x):
        mean_b, var_b = st[0], st[1]                # the statistics of the forward call on this very tensor
    else:
        # another batch went through the bijector since (micro-batches, a validation batch, the layer used twice in a flow), x is
        # a different tensor object with the same values, or it was written: the batch mean and variance are functions of x alone
        # — recompute them from the x of THIS pullback (storage identity is not a batch identity).  The shift of the
        # sums is the current moving mean; mean = m + Σ(x − m)/N and var = Σ(x − m)²/N − (Σ(x − m)/N)² do not depend on it.
        stats = bn.batch_stats(x)
        grp0 = bn._sync_group()
        if grp0 is not False:
            from . import shard as _shard

            _shard.allreduce_logabsdetjac(stats, grp0)
        n = stats[2 * dim]
        s1, s2 = stats[:dim] / n, stats[dim:2 * dim] / n
        mean_b, var_b = (_param(bn.m, xc).double() + s1).to(xc.dtype), (s2 - s1 * s1).to(xc.dtype)
    ctx = context(xc.device)
    lib = L.load()
    mom = torch.empty(2 * dim + 2, dtype=torch.float64, device=xc.device)       # (Σȳ, Σȳx, N, Σℓ̄): ONE bucket for the collective
    L.check(ctx.h, lib.bjx_row_moments(ctx.h, _dt(xc), _ptr(gc), _ptr(xc), _ptr(mom), dim, batch), "bjx_row_moments")
    lb = _ladj_bar(ladj_bar, batch, xc)
    mom[2 * dim + 1] = lb.double().sum() if lb is not None else 0.0
    grp = bn._sync_group()
    if grp is not False:
        from . import shard as _shard

        _shard.allreduce_logabsdetjac(mom, grp)
    logs = _param(bn.logs, xc)
    xb = _empty(dim, batch, xc, vec)
    b_bar, logs_bar = torch.empty(dim, dtype=xc.dtype, device=xc.device), torch.empty(dim, dtype=xc.dtype, device=xc.device)
    rc = lib.bjx_batchnorm_train_vjp(ctx.h, _dt(xc), _ptr(logs), _ptr(mean_b), _ptr(var_b), float(bn.eps), _ptr(mom), mom.data_ptr() + 8 * (2 * dim + 1),
                                     _ptr(xc), _ptr(gc), _ptr(xb), _ptr(b_bar), _ptr(logs_bar), dim, batch)
    L.check(ctx.h, rc, "bjx_batchnorm_train_vjp")
    return xb, {"b": b_bar, "logs": logs_bar}


def _vjp_params_radial(b, x, out_bar, ladj_bar=None):
    """(x_bar, {"alpha_": ᾱ_, "beta": β̄, "z_0": z̄₀}) for a RadialLayer — the raw parameters behind softplus
    (radial_layer.jl:43-60), cotangents summed over the batch (bjx_radial_vjp_params)."""
    xc, dim, batch, vec = _prep(x)
    gc, gdim, gbatch, _ = _prep(out_bar)
    if (gdim, gbatch) != (dim, batch) or gc.dtype != xc.dtype:
        raise ValueError("DimensionMismatch: out_bar must have the shape and dtype of the output")
    z0 = _param(b.z_0, xc)
    if z0.numel() != dim:
        raise ValueError(f"DimensionMismatch: RadialLayer of dimension {z0.numel()} applied to {dim} rows")
    a, be = _param(b.alpha_, xc), _param(b.beta, xc)
    lb = _ladj_bar(ladj_bar, batch, xc)
    ctx = context(xc.device)
    xb = _empty(dim, batch, xc, vec)
    ab, bb, zb = torch.empty_like(a), torch.empty_like(be), torch.empty_like(z0)
    work = torch.empty(2 * max(batch, 1), dtype=xc.dtype, device=xc.device)
    rc = L.load().bjx_radial_vjp_params(ctx.h, _dt(xc), _ptr(a), _ptr(be), _ptr(z0), _ptr(xc), _ptr(gc), _ptr(lb), _ptr(xb),
                                        _ptr(ab), _ptr(bb), _ptr(zb), _ptr(work), dim, batch)
    L.check(ctx.h, rc, "bjx_radial_vjp_params")
    return xb, {"alpha_": ab, "beta": bb, "z_0": zb}


# ------------------------------------------------------------------ columnwise (src/interface.jl:41-78)
class Columnwise(Transform):
    """`columnwise(f)` = Base.Fix1(eachcolmaphcat, f): `f` applied to every column, log-det = the SUM over the
    columns (src/interface.jl:71-78).  Every device kernel here is already batched over columns, so this is
    the same launch with the reference's scalar return shape; it is how the reference batches the bijectors
    that only have vector methods (RQS with matrix parameters, Coupling, VecCholesky, Stacked)."""

    def __init__(self, f):
        self.x = f        # the reference's field name (Base.Fix1.x)

    def _key(self):
        return (self.x,)

    def _wlj(self, x, per_sample, want_ladj=True):
        if x.dim() != 2 and not isinstance(self.x, (VecCholeskyBijector,)):
            raise ValueError("columnwise(f) applies to a matrix (one sample per column)")
        if not want_ladj:
            return self.x._wlj(x, per_sample=False, want_ladj=False)
        if per_sample is True:
            return self.x._wlj(x, per_sample=True)
        if per_sample in ("both", "sum64"):
            return self.x._wlj(x, per_sample=per_sample)
        y, (ps, sm) = self.x._wlj(x, per_sample="both")
        return y, sm[0].to(ps.dtype)          # interface.jl:75-77: one scalar, summed over the columns (float64 accumulation)


def columnwise(f):
    """src/interface.jl:70"""
    return Columnwise(f)


# ------------------------------------------------------------------ TransformedDistribution
class MvNormal:
    """Base distribution of a `TransformedDistribution` (src/transformed_distribution.jl:159-240 takes any `MvNormal`):
    `MvNormal(dim)` the standard normal (the base of every flow in the reference's docs/tests, e.g. test/normalising_flows.jl:74-91),
    `MvNormal(μ, σ)` the diagonal `MvNormal(μ, Diagonal(σ.^2))`, `MvNormal(μ, cov=Σ)` / `MvNormal(μ, scale_tril=L)` a FULL
    covariance Σ = L Lᵀ: whitening is the matrix `Scale` of scale.jl:14-36 (x ↦ L⁻¹ (x − μ), log-det −logabsdet L: bjx_scale_matrix)."""

    def __init__(self, mu, sigma=None, cov=None, scale_tril=None):
        self.scale_tril = None
        if isinstance(mu, int) and sigma is None and cov is None and scale_tril is None:
            self.dim, self.mu, self.sigma = mu, None, None
        else:
            self.mu = torch.as_tensor(mu).reshape(-1)
            self.sigma = None if sigma is None else torch.as_tensor(sigma).reshape(-1)
            self.dim = self.mu.numel()
            if cov is not None or scale_tril is not None:
                if sigma is not None or (cov is not None and scale_tril is not None):
                    raise ValueError("MvNormal: give ONE of sigma (diagonal), cov or scale_tril")
                Lm = torch.linalg.cholesky(torch.as_tensor(cov)) if cov is not None else torch.as_tensor(scale_tril)   # dim x dim: host-side prep, once
                if tuple(Lm.shape) != (self.dim, self.dim):
                    raise ValueError(f"DimensionMismatch: covariance factor of shape {tuple(Lm.shape)} for a mean of length {self.dim}")
                self.scale_tril = Lm

    def _tril_on(self, like):
        """(Scale(L) as a bijector, -L⁻¹μ) on the device / dtype of `like`, cached"""
        key = (like.device, like.dtype, id(self.scale_tril), self.scale_tril._version, id(self.mu), self.mu._version)
        if getattr(self, "_tril_key", None) != key:
            Ld = self.scale_tril.to(device=like.device, dtype=like.dtype)
            mu = self.mu.to(device=like.device, dtype=like.dtype)
            self._tril_cache = (Scale(colmajor(Ld)), -torch.linalg.solve_triangular(Ld, mu[:, None], upper=False)[:, 0].contiguous(), mu)
            self._tril_key = key
        return self._tril_cache

    def _whiten_ops(self):
        """x -> (x - μ)/σ as chain ops; SCALE_INV's log-det supplies the -Σ log σ of the density."""
        ops = []
        if self.mu is not None:
            key = (id(self.mu), self.mu._version)                           # negated once per value of μ (in-place updates bump _version)
            if getattr(self, "_neg_mu_key", None) != key:
                self._neg_mu, self._neg_mu_key = -self.mu, key
            ops.append((L.OP_SHIFT, self._neg_mu, None))
        if self.sigma is not None:
            ops.append((L.OP_SCALE_INV, self.sigma, None))
        return ops

    def _color_ops(self):
        ops = []
        if self.sigma is not None:
            ops.append((L.OP_SCALE, self.sigma, None))
        if self.mu is not None:
            ops.append((L.OP_SHIFT, self.mu, None))
        return ops


class TorchBase:
    """Any other base distribution (src/transformed_distribution.jl:159-240 is generic in `td.dist`; VERDICT r03 missing #6): an adapter
    around a `torch.distributions.Distribution` whose event is the column of `dim` rows — e.g. `Independent(Laplace(loc, scale), 1)`,
    `MultivariateNormal`, `StudentT` factors, a mixture.  The protocol `logpdf` / `rand` use is duck-typed, any object with
        logpdf(x[dim, N]) -> [N]      and      rand(n, device, dtype, seed) -> x[dim, n] (column-major)
    can be the `dist` of `transformed`.  Nothing is fused for such a base: the flow's kernel writes the pre-image, the base reads it."""

    def __init__(self, dist, dim=None):
        self.dist = dist
        es = tuple(dist.event_shape) if len(dist.event_shape) else tuple(dist.batch_shape)
        self.dim = int(dim if dim is not None else (es[-1] if es else 1))

    def logpdf(self, x):
        lp = self.dist.log_prob(x.T)                     # rows of x.T are samples
        return lp.reshape(lp.shape[0], -1).sum(dim=1) if lp.dim() > 1 else lp     # independent factors given as a batch shape

    def rand(self, n, device, dtype, seed=0):
        with torch.random.fork_rng(devices=[device] if torch.device(device).type == "cuda" else []):
            torch.manual_seed(seed)
            smp = self.dist.sample((n,)).reshape(n, -1).to(device=device, dtype=dtype)
        return colmajor(smp.T)


def _preimage(ib, y):
    """(x, per-column log-det) of the inverse transform, materialised"""
    if ib is identity:
        return y, None
    x, lj = with_logabsdet_jacobian(ib, y, per_sample=True) if isinstance(ib, ComposedFunction) else ib._wlj(y, per_sample=True)
    if hasattr(x, "result"):
        x, lj = x.result, x.logabsdetjac
    return x, lj


class TransformedDistribution:
    """src/transformed_distribution.jl:2-12: `transformed(dist, b)`; y = b(x), x ~ dist."""

    def __init__(self, dist, transform):
        self.dist, self.transform = dist, transform


def transformed(dist, b=None):
    """src/transformed_distribution.jl:20-28"""
    return TransformedDistribution(dist, identity if b is None else b)


def _marshal_ops(ops, xc, dim):
    """[(kind, p0, p1)] -> (ctypes bjx_op array, tensors to keep alive): the marshalling of `_run_chain`, without its cache."""
    arr = (L.BjxOp * max(len(ops), 1))()
    keep = []
    for i, (kind, p0, p1) in enumerate(ops):
        o = arr[i]
        o.kind, o.param_len, o.p0, o.p1, o.v0, o.v1 = kind, 0, 0.0, 0.0, None, None
        seq = any(_is_seq(p) for p in (p0, p1) if p is not None)
        for j, p in enumerate((p0, p1)):
            if p is None:
                continue
            if seq:
                t = _param(p, xc).reshape(-1) if _is_seq(p) else torch.full((dim,), float(p), dtype=xc.dtype, device=xc.device)
                if t.numel() != dim:
                    raise ValueError(f"DimensionMismatch: parameter of length {t.numel()} for input with {dim} rows")
                keep.append(t)
                o.param_len = dim
                setattr(o, f"v{j}", t.data_ptr())
            else:
                o.param_len = 1
                setattr(o, f"p{j}", float(p))
    return arr, keep


_MATRIX_CHAIN_KINDS = (L.OP_EXP, L.OP_LOG, L.OP_SHIFT, L.OP_SCALE, L.OP_SCALE_INV)


_AFFINE_KINDS = (L.OP_SHIFT, L.OP_SCALE, L.OP_SCALE_INV)


def _merge_affine_tail(d, ops, y):
    """The runs of three or more affine stages of a density / sampling chain — the tail of the inverse transform (Shift / Scale with scalars)
    followed by the whitening (x − μ)/σ, (any run of them in the list) — each collapsed into ONE Scale and
    ONE Shift with per-row parameters: ((x + s)·c − μ)/σ = x·(c/σ) + (s·c − μ)/σ.  The chain kernel is issue-bound on such passes (six stages,
    nothing stored: 50 % of the HBM peak; four: 59 %), the log-det is unchanged (Σ log|c/σ| is what the Scale stages add up to).  The merged
    vectors are kept on the distribution object, keyed by the scalars and the (id, version) of the tensors."""
    if not isinstance(y, torch.Tensor) or y.dim() != 2:
        return ops
    dim = y.shape[0]
    out, i, n, run_no = [], 0, len(ops), 0
    while i < n:
        j = i
        while j < n and ops[j][0] in _AFFINE_KINDS and ops[j][2] is None and isinstance(ops[j][1], (int, float, torch.Tensor)):
            j += 1
        if j - i < 3:
            out.extend(ops[i:max(j, i + 1)])
            i = max(j, i + 1)
            continue
        run = ops[i:j]
        key = (y.dtype, y.device.index, dim, run_no) + tuple((k, id(p), p._version) if isinstance(p, torch.Tensor) else (k, float(p)) for k, p, _ in run)
        cache = d.__dict__.setdefault("_affine_runs", {})
        hit = cache.get(run_no)
        if hit is None or hit[0] != key:
            A = torch.ones(dim, dtype=torch.float64, device=y.device)
            B = torch.zeros(dim, dtype=torch.float64, device=y.device)
            ok = True
            for kind, p, _ in run:
                pv = p.to(device=y.device, dtype=torch.float64).reshape(-1) if isinstance(p, torch.Tensor) else float(p)
                if isinstance(pv, torch.Tensor) and pv.numel() not in (1, dim):
                    ok = False
                    break
                if kind == L.OP_SHIFT:
                    B = B + pv
                elif kind == L.OP_SCALE:
                    A, B = A * pv, B * pv
                else:
                    A, B = A / pv, B / pv
            if not ok:
                out.extend(run)
                i = j
                run_no += 1
                continue
            hit = (key, A.to(y.dtype).contiguous(), B.to(y.dtype).contiguous())
            cache[run_no] = hit
        out.extend([(L.OP_SCALE, hit[1], None), (L.OP_SHIFT, hit[2], None)])
        i = j
        run_no += 1
    return out


def _logpdf_full_cov_fused(d, ib, y):
    """Full-covariance base, fusable inverse transform, dim <= 128: ONE launch when the inverse is at most three stages of exp / log / Shift /
    Scale / Scale⁻¹ (bjx_scale_matrix_chain, round 6); else (round 5) TWO launches and three array passes instead of four
    launches and five passes — the inverse chain with the shift −μ appended writes x − μ and its per-column log-det, then the matrix
    `Scale` kernel whitens with L⁻¹ and accumulates log N(z; 0, I) − logabsdet L per column while the whitened tile is still in LDS
    (BJX_BASE_STDNORMAL on bjx_scale_matrix: nothing is stored).  None when the shape is not served (the caller takes the general path)."""
    yc, dim, batch, vec = _prep(y)
    if vec or dim > 128 or batch == 0:
        return None
    ops = [] if ib is identity else _fused_ops(ib)
    if ops is None or len(ops) + 1 > L.BJX_MAX_OPS:
        return None
    sc, _, mu = d._tril_on(yc)
    key = (id(mu), mu._version)
    if getattr(d, "_neg_mu_full_key", None) != key:
        d._neg_mu_full, d._neg_mu_full_key = (-mu).contiguous(), key
    pre = list(ops) + [(L.OP_SHIFT, d._neg_mu_full, None)]
    if len(pre) <= 4 and all(k in _MATRIX_CHAIN_KINDS for k, _, _ in pre):
        # ONE launch (round 6, bjx_scale_matrix_chain): the inverse chain and the shift by the mean are applied to each tile as the matrix-core
        # kernel loads it, the whitened tile never leaves LDS — y is read once, nothing but the densities is written
        a1 = colmajor(_param(sc.a, yc))
        ctx1 = context(yc.device)
        _note_params(ctx1, a1)
        arr, keep = _marshal_ops(pre, yc, dim)
        lp = torch.empty(batch, dtype=yc.dtype, device=yc.device)
        rc = L.load().bjx_scale_matrix_chain(ctx1.h, _dt(yc), 1, _ptr(a1), arr, len(pre), _ptr(yc), None, _ptr(lp), dim, batch, L.BJX_BASE_STDNORMAL)
        del keep
        if rc == 0:
            return lp
        if rc != L.ERR_UNSUPPORTED:
            L.check(ctx1.h, rc, "bjx_scale_matrix_chain")
    xm, lj = _run_chain(pre, yc, True, True)
    a = colmajor(_param(sc.a, xm))
    ctx = context(xm.device)
    _note_params(ctx, a)
    rc = L.load().bjx_scale_matrix(ctx.h, _dt(xm), 1, _ptr(a), _ptr(xm), None, _ptr(lj), None, dim, batch, L.BJX_ACCUMULATE | L.BJX_BASE_STDNORMAL)
    if rc == L.ERR_UNSUPPORTED:
        return None
    L.check(ctx.h, rc, "bjx_scale_matrix")
    return lj


def logpdf(td: TransformedDistribution, y, reference_shape: bool = False):
    """`logpdf(td::MvTransformed, y::AbstractMatrix)` (src/transformed_distribution.jl:164-169):
        x, logjac = with_logabsdet_jacobian(inverse(td.transform), y);  logpdf(td.dist, x) + logjac
    evaluated in ONE pass over `y` where the inverse is a fused chain or a PlanarLayer stack: the base density
    is accumulated inside the kernel that inverts the flow and the pre-image `x` is never stored (the reference
    marks this path "TODO: implement more efficiently for flows").  Returns the per-column log-density.

    reference_shape=True reproduces what the reference's `+` does literally: elementwise bijectors return ONE
    scalar log-det for the whole matrix, which it adds to every column (SURVEY.md §8a″)."""
    ib = inverse(td.transform)
    d = td.dist
    if not isinstance(d, MvNormal):
        # any base with logpdf(x[dim, N]) -> [N] (TorchBase): transformed_distribution.jl:164-169 literally
        x, lj = _preimage(ib, y)
        lp = d.logpdf(x)
        return lp if lj is None else lp + lj
    if getattr(d, "scale_tril", None) is not None:
        # full covariance: x = b⁻¹(y); z = L \ x (bjx_scale_matrix, log-det −logabsdet L per column); then the standard-normal
        # density of z − L⁻¹μ accumulated without storing anything (one chain launch, store=False)
        fused = _logpdf_full_cov_fused(d, ib, y)
        if fused is not None:
            return fused
        if ib is identity:
            x, lj = y, None
        else:
            x, lj = ib._wlj(y, per_sample=True) if not isinstance(ib, ComposedFunction) else with_logabsdet_jacobian(ib, y, per_sample=True)
            if hasattr(x, "result"):
                x, lj = x.result, x.logabsdetjac
        sc, shift, _ = d._tril_on(x)
        z, lz = inverse(sc)._wlj(x, per_sample=True)
        lp = _run_chain([(L.OP_SHIFT, shift, None), (L.OP_STDNORMAL_LOGPDF, None, None)], z, True, True, store=False)[1]
        return lp + lz if lj is None else lp + lz + lj
    base = d._whiten_ops() + [(L.OP_STDNORMAL_LOGPDF, None, None)]
    if reference_shape:
        x, lj = with_logabsdet_jacobian(ib, y)
        if isinstance(lj, PlanarResult) or hasattr(x, "result"):
            x, lj = x.result, x.logabsdetjac
        lp = _run_chain(base, x, True, True, store=False)[1]
        return lp + lj
    ops = _fused_ops(ib)
    if ops is not None and len(ops) + len(base) <= L.BJX_MAX_OPS:
        return _run_chain(_merge_affine_tail(d, list(ops) + base[:-1], y) + base[-1:], y, True, True, store=False)[1]
    if isinstance(ib, ComposedFunction) and len(ib._plan()[0]) == 1:       # inverse(l8 ∘ … ∘ l1): the planner's single inverse run
        ib = ib._plan()[0][0]
    pl = ib.orig if isinstance(ib, Inverse) else None
    if isinstance(pl, PlanarLayer) and d.mu is None and d.sigma is None:
        return pl._run(y, True, True, True, flags=L.BJX_BASE_STDNORMAL, store=False)[1]
    x, lj = ib._wlj(y, per_sample=True)
    return _run_chain(base, x, True, True, store=False)[1] + lj


def rand(td: TransformedDistribution, n: int, seed: int = 0, device=None, dtype=torch.float32, col0: int = 0, fused: bool = True):
    """`rand(rng, td::MvTransformed, n)` (src/transformed_distribution.jl:214-224: sample the base, push every
    column through the transform): base samples from the counter-based generator of bjx_fill_normal (identical
    for any shard count: keyed by seed, global column and row), colouring μ + σ·z fused into the transform's
    chain when there is one — and then drawn INSIDE that kernel (`fused=False`: fill, then transform; same bits)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if not isinstance(td.dist, MvNormal):
        x = td.dist.rand(n, device, dtype, seed)             # any base (TorchBase protocol), then the transform over all columns
        return x if td.transform is identity else transform(td.transform, x)
    dim = td.dist.dim
    z = torch.empty((n, dim), dtype=dtype, device=device).T
    ctx = context(device)
    if getattr(td.dist, "scale_tril", None) is not None:
        # full covariance: base samples, x = μ + L z (bjx_scale_matrix, then the shift fused into the transform's chain when it has one)
        L.check(ctx.h, L.load().bjx_fill_normal(ctx.h, _dt(z), _ptr(z), dim, n, col0, seed, 0.0, 1.0), "bjx_fill_normal")
        sc, _, mu = td.dist._tril_on(z)
        xz = transform(sc, z)
        ops = _fused_ops(td.transform) if td.transform is not identity else []
        if ops is not None and len(ops) + 1 <= L.BJX_MAX_OPS:
            return _run_chain([(L.OP_SHIFT, mu, None)] + list(ops), xz, False, False, out_y=xz)[0]
        _run_chain([(L.OP_SHIFT, mu, None)], xz, False, False, out_y=xz)
        return transform(td.transform, xz)
    color = td.dist._color_ops()
    ops = _fused_ops(td.transform)
    if ops is not None and len(ops) + len(color) <= L.BJX_MAX_OPS and fused:
        # ONE launch: the base samples are drawn inside the chain kernel (BJX_INPUT_STDNORMAL), never written
        L.check(ctx.h, L.load().bjx_set_rng(ctx.h, seed, col0), "bjx_set_rng")
        return _run_chain(color + list(ops), z, False, False, out_y=z, flags=L.BJX_INPUT_STDNORMAL)[0]
    L.check(ctx.h, L.load().bjx_fill_normal(ctx.h, _dt(z), _ptr(z), dim, n, col0, seed, 0.0, 1.0), "bjx_fill_normal")
    if ops is not None and len(ops) + len(color) <= L.BJX_MAX_OPS:
        return _run_chain(color + list(ops), z, False, False, out_y=z)[0]
    if color:
        _run_chain(color, z, False, False, out_y=z)
    return transform(td.transform, z)
